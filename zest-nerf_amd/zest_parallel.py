"""Ray-sharded rendering across the GPUs of one node (one process per GPU, RCCL over xGMI).

Rays are independent (every op in the renderer is per-sample or a reduction
along one ray), so the path shards with no data-path collective: rank g renders the
contiguous ray block [g*ceil(R/G), ...).  Encoding volumes, source images, cameras and MLP
weights are replicated.  The only exchange is ONE all-gather of the packed per-ray maps
([R/G, 16] fp32 = 64 B per ray) so that every rank ends up with the full set of rendered
pixels; at 1024 rays per GPU that is 64 KiB per rank - latency-bound, one fused collective
instead of one per map.

Works with any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU
tests, where the per-shard render function is the oracle.
"""
import os

import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "shard_rays", "gather_maps", "render_sharded", "broadcast_scene",
           "allreduce_grads", "collectives_active"]

# Rehearsal switch: run the collectives even in a ONE-rank group, so the RCCL legs of this module (and of
# DyMVSNeRF_G.forward_val, bench.py) execute on a single-GPU box: tests/test_hip_rccl.py sets it,
# ZEST_FORCE_COLLECTIVE=1 does the same from the environment.  With more than one rank it changes nothing.
FORCE_SINGLE_RANK = os.environ.get("ZEST_FORCE_COLLECTIVE") == "1"


def _world(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def collectives_active(group=None, force=False):
    """True when the exchange steps run: more than one rank, or a one-rank group under the rehearsal switch."""
    world, _ = _world(group)
    return world > 1 or ((force or FORCE_SINGLE_RANK) and dist.is_available() and dist.is_initialized())


def shard_bounds(n_rays, world, rank):
    """Contiguous block of rank `rank`: (lo, hi, per) with per = ceil(n_rays / world)."""
    per = (n_rays + world - 1) // world
    lo = min(rank * per, n_rays)
    return lo, min(lo + per, n_rays), per


def shard_rays(rays, group=None, dim=0):
    """Slice every tensor of a dict (or a single tensor) to this rank's ray block along `dim`."""
    world, rank = _world(group)

    def cut(t):
        lo, hi, _ = shard_bounds(t.shape[dim], world, rank)
        return t.narrow(dim, lo, hi - lo)
    if torch.is_tensor(rays):
        return cut(rays)
    return {k: (cut(v) if torch.is_tensor(v) else v) for k, v in rays.items()}


def gather_maps(local_maps, n_rays, group=None, async_op=False, force=False, per=None):
    """All-gather per-ray rows.  local_maps [r_local, C] (this rank's block, r_local <= per);
    returns [n_rays, C] on every rank, rows in global ray order.  `per`: rows per rank when the
    blocks are not ceil(n_rays / world) (whole-image loops shard by chunks of rays).  With async_op the collective is
    only enqueued (on RCCL's own stream) and (tensor, work) is returned: call work.wait() before
    reading the tensor, so the gather of one batch overlaps the rendering of the next.
    `force` runs the collective even in a 1-rank group (rehearsal of the code path)."""
    world, rank = _world(group)
    if not collectives_active(group, force):
        return (local_maps, None) if async_op else local_maps
    if per is None:
        _, _, per = shard_bounds(n_rays, world, rank)
    C = local_maps.shape[1]
    send = local_maps
    if local_maps.shape[0] != per:                      # last ranks of an uneven split: pad
        send = local_maps.new_zeros(per, C)
        send[:local_maps.shape[0]] = local_maps
    out = local_maps.new_empty(world * per, C)
    work = dist.all_gather_into_tensor(out, send.contiguous(), group=group, async_op=async_op)
    return (out[:n_rays], work) if async_op else out[:n_rays]


def allreduce_grads(params, group=None, bucket_bytes=32 << 20, force=False):
    """Average the .grad of `params` over the ranks (data-parallel training of the MLPs:
    ~4.9 MB of fp32 gradients per net).  Gradients are packed into buckets of bucket_bytes so
    the ring all-reduce over xGMI moves a few large messages instead of 60 small ones."""
    world, _ = _world(group)
    grads = [p.grad for p in params if p.grad is not None]
    if not collectives_active(group, force) or not grads:
        return
    bucket, size = [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, group=group)
        flat /= world
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        bucket.clear()
    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
            size = 0
    flush()


def render_sharded(render_fn, rays, n_rays, group=None):
    """Shard -> render -> gather.  `render_fn(ray_dict) -> [r_local, C]` renders the local
    block (e.g. the `zest_packed_maps` of renderer.rendering); `rays` holds per-ray tensors
    with the ray dimension first."""
    local = render_fn(shard_rays(rays, group))
    return gather_maps(local, n_rays, group)


def broadcast_scene(tensors, src=0, group=None, force=False):
    """Replicate per-image tensors (encoding volumes, source images, cameras) built on one rank:
    86.5 MB per NSFF volume, once per image, amortised over ~144 ray chunks."""
    if collectives_active(group, force):
        for t in tensors:
            if t is not None:
                dist.broadcast(t, src=src, group=group)
    return tensors
