"""GPU: the RCCL legs of the multi-GPU path, executed on hardware inside the test process.

One process per GPU is the design (zest_parallel.py); the driver's GPU test box has one GPU, so this module
initialises a ONE-rank `nccl` (= RCCL on ROCm) process group in the test process itself (TCP store on 127.0.0.1;
no exec, no relaunch) and switches zest_parallel's rehearsal flag on, under which every exchange step runs its
collective although the group has a single rank: gather_maps (sync and async), broadcast_scene,
allreduce_grads, DyMVSNeRF_G.forward_val's one gather per image and render_sharded around the fused renderer.
The multi-rank arithmetic (shard bounds, row order, averaging) is covered on CPU over gloo with two ranks
(tests/test_sharding_gloo.py, tests/test_generators.py, tests/test_bench_launcher.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def rccl(hip):
    import zest_parallel
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    old = zest_parallel.FORCE_SINGLE_RANK
    zest_parallel.FORCE_SINGLE_RANK = True
    try:
        yield zest_parallel
    finally:
        zest_parallel.FORCE_SINGLE_RANK = old
        torch.cuda.synchronize()
        dist.destroy_process_group()


def test_gather_maps_runs_the_collective(rccl, monkeypatch):
    calls = []
    real = dist.all_gather_into_tensor
    monkeypatch.setattr(dist, "all_gather_into_tensor", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    assert dist.get_backend() == "nccl" and rccl.collectives_active()
    loc = torch.arange(1024 * 16, device="cuda:0", dtype=torch.float32).view(1024, 16)
    full = rccl.gather_maps(loc, 1024)
    assert len(calls) == 1 and full.data_ptr() != loc.data_ptr() and torch.equal(full, loc)
    short = rccl.gather_maps(loc[:1000], 1000, per=1024)                 # a padded last block
    assert short.shape == (1000, 16) and torch.equal(short, loc[:1000])
    t, work = rccl.gather_maps(loc, 1024, async_op=True)                 # enqueued on RCCL's stream
    assert work is not None
    work.wait()
    assert torch.equal(t, loc) and len(calls) == 3


def test_broadcast_and_gradient_allreduce(rccl):
    vol = torch.randn(8, 16, 12, 20, device="cuda:0")
    keep = vol.clone()
    rccl.broadcast_scene([vol, None])
    assert torch.equal(vol, keep)
    params = [torch.nn.Parameter(torch.zeros(s, device="cuda:0")) for s in ((256, 63), (256,), (3, 128), (1,))]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, 0.5 * (i + 1))
    rccl.allreduce_grads(params, bucket_bytes=40000)                     # several buckets; mean over one rank
    torch.cuda.synchronize()
    for i, p in enumerate(params):
        assert torch.equal(p.grad, torch.full_like(p, 0.5 * (i + 1)))


def test_render_sharded_over_the_fused_renderer(rccl):
    """shard_rays -> rendering (fused bf16 maps) -> ONE all-gather of the packed rows: equal to the plain call."""
    import bench
    d = bench.build_workload("nsff_static_mvs_1024x128", 5, torch.device("cuda:0"), rays=256)
    with torch.no_grad():
        want = bench.render_step(d)["zest_packed_maps"].clone()
        rays = {k: v[0] for k, v in d.t.items()}

        def render(loc):
            d.t = {k: v[None] for k, v in loc.items()}
            return bench.render_step(d)["zest_packed_maps"]
        got = rccl.render_sharded(render, rays, 256)
    assert got.shape == want.shape and torch.equal(got, want)


def test_forward_val_gathers_once_per_image_over_rccl(rccl, monkeypatch):
    """The whole-image loop inside the group: one all-gather of the [H W, 13] maps, bit-equal to the image
    rendered without a group."""
    import test_generators as tg
    x = tg._batch(91)
    gen = tg._generator(tg._args(chunk=256, precision=16))
    calls = []
    real = dist.all_gather_into_tensor
    monkeypatch.setattr(dist, "all_gather_into_tensor", lambda *a, **k: (calls.append(a[1].shape), real(*a, **k))[1])
    inside = gen.forward_val(x)
    assert calls == [torch.Size([1024, 13])]
    assert all(len(lst) == 1 for lst in inside[1:])                      # one tensor per map covers the image
    monkeypatch.setattr(rccl, "FORCE_SINGLE_RANK", False)
    gen.chain_bwd = False
    plain = gen.forward_val(x)
    assert len(calls) == 1
    for a, b in zip(inside[1:], plain[1:]):
        assert torch.equal(torch.cat(a), torch.cat(b))
