#!/usr/bin/env python3
"""Throughput of the ZeST-NeRF rendering hot path on MI355X: rendered rays/s.

    python bench.py [--gpus N --steps K --warmup W] [--workload NAME] [--mode bf16|f16|f16x3]
                    [--scaling weak|strong]

A step is one `renderer.rendering(...)` call (the drop-in boundary) over one batch of
synthetic rays through the fused HIP renderer, inputs resident in HBM.  Default workload
(BASELINE.json configs[1]): NSFF Balloon1 geometry, 1024 rays x 128 samples, static MLP,
bf16 MFMA.  With N > 1 there is one rank per GPU - started by torch.distributed.run, or by this
script itself when it is called from a plain shell (it spawns the ranks as child processes
before touching the GPU) - one scene is replicated on all of them (broadcast
from rank 0 before the timed region), rank g renders the g-th contiguous block of one ray batch and the rendered
pixels are all-gathered over RCCL inside the timed region (weak scaling: the batch is --gpus times the workload's
rays, per-GPU work fixed; --scaling strong splits the workload's own rays over the GPUs, e.g. configs[3]:
4096 x 192 over 8).
At N = 1 the JSON also carries `modes`: the fp16 and the fp32-tolerance (split fp16) runs of the
same kernel on the same workload.

Prints ONE JSON line: metric/value/unit..., plus
  roofline:     MFMA bound for the fused kernel - algorithmic MLP FLOPs per launch / average
                kernel time from HIP events on the launch stream, against 2.5 PFLOP/s
                (dense bf16, /opt/skills/guides/MI355X_MICROARCH.md)
  cpu_baseline: the CPU oracle (oracle/zest_oracle.py: the reference's op sequence in
                PyTorch-CPU fp32) timed on this box's host cores on a bounded sample of the
                same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "zest-nerf_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

IMAGE_PIXELS = 288 * 512          # the NSFF evaluation image (configs/*.txt), 144 chunks of 1024 rays
PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"

WORKLOADS = {
    # name: rays, samples, static volume+views, dynamic net, dyn volume
    "nsff_static_1024x128": dict(R=1024, S=128, use_mvs=False, scene_flow=False,
                                 note="BASELINE configs[1], use_mvs off (SURVEY 8(d) reading): C_in=90"),
    "nsff_static_mvs_1024x128": dict(R=1024, S=128, use_mvs=True, scene_flow=False,
                                     note="configs[1] with the K=8 encoding volume: C_in=130"),
    "nsff_zest_val_1024x128": dict(R=1024, S=128, use_mvs=True, scene_flow=True,
                                   note="BASELINE configs[2], inference: static + dynamic nets"),
    "zest_val_4096x192": dict(R=4096, S=192, use_mvs=True, scene_flow=True,
                              note="BASELINE configs[3] shape on one GPU"),
    # coherent rays: 1024 consecutive pixels (two image rows) of the 288 x 512 evaluation image, unjittered depths -
    # the chunk the reference's whole-image loops render (networks.py:660-673); the ones above draw one random
    # direction per ray, as a training batch does
    "nsff_static_mvs_grid_1024x128": dict(R=1024, S=128, use_mvs=True, scene_flow=False, ray_mode="grid",
                                          note="configs[1] + K=8 volume, rays of 1024 consecutive pixels"),
    "nsff_zest_val_grid_1024x128": dict(R=1024, S=128, use_mvs=True, scene_flow=True, ray_mode="grid",
                                        note="configs[2] inference, rays of 1024 consecutive pixels"),
    # parity-test cases (not bench lines): the other BASELINE configurations' geometry
    "llff_static_256x64": dict(R=256, S=64, use_mvs=False, scene_flow=False, H=640, W=960, V=3, focal=800.0,
                               note="BASELINE configs[0]: LLFF 640x960, static, use_mvs off"),
    "dtu_static_8192x128": dict(R=8192, S=128, use_mvs=True, scene_flow=False, H=512, W=640, V=3, focal=800.0,
                                note="BASELINE configs[4] geometry on one GPU: DTU 512x640, V=3, F=20"),
}


def build_workload(name, seed, device, rays=None, lively=True):
    """lively: He-scale weights (O(1) activations; colours saturate) as in the bench; False: nn.Linear's
    default scale (colours near 0.5, semi-transparent rays), the sensitive case for the PSNR check."""
    import zest_networks as networks
    import zest_synth as zs
    w = dict(WORKLOADS[name])
    R = rays or w["R"]
    V = w.get("V", 8)
    sc = zs.make_scene(seed, R, w["S"], H=w.get("H", 288), W=w.get("W", 512), V=V, V_dy=4, pad=24, vol_depth=128,
                       focal=w.get("focal", 400.0), static_volume=w["use_mvs"], dynamic=w["scene_flow"],
                       ray_mode=w.get("ray_mode", "random"), grid_start=100 * w.get("W", 512))
    feat_dim = 8 + 4 * V
    sf = w["scene_flow"]

    def net(P, Fd, static, use_mvs, sd):
        m = networks.MVSNeRF(D=8, W=256, input_ch_pts=P, output_ch=4, input_ch_views=27,
                             input_ch_feat=Fd, skips=[4], net_type="v0", sceneflow=sf, static=static,
                             use_mvs=use_mvs)
        lay = zs.mlp_layout(P, 27, Fd, sf, static, use_mvs)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in zs.fill_mlp_state(lay, sd, lively=lively).items()})
        return m.to(device)

    G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    d = SimpleNamespace(name=name, R=R, S=w["S"], cfg=w, sc=sc)
    d.net_s = net(63, feat_dim, True, w["use_mvs"], seed + 1)
    d.net_d = net(84, 24, False, True, seed + 2) if sf else None
    d.args = SimpleNamespace(netchunk=1024, feat_dim=feat_dim, feat_dim_dy=24, img_downscale=1.0,
                             use_color_volume=False, net_type="v0", precision=16, zest_maps_only=True)
    d.emb = (networks.Embedding(3, 10), networks.Embedding(4, 10), networks.Embedding(3, 4))
    d.t = {k: G(sc[k]) for k in ("rays_pts", "rays_ndc", "depth_candidates", "rays_dir")}
    d.cam = {"w2cs": G(sc["w2cs"]), "intrinsics": G(sc["intrinsics"])}
    d.vol_s = G(sc["vol_static"]) if w["use_mvs"] else None
    d.imgs = G(sc["imgs"]) if w["use_mvs"] else None
    d.vol_d = G(sc["vol_dynamic"]) if sf else None
    d.nb_imgs = G(sc["nb_imgs"]) if sf else None
    d.nb_cam = {"w2cs": G(sc["nb_w2cs"]), "intrinsics": G(sc["nb_intrinsics"])} if sf else None
    return d


def render_step(d):
    import zest_renderer as renderer
    return renderer.rendering(
        d.args, d.t["rays_pts"], d.t["rays_ndc"], d.t["depth_candidates"], d.t["rays_dir"],
        volume_feature_static=d.vol_s, volume_feature_dynamic=d.vol_d, imgs=d.imgs,
        neighbour_frames=d.nb_imgs, im_cam_mat=d.cam, nb_cam_mat=d.nb_cam, network_fn=d.net_s,
        network_fn_dy=d.net_d, embedding_pts=d.emb[0], embedding_xyzt=d.emb[1], embedding_dir=d.emb[2],
        ref_frame_idx=0.1, num_frames=24, scene_flow=d.cfg["scene_flow"], val=True)


def flops_per_ray_batch(d):
    """Algorithmic MLP FLOPs of one step: 2 * sum(in*out) per sample."""
    from oracle import zest_oracle as zo
    sf = d.cfg["scene_flow"]
    f = zo.mlp_flops_per_sample(zo.MlpSpec(63, 27, d.args.feat_dim, sf, True, d.cfg["use_mvs"]))
    if sf:
        f += zo.mlp_flops_per_sample(zo.MlpSpec(84, 27, 24, True, False, True))
    return f * d.R * d.S, f


def pmc_traffic(workload, mode="bf16"):
    """HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
    gfx950 correction: FETCH_SIZE counts half of a wide coalesced read) and archived by
    tools/pmc_traffic.py; None when no measurement exists for the workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(workload if mode == "bf16" else "%s@%s" % (workload, mode), {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def psnr_report(build_rgb, ref_rgb, seed=0):
    """BASELINE metric 'PSNR vs ref' (SURVEY 8(d)): PSNR of the build's colours against the
    reference's, and the north-star criterion |PSNR(build, target) - PSNR(ref, target)| against a
    synthetic target image (reference colours + N(0, 0.05^2) noise, clipped to [0, 1])."""
    b, r = build_rgb.double().cpu(), ref_rgb.double().cpu()
    g = torch.Generator().manual_seed(seed)
    target = (r + 0.05 * torch.randn(r.shape, generator=g, dtype=torch.float64)).clamp(0, 1)
    psnr = lambda x, y: float(10.0 * torch.log10(1.0 / (x - y).square().mean().clamp_min(1e-30)))
    return {"build_vs_ref_db": psnr(b, r), "delta_vs_target_db": abs(psnr(b, target) - psnr(r, target)),
            "criterion_db": 0.05, "sample": "%d rays of the workload; target = reference colours + N(0, 0.05^2)" % r.shape[0]}


def oracle_call(d, rays=256):
    """-> f() that runs the oracle (reference op sequence, PyTorch-CPU fp32) on the first `rays` rays of the
    workload and returns its result dict; and the ray count."""
    from oracle import zest_oracle as zo
    sc, sf = d.sc, d.cfg["scene_flow"]
    Rc = min(d.R, rays)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    st = lambda net: {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ns = zo.Net(st(d.net_s), zo.MlpSpec(63, 27, d.args.feat_dim, sf, True, d.cfg["use_mvs"]))
    nd = zo.Net(st(d.net_d), zo.MlpSpec(84, 27, 24, True, False, True)) if sf else None
    cams = (T(sc["w2cs"])[0], T(sc["intrinsics"])[0])
    nb = (T(sc["nb_w2cs"])[0], T(sc["nb_intrinsics"])[0]) if sf else None
    kw = dict(vol_static=T(sc["vol_static"])[0] if d.cfg["use_mvs"] else None,
              vol_dynamic=T(sc["vol_dynamic"])[0] if sf else None,
              imgs=T(sc["imgs"])[0] if d.cfg["use_mvs"] else None,
              nb_imgs=T(sc["nb_imgs"])[0] if sf else None, cams=cams, nb_cams=nb, scene_flow=sf,
              val=True, ref_frame_idx=0.1, num_frames=24, explicit=False)
    a = [T(sc[k])[0, :Rc] for k in ("rays_pts", "rays_ndc", "depth_candidates", "rays_dir")]

    def run():
        with torch.no_grad():
            return zo.rendering(*a, ns, nd, **kw)
    return run, Rc


def cpu_baseline(d, budget_s=15.0, build_ret=None):
    """Oracle (reference op sequence, PyTorch-CPU fp32, all host cores) on the first rays of the
    same workload; bounded to ~budget_s seconds.  With build_ret (the GPU result of the same
    batch) also returns the PSNR report of its colours against the oracle's."""
    sf = d.cfg["scene_flow"]
    run, Rc = oracle_call(d, 256)
    # the GPU box gives one GPU's job a 16-core share of the host; more threads only thrash
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, ncpu)))
    ref = run()                                              # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        run()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    out = {"value": Rc * n / el, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "%d calls of %d rays x %d samples of the same workload (oracle, torch-CPU fp32, "
                     "grid_sample/linear/cumprod op sequence of the reference)" % (n, Rc, d.S)}
    psnr = None
    if build_ret is not None:
        key = "rgb_map_ref" if sf else "rgb_map"
        psnr = psnr_report(build_ret[key][0, :Rc], ref[key].reshape(-1, 3))
    return out, psnr


MODES = {            # name: (args.precision, args.zest_dtype16, dense MFMA peak the mode is priced against)
    "bf16": (16, "bf16", PEAK_BF16_TFLOPS),
    "f16": (16, "f16", PEAK_BF16_TFLOPS),          # fp16 MFMA: the bf16 rate (MI355X_MICROARCH.md, Matrix cores)
    "f16x3": (32, "bf16", PEAK_BF16_TFLOPS),       # fp32 mode of the fused renderer: 3 fp16 MFMAs per product
}


def set_mode(d, mode):
    d.args.precision, d.args.zest_dtype16 = MODES[mode][0], MODES[mode][1]


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a, argv):
    """`python bench.py --gpus N` from a plain shell: start N ranks (one per GPU) with
    torch.distributed.run as CHILD processes and hand their exit code on.  This process has made
    no GPU call (nothing here touches torch.cuda) and never replaces itself with another program."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def _dry_maps(rank, R, step):
    """Stand-in for the rendered maps of one step in --dry runs (CPU, no HIP): row r of rank g holds
    its global ray index, so the gathered tensor can be checked."""
    base = torch.arange(rank * R, (rank + 1) * R, dtype=torch.float32)
    return base[:, None].repeat(1, 16) + 0.001 * step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="nsff_static_1024x128", choices=list(WORKLOADS))
    ap.add_argument("--mode", default="bf16", choices=list(MODES),
                    help="operand type of the headline line (BASELINE configs[1]: bf16)")
    ap.add_argument("--rays", type=int, default=None,
                    help="rays per GPU (weak scaling) / in total (strong scaling); default: the workload's")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU renders the workload's ray count; strong: the ray count is split over the GPUs")
    ap.add_argument("--gather", default="image", choices=["image", "step"],
                    help="N>1: all-gather the rendered maps once per image (default) or per call")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the extra f16 / f16x3 lines of the JSON")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--dry", action="store_true",
                    help="CPU rehearsal of the launcher + collective (gloo, stand-in maps, no HIP): tests only")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    if not a.dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the rendering path has no CPU fallback")
    dev = torch.device("cpu")
    if not a.dry:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != a.gpus:
            raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), a.gpus))

    total_rays = a.rays or WORKLOADS[a.workload]["R"]
    if a.scaling == "strong":
        if total_rays % world:
            raise SystemExit("bench.py: %d rays do not split over %d GPUs" % (total_rays, world))
        rays_per_gpu = total_rays // world
    else:
        rays_per_gpu = total_rays
    import zest_parallel
    if a.dry:
        d = SimpleNamespace(R=rays_per_gpu, S=WORKLOADS[a.workload]["S"], cfg=WORKLOADS[a.workload])
    else:
        import zest_hip
        zest_hip.lib()
        # The documented design: ONE scene - encoding volumes, source images, cameras and
        # MLP weights replicated on every rank - and one batch of rays_per_gpu x world rays cut into contiguous
        # blocks, rank g rendering block g (zest_parallel.shard_rays).  Every rank generates the batch from the same
        # seed; the per-image tensors are then broadcast from rank 0 over RCCL, once, outside the timed region,
        # as a whole-image loop would after building the volumes on one rank (zest_parallel.broadcast_scene).
        d = build_workload(a.workload, 1234, dev, rays_per_gpu * world)
        d.t = zest_parallel.shard_rays(d.t, dim=1)
        d.t = {k: v.contiguous() for k, v in d.t.items()}
        d.R = rays_per_gpu
        zest_parallel.broadcast_scene([d.vol_s, d.imgs, d.vol_d, d.nb_imgs])
        set_mode(d, a.mode)
    force = os.environ.get("ZEST_FORCE_COLLECTIVE") == "1" and dist is not None     # 1-rank rehearsal
    # Multi-GPU: rank g renders its own ray blocks; the packed per-ray maps reach every
    # rank by ONE RCCL all-gather, either per rendering call (--gather step) or, as the reference's
    # whole-image loops allow (networks.py:697-704 concatenates chunks per image), once per image
    # of IMAGE_PIXELS rays (--gather image, default): every rank keeps its chunks and pushes them
    # in one message, so the latency-bound collective is paid once per image, not per chunk.
    per_image = max(1, IMAGE_PIXELS // max(1, world * d.R))
    chunks, gathered, n_step = [], [], [0]

    def flush():
        if chunks:
            loc = chunks[0] if len(chunks) == 1 else torch.cat(chunks, 0)
            full = zest_parallel.gather_maps(loc, world * loc.shape[0], force=force)
            if a.dry:
                gathered.append(full)
            chunks.clear()

    def step():
        if a.dry:
            ret = {"zest_packed_maps": _dry_maps(rank, d.R, n_step[0])}
            n_step[0] += 1
        else:
            ret = render_step(d)
        if world > 1 or force:
            chunks.append(ret["zest_packed_maps"])
            if a.gather == "step" or len(chunks) >= per_image:
                flush()
        return ret

    def fence():
        flush()                     # rays rendered in the timed region are gathered inside it
        if dist is not None:
            dist.barrier()
        if not a.dry:
            torch.cuda.synchronize()

    def timed_loop():
        """W warm-up steps, then K steps between two fences; max over the ranks."""
        with torch.no_grad():
            for _ in range(a.warmup):
                step()
            fence()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            fence()
            el = time.perf_counter() - t0
        tmax = torch.tensor([el], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    def kernel_ms():
        """per-launch time of the fused kernel from HIP events on the launch stream (median of <= 50 launches)"""
        evs = []
        with torch.no_grad():
            for _ in range(min(a.steps, 50)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                render_step(d)
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
        return float(np.median([x.elapsed_time(y) for x, y in evs]))      # median: one pre-empted launch must not skew it

    el = timed_loop()
    if a.dry:
        # every gathered tensor: `world` equal blocks in rank order, each a run of whole chunks whose
        # rows carry the global ray indices rank * R .. rank * R + R - 1 in order
        ok = True
        for g in gathered:
            idx = torch.floor(g[:, 0]).long().view(world, -1, d.R)
            want = (torch.arange(world) * d.R)[:, None, None] + torch.arange(d.R)[None, None, :]
            ok = ok and bool((idx == want).all())
        if rank == 0:
            print(json.dumps({"dry": True, "n_gpus": world, "world_size": dist.get_world_size() if dist else 1,
                              "steps": a.steps, "warmup": a.warmup, "scaling": a.scaling, "rays_per_gpu": d.R,
                              "gathers": len(gathered), "gathered_rows": [int(g.shape[0]) for g in gathered[:4]],
                              "rows_in_order": bool(ok), "value": None}))
        if dist is not None:
            dist.destroy_process_group()
        return
    k_ms = kernel_ms()
    if rank == 0:
        flops, fps = flops_per_ray_batch(d)
        peak = MODES[a.mode][2]
        ach = flops / (k_ms * 1e-3) / 1e12
        coll = "none"
        if world > 1:
            coll = ("all_gather(packed per-ray maps) per rendering call" if a.gather == "step" else
                    "all_gather(packed per-ray maps) once per %d-ray image = every %d calls" % (IMAGE_PIXELS, per_image))
        out = {
            "metric": "rendered rays/sec (%d-ray x %d-sample batch per GPU)" % (d.R, d.S),
            "value": world * d.R * a.steps / el,
            "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": a.mode, "data": "synthetic",
            "config": {"workload": a.workload, "rays_per_gpu": d.R, "samples_per_ray": d.S,
                       "rays_total": world * d.R, "note": d.cfg["note"],
                       "path": "renderer.rendering -> zest_render_fused_fwd", "collective": coll,
                       "ranks": dist.get_world_size() if dist is not None else 1,
                       "backend": ("rccl" if dist is not None else "none")},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": ach / peak, "traffic": pmc_traffic(a.workload, a.mode),
                         "kernel": "fused_blocks_kernel (one launch: a range of whole rays per workgroup, rays finished in the kernel)",
                         "kernel_ms": k_ms, "flop_per_sample": fps,
                         "note": "achieved = algorithmic MLP FLOPs of one launch / HIP-event time of the "
                                 "launch on its stream; traffic = HBM bytes per launch from the "
                                 "rocprofv3 PMC passes archived in profiles/ (null if none for this workload)"},
        }
    if world == 1 and not a.no_modes:
        # the other operand types of the same kernel on the same workload: fp16 (configs[4]) and the
        # fp32-tolerance mode (split fp16, north-star 1e-4 abs / 1e-3 rel), same protocol
        modes = {}
        for m in MODES:
            if m == a.mode:
                continue
            set_mode(d, m)
            el_m = timed_loop()
            km = kernel_ms()
            fl, _ = flops_per_ray_batch(d)
            ach_m = fl / (km * 1e-3) / 1e12
            modes[m] = {"value": d.R * a.steps / el_m, "unit": "rays/s", "ms_per_step": el_m / a.steps * 1e3,
                        "kernel_ms": km, "roofline": {"bound": "mfma", "achieved": ach_m, "peak": MODES[m][2],
                                                      "unit": "TFLOP/s", "frac": ach_m / MODES[m][2],
                                                      "traffic": pmc_traffic(a.workload, m)}}
        if "f16x3" in modes:
            modes["f16x3"]["note"] = ("fp32 mode of the fused renderer: every product is 3 fp16 MFMAs (hi*hi + hi*lo + lo*hi); "
                                      "achieved counts the ALGORITHMIC FLOPs once, so frac <= 1/3 by construction; "
                                      "executed MFMA work = 3x; against the fp32 MFMA peak (157.3 TFLOP/s) the same "
                                      "number is %.2fx" % (modes["f16x3"]["roofline"]["achieved"] / 157.3))
        set_mode(d, a.mode)
        out["modes"] = modes
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            with torch.no_grad():
                build_ret = render_step(d)
            torch.cuda.synchronize()
            out["cpu_baseline"], out["psnr"] = cpu_baseline(d, a.cpu_budget, build_ret)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
