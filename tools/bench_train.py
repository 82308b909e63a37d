#!/usr/bin/env python3
"""One training-shaped step of the rendering path on one MI355X: train-mode rendering()
(all 28 outputs) + a loss over the differentiable outputs + backward into both MLPs and both
encoding volumes, fp32.  Next to it the same step on the CPU oracle (reference op sequence +
torch autograd) on a bounded ray subset.

    python tools/bench_train.py [--rays 1024] [--samples 128] [--frames 3|5]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import zest_networks as networks
import zest_renderer as renderer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=3, choices=[3, 5])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu-rays", type=int, default=64)
    ap.add_argument("--precision", type=int, default=16, choices=[16, 32],
                    help="16: bf16 MFMA training kernels (MlpFn16); 32: fp32 parity path (rocBLAS sgemm)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    d = bench.build_workload("nsff_zest_val_1024x128", 5, dev, a.rays)
    d.args.precision, d.args.zest_maps_only = a.precision, False
    vol_s, vol_d = d.vol_s.clone().requires_grad_(True), d.vol_d.clone().requires_grad_(True)
    params = list(d.net_s.parameters()) + list(d.net_d.parameters())

    def step():
        for p in params:
            p.grad = None
        vol_s.grad = vol_d.grad = None
        ret = renderer.rendering(
            d.args, d.t["rays_pts"], d.t["rays_ndc"], d.t["depth_candidates"], d.t["rays_dir"],
            volume_feature_static=vol_s, volume_feature_dynamic=vol_d, imgs=d.imgs, neighbour_frames=d.nb_imgs,
            im_cam_mat=d.cam, nb_cam_mat=d.nb_cam, network_fn=d.net_s, network_fn_dy=d.net_d,
            embedding_pts=d.emb[0], embedding_xyzt=d.emb[1], embedding_dir=d.emb[2], ref_frame_idx=0.1,
            num_frames=24, scene_flow=True, val=False, chain_5frames=(a.frames == 5), raw_noise_std=0)
        loss = sum(v.square().mean() for k, v in ret.items()
                   if v is not None and v.requires_grad and k not in ("raw_rgba", "input_feat"))
        loss.backward()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    out = {"op": "train step (rendering fwd + bwd, %d-frame ZeST, %s MLP)" % (a.frames, "bf16 MFMA" if a.precision == 16 else "fp32 rocBLAS"), "rays": d.R, "samples": d.S,
           "ms_per_step": dt * 1e3, "rays_per_s": d.R / dt,
           "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}
    if a.cpu_rays <= 0:
        print(json.dumps(out))
        return
    # CPU: oracle + autograd on a subset of the same rays
    from oracle import zest_oracle as zo
    sc, Rc = d.sc, a.cpu_rays
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    st = lambda net: {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    ns = zo.Net(st(d.net_s), zo.MlpSpec(63, 27, 40, True, True, True))
    nd = zo.Net(st(d.net_d), zo.MlpSpec(84, 27, 24, True, False, True))
    vs, vd = T(sc["vol_static"])[0].requires_grad_(True), T(sc["vol_dynamic"])[0].requires_grad_(True)
    ncpu = min(16, len(os.sched_getaffinity(0)))
    torch.set_num_threads(ncpu)

    def cpu_step():
        ret = zo.rendering(*[T(sc[k])[0, :Rc] for k in ("rays_pts", "rays_ndc", "depth_candidates", "rays_dir")],
                           ns, nd, vol_static=vs, vol_dynamic=vd, imgs=T(sc["imgs"])[0], nb_imgs=T(sc["nb_imgs"])[0],
                           cams=(T(sc["w2cs"])[0], T(sc["intrinsics"])[0]),
                           nb_cams=(T(sc["nb_w2cs"])[0], T(sc["nb_intrinsics"])[0]), scene_flow=True, val=False,
                           chain_5frames=(a.frames == 5), ref_frame_idx=0.1, num_frames=24, explicit=False)
        sum(v.square().mean() for k, v in ret.items()
            if v is not None and v.requires_grad and k not in ("raw_rgba", "input_feat")).backward()
    cpu_step()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 10:
        cpu_step()
        n += 1
    dtc = (time.perf_counter() - t0) / n
    out["cpu_oracle"] = {"rays": Rc, "threads": ncpu, "ms_per_step": dtc * 1e3, "rays_per_s": Rc / dtc}
    out["speedup_rays_per_s"] = out["rays_per_s"] / out["cpu_oracle"]["rays_per_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
