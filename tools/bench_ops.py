#!/usr/bin/env python3
"""Per-operator timings on one MI355X against each operator's own roofline (HBM bytes or
MLP FLOPs per call), at the headline shape 1024 rays x 128 samples, NSFF geometry, V=8.

    python tools/bench_ops.py            # prints one JSON object per operator
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import zest_hip
import zest_utils as zutils

HBM_PEAK, F32_MFMA_PEAK, BF16_PEAK = 8000.0, 157.3, 2500.0      # GB/s, TFLOP/s, TFLOP/s


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def report(name, t, bytes_=None, flops=None, peak=None, unit=None):
    ach = (bytes_ / t / 1e9) if bytes_ else (flops / t / 1e12)
    print(json.dumps({"op": name, "us": round(t * 1e6, 1), "achieved": round(ach, 1), "unit": unit,
                      "peak": peak, "frac": round(ach / peak, 3)}))


def main():
    dev = torch.device("cuda:0")
    d = bench.build_workload("nsff_static_mvs_1024x128", 3, dev)
    R, S, V = d.R, d.S, 8
    M = R * S
    ndc, pts, z, dirs = (d.t[k][0] for k in ("rays_ndc", "rays_pts", "depth_candidates", "rays_dir"))
    vcl, icl = zutils.volume_channels_last(d.vol_s), zutils.images_channels_last(d.imgs)
    w2cs, intr = d.cam["w2cs"][0], d.cam["intrinsics"][0]
    with torch.no_grad():
        # ray sampling: 8 B in per ray + 4 B jitter per sample, 28 B out per sample
        xs, ys = torch.rand(R, device=dev) * 500, torch.rand(R, device=dev) * 280
        tr = torch.rand(R, S, device=dev)
        nf, eye = torch.tensor([2.0, 6.0], device=dev), torch.eye(4, device=dev)
        t = timeit(lambda: zest_hip.build_rays(xs, ys, tr, S, intr[-1], eye, w2cs[0], intr[0], nf, nf, 24, 512, 288))
        report("build_rays", t, bytes_=M * 32, peak=HBM_PEAK, unit="GB/s")
        raw = torch.randn(R, S, 4, device=dev)
        t = timeit(lambda: zest_hip.composite(raw, z, dirs))
        report("composite", t, bytes_=M * (16 + 4 + 8), peak=HBM_PEAK, unit="GB/s")
        t = timeit(lambda: zest_hip.volume_lookup(vcl, ndc))
        report("volume_lookup (gathered 256 B/sample)", t, bytes_=M * (256 + 12 + 32), peak=HBM_PEAK, unit="GB/s")
        t = timeit(lambda: zest_hip.color_lookup(icl, w2cs, intr, pts))
        report("color_lookup V=8 (gathered 512 B/sample)", t, bytes_=M * (64 * V + 12 + 16 * V), peak=HBM_PEAK,
               unit="GB/s")
        t = timeit(lambda: zest_hip.encode(ndc, pts, dirs, None, vcl, icl, w2cs, intr))
        report("encode -> x[M,130]", t, bytes_=M * (130 * 4 + 24 + 256 + 64 * V), peak=HBM_PEAK, unit="GB/s")
        x = zest_hip.encode(ndc, pts, dirs, None, vcl, icl, w2cs, intr)
        desc = d.net_s.desc()
        from oracle import zest_oracle as zo
        fl = zo.mlp_flops_per_sample(zo.MlpSpec(63, 27, 40, False, True, True)) * M
        for prec, name, peak in ((zest_hip.PREC_F32, "mlp fp32 (32x32x2 f32 MFMA)", F32_MFMA_PEAK),
                                 (zest_hip.PREC_BF16, "mlp bf16 standalone (x rows in HBM -> LDS-ring engine)", BF16_PEAK)):
            packed = d.net_s.packed(prec)
            t = timeit(lambda: zest_hip.mlp_fwd(desc, prec, packed, x), n=20)
            report(name, t, flops=fl, peak=peak, unit="TFLOP/s")
        # plane-sweep cost volume at the NSFF geometry: 3 views, 72 x 128 features, pad 24, 128 planes
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import golden_cases as gc
        inp = gc.cost_inputs(5, V=3, H=72, W=128, D=128, pad=24, spread=0.15)
        g = lambda k: torch.from_numpy(inp[k]).to(dev)
        feats, proj, depth = g("feats")[0], g("proj_mats")[0, 1:], g("depth_values")[0]
        imgs_lr = torch.nn.functional.interpolate(g("imgs")[0], (72, 128), mode="bilinear", align_corners=False)
        fcl, icl3 = zest_hip.nchw_to_nhwc(feats), zest_hip.images_to_cl(imgs_lr)
        nvox = 128 * (72 + 48) * (128 + 48)
        img_feat = torch.empty(41, 128, 120, 176, device=dev)
        masks = torch.empty(3, 128, 120, 176, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def sweep():
            rc = zest_hip.lib().zest_volume_cost_fwd(fcl.data_ptr(), icl3.data_ptr(), proj.data_ptr(), depth.data_ptr(),
                                                      3, 32, 128, 72, 128, 24, img_feat.data_ptr(), masks.data_ptr(), st)
            assert rc == 0
        t = timeit(sweep, n=20)
        report("volume_cost plane sweep (writes 44 ch x 2.7 M voxels)", t, bytes_=nvox * 44 * 4, peak=HBM_PEAK, unit="GB/s")


if __name__ == "__main__":
    main()
