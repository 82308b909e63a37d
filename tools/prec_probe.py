#!/usr/bin/env python3
"""GPU probe: error of every operand type of the MLP engine against the reference fixtures, and
their speed on the bench workloads.  `python tools/prec_probe.py [--time]`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "zest-nerf_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch

import golden_cases as gc
import zest_hip as zh
from test_hip_render import call_rendering
from test_hip_ops import G


def worst(got, want):
    g, w = got.double().cpu().numpy(), np.asarray(want, np.float64)
    err = np.abs(g - w)
    return float(err.max()), float((err / (1e-4 + 1e-3 * np.abs(w))).max())


def main():
    out = {}
    for case in ("render_static_mvs", "render_static_nomvs", "render_static_white", "render_zest_val", "render_zest_nomvsdy"):
        gold = gc.load_golden(case)
        for tag, kw in (("x3", dict(precision=32)), ("f16", dict(precision=16, dtype16="f16")),
                        ("bf16", dict(precision=16, dtype16="bf16"))):
            ret = call_rendering(case, maps_only=True, **kw)
            for k in ("rgb_map", "depth_map", "rgb_map_ref", "depth_map_ref", "rgb_map_ref_dy", "depth_map_ref_dy", "weights_map_dd"):
                if k in gold and k in ret:
                    a, r = worst(ret[k][0], gold[k])
                    out["%s/%s/%s" % (case, tag, k)] = "abs %.2e  tol-units %.2f" % (a, r)
    # standalone MLP
    import test_hip_ops as tho
    for case in tho.MLP_CASES:
        _, inp, desc, tab = tho._mlp_setup(case)
        gold = gc.load_golden(case)
        for prec in (zh.PREC_F32, zh.PREC_F16X3, zh.PREC_F16, zh.PREC_BF16):
            y = zh.mlp_fwd(desc, prec, zh.mlp_pack(desc, prec, tab), G(inp["x"])[0])
            a, r = worst(y, gold["y"])
            out["%s/%s" % (case, zh.PREC_NAMES[prec])] = "abs %.2e  tol-units %.2f  (|y|max %.2f)" % (a, r, np.abs(gold["y"]).max())
    for k, v in out.items():
        print("%-60s %s" % (k, v))
    if "--time" in sys.argv:
        import bench
        for name in ("nsff_static_1024x128", "nsff_static_mvs_1024x128", "nsff_zest_val_1024x128"):
            d = bench.build_workload(name, 1234, torch.device("cuda:0"))
            for tag, prec, d16 in (("bf16", 16, "bf16"), ("f16", 16, "f16"), ("f16x3", 32, "bf16")):
                d.args.precision, d.args.zest_dtype16 = prec, d16
                with torch.no_grad():
                    for _ in range(20):
                        bench.render_step(d)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(100):
                        bench.render_step(d)
                    e1.record()
                    torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 100
                fl, _ = bench.flops_per_ray_batch(d)
                print(json.dumps({"workload": name, "mode": tag, "ms": ms, "rays_per_s": d.R / ms * 1e3,
                                  "algorithmic_tflops": fl / ms / 1e9}))


if __name__ == "__main__":
    main()
