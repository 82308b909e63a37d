import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "zest-nerf_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
import golden_cases as gc
from test_hip_render import call_rendering
case = "render_zest_nomvsdy"
gold = gc.load_golden(case)
for tag, kw in (("f16", dict(precision=16, dtype16="f16")), ("bf16", dict(precision=16, dtype16="bf16"))):
    f = call_rendering(case, maps_only=True, **kw)
    p = call_rendering(case, maps_only=False, **kw)
    for k in ("rgb_map_ref", "depth_map_ref", "rgb_map_ref_dy", "weights_map_dd"):
        a = f[k][0].cpu().numpy(); b = p[k][0].cpu().numpy(); g = gold[k]
        print(tag, k, "fused-gold", np.abs(a-g).max(), "perop-gold", np.abs(b-g).max())
        if tag == "f16" and k == "weights_map_dd":
            print(np.round(a[:16],3)); print(np.round(g[:16],3))
