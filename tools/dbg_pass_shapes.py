"""Diagnostic for tests/test_hip_precision.py::test_pass_shapes_agree: do the pass shapes of the fused renderer agree
launch after launch - before and after a volume builder has been recorded as a HIP graph in the same process?
    python tools/dbg_pass_shapes.py [R] [S] [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "zest-nerf_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import torch
import golden_cases as gc
import zest_hip
from test_hip_render import render_scene
R, S = int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 192
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
sc = gc.render_inputs(2100 + R + S, R=R, S=S, V=3, use_mvs=True, scene_flow=True, use_mvs_dy=True)
kw = dict(precision=16, dtype16="bf16")


def sweep(tag):
    base, bad = None, 0
    for rep in range(N):
        for shape in ("dense", "ranges", None):
            zest_hip.set_fused_passes(shape)
            v = render_scene(sc, dict(val=True), maps_only=True, **kw)["zest_packed_maps"].clone()
            if base is None:
                base = v
            elif not torch.equal(v, base):
                ne = v != base
                bad += 1
                if bad <= 8:
                    print(tag, "rep", rep, shape, "rows", ne.any(1).nonzero().flatten().tolist()[:10], "cols",
                          ne.any(0).nonzero().flatten().tolist(), "max", float((v - base).abs().max()),
                          "nan", int(torch.isnan(v).sum()))
    zest_hip.set_fused_passes(None)
    print(tag, "launches", 3 * N, "mismatching", bad)


if len(sys.argv) > 4 and sys.argv[4] == "all":
    # every case of the test, both modes
    for (R, S) in [(40, 96), (40, 192), (8, 250), (9, 300), (300, 192), (1100, 160)]:
        sc = gc.render_inputs(2100 + R + S, R=R, S=S, V=3, use_mvs=True, scene_flow=True, use_mvs_dy=True)
        for kw in (dict(precision=16, dtype16="bf16"), dict(precision=32)):
            sweep("%dx%d/%s" % (R, S, "bf16" if kw["precision"] == 16 else "f16x3"))
    sys.exit(0)
sweep("before-graph")
import test_generators as tg
gen = tg._generator(tg._args(chunk=256, precision=16))
x = tg._batch(91)
with torch.no_grad():
    for _ in range(3):
        gen._scene(x, bn_batch_stats=True)
torch.cuda.synchronize()
print("graphs recorded:", len(gen.__dict__.get("_zest_builder_graphs", {})))
sweep("after-graph")
