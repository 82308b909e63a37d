"""Kernel breakdown of the two volume builders of one image, steady state (after the library's convolution autotuning):
torch.profiler's device-side kernel table over N calls of DyMVSNeRF_G._scene.
    python tools/prof_builder.py [--serial] [--calls 5]"""
import argparse
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench  # noqa: F401
import test_generators as tg
ap = argparse.ArgumentParser()
ap.add_argument("--serial", action="store_true")
ap.add_argument("--calls", type=int, default=5)
a = ap.parse_args()
x = tg._batch(7, H=288, W=512)
gen = tg._generator(tg._args(pad=24, N_samples=128))
gen.args.zest_overlap_builders = not a.serial
with torch.no_grad():
    for _ in range(4):
        gen._scene(x, bn_batch_stats=True)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(a.calls):
            gen._scene(x, bn_batch_stats=True)
        torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print("device time per call: %.2f ms over %d kernels/call" % (tot / a.calls / 1e3, sum(e.count for e in rows) // a.calls))
for e in rows[:40]:
    print("%8.1f us/call %5.1f%%  x%-4d %s" % (e.device_time_total / a.calls, 100 * e.device_time_total / tot,
                                              e.count // a.calls, e.key[:120]))
