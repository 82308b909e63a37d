"""Kernel breakdown of the two volume builders of one image (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench  # noqa: F401
import test_generators as tg
x = tg._batch(7, H=288, W=512)
gen = tg._generator(tg._args(pad=24, N_samples=128))
with torch.no_grad():
    for _ in range(4):
        gen._scene(x, bn_batch_stats=True)
    torch.cuda.synchronize()
