"""Kernel times of the HIP regularisation net (csrc/costreg.hip) at a builder geometry (default NSFF: 128 x 120 x 176).
    python tools/bench_costreg.py [--passes 3] [--shape 128,120,176] [--library]"""
import argparse
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench  # noqa: F401
import zest_networks as networks
ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--shape", default="128,120,176")
ap.add_argument("--library", action="store_true", help="the module's library-convolution path instead")
a = ap.parse_args()
D, H, W = (int(v) for v in a.shape.split(","))
torch.manual_seed(0)
net = networks.CostRegNet(41).cuda().train()
cost = torch.randn(D, H, W, 48, device="cuda")
cost[..., 41:] = 0
cf = cost[..., :41].permute(3, 0, 1, 2)[None].contiguous()
amp = torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.library and a.passes == 1)
run = (lambda: net(cf)[0]) if a.library else (lambda: net.forward_hip(cost, passes=a.passes))
with torch.no_grad(), amp:
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            run()
        torch.cuda.synchronize()
print("%s passes %d  %dx%dx%d: %.3f ms per call" % ("library" if a.library else "hip", a.passes, D, H, W, e0.elapsed_time(e1) / 10))
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
for e in rows[:16]:
    if e.device_time_total > 0:
        print("%8.1f us  x%-3d %s" % (e.device_time_total / 5, e.count // 5, e.key[:110]))
