"""Time and profile one MVSNet volume builder at the NSFF geometry (288 x 512, 3 views, 128 planes, pad 24) under
bf16 autocast: the library convolutions around the HIP plane sweep (6.6 ms of kernels per builder, the sweep 0.16).
    python tools/prof_mvsnet.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "zest-nerf_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import torch
import zest_networks as networks
import test_generators as tg
x = tg._batch(7, H=288, W=512)
net = networks.MVSNet().cuda().eval()
imgs, proj, nf = x["images"][:, :-1], x["proj_mats"][:, :-1], x["near_fars"][0, 0]
def run(n=5):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(2): net(imgs, proj, nf, pad=24)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): net(imgs, proj, nf, pad=24)
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("default ms", round(run(), 2))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        net(imgs, proj, nf, pad=24)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
