"""One training-shaped step of the WHOLE generator (DyMVSNeRF_G.forward: both volume builders under autograd + ray sampling +
train-mode rendering, then a loss and backward into the builders and both MLPs) at the NSFF geometry: where a real
training step spends its time once the rendering path is 5.7 ms.
    python tools/bench_generator_train.py [--precision 16] [--steps 5]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench  # noqa: F401
import test_generators as tg
ap = argparse.ArgumentParser()
ap.add_argument("--precision", type=int, default=16)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--library-costreg", action="store_true", help="MVSNet.zest_hip_costreg_train = False: the regularisation nets through the library under autograd (the product default in --precision 16 is their forward on the HIP kernels, zest_autograd.CostRegFn)")
a = ap.parse_args()
torch.backends.cudnn.benchmark = True
x = tg._batch(7, H=288, W=512)
args = tg._args(precision=a.precision, N_samples=128, pad=24, batch_size=1024, chunk=1024, num_extra_samples=0, use_motion_mask=False)
gen = tg._generator(args, train_builders=True).train()
if a.library_costreg:
    args.zest_hip_costreg_train = False


def step():
    gen.zero_grad(set_to_none=True)
    ret = gen(x, step=0)
    loss = sum(v.float().square().mean() for k, v in ret.items() if torch.is_tensor(v) and v.requires_grad and v.dtype.is_floating_point)
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
print("generator training step: %.1f ms" % ((time.perf_counter() - t0) / a.steps * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print("device time %.1f ms" % (tot / 1e3))
for e in rows[:14]:
    print("%8.1f us %5.1f%% x%-3d %s" % (e.device_time_total, 100 * e.device_time_total / tot, e.count, e.key[:100]))
