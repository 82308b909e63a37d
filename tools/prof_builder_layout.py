"""Volume builder (MVSNet at the NSFF geometry, bf16 autocast, no graph): default memory format against channels-last
(2-D feature pyramid: torch.channels_last; 3-D regulariser: torch.channels_last_3d).  python tools/prof_builder_layout.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "zest-nerf_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import torch
import zest_networks as networks
import test_generators as tg
x = tg._batch(7, H=288, W=512)
imgs, proj, nf = x["images"][:, :-1], x["proj_mats"][:, :-1], x["near_fars"][0, 0]

def run(net, n=5):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(3): out = net(imgs, proj, nf, pad=24)[0]
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): out = net(imgs, proj, nf, pad=24)[0]
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, out.float()

torch.manual_seed(0)
net = networks.MVSNet().cuda().eval()
t0, o0 = run(net)
res = {"default_ms": round(t0, 2)}
for name, fmt3, fmt2 in (("cost_reg channels_last_3d", True, False), ("both channels-last", True, True)):
    n2 = networks.MVSNet().cuda().eval()
    n2.load_state_dict(net.state_dict())
    if fmt3:
        n2.cost_reg_2 = n2.cost_reg_2.to(memory_format=torch.channels_last_3d)
    if fmt2:
        n2.feature = n2.feature.to(memory_format=torch.channels_last)
    t, o = run(n2)
    res[name] = {"ms": round(t, 2), "max_abs_diff_vs_default": float((o - o0).abs().max()), "scale": float(o0.abs().max())}
print(json.dumps(res))
