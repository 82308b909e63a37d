"""HIP-event time of each kernel of the bf16 training path on M = 131072 samples (one MLP call of the 1024 x 128 step):
forward with stash, backward data kernel, finishing kernel, weight-gradient kernel (the `stages` bits of zest_mlp_train16_bwd).
    python tools/time_train16.py [case] [M]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "zest-nerf_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import json
import numpy as np, torch
import golden_cases as gc
from test_hip_ops import G, _mlp_setup
case = sys.argv[1] if len(sys.argv) > 1 else "mlp_dynamic_mvs24"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
zh, inp, desc, tab = _mlp_setup(case)
x = torch.rand(M, desc.in_ch, device="cuda:0") * 2 - 1
gw = torch.randn(M, desc.out_ch, device="cuda:0")
pf, pb = zh.mlp_pack(desc, zh.PREC_BF16, tab), zh.mlp_train16_pack_bwd(desc, tab)
out, stash = zh.mlp_train16_fwd(desc, pf, x)
_, _, work = zh.mlp_train16_bwd(desc, pb, tab, x, stash, out, gw)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {"case": case, "M": M, "fwd_stash_us": timed(lambda: zh.mlp_train16_fwd(desc, pf, x))}
for name, bits in (("data_us", 1), ("finish_us", 2), ("weights_us", 4), ("all_us", 7)):
    res[name] = timed(lambda: zh.mlp_train16_bwd(desc, pb, tab, x, stash, out, gw, stages=bits, work=work))
print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}))
