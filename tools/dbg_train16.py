import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "zest-nerf_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
import golden_cases as gc
from test_hip_ops import G, _mlp_setup
from test_hip_train16 import oracle_grads, rel
case = sys.argv[1] if len(sys.argv) > 1 else "mlp_static_mvs20"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 64
zh, inp, desc, tab = _mlp_setup(case)
g = gc.zs.rng(900 + M)
x = g.uniform(-1, 1, size=(M, desc.in_ch)).astype(np.float32)
gw = g.standard_normal((M, desc.out_ch)).astype(np.float32)
if len(sys.argv) > 3:      # only some output columns carry gradient
    keep = [int(c) for c in sys.argv[3].split(",")]
    m = np.zeros_like(gw); m[:, keep] = 1; gw = gw * m
y_ref, gx_ref, gp_ref = oracle_grads(inp, x, gw)
out, stash = zh.mlp_train16_fwd(desc, zh.mlp_pack(desc, zh.PREC_BF16, tab), G(x))
print("fwd", rel(out.cpu().numpy(), y_ref))
g_x, grads, _ = zh.mlp_train16_bwd(desc, zh.mlp_train16_pack_bwd(desc, tab), tab, G(x), stash, out, G(gw))
torch.cuda.synchronize()
P, F = inp["P"], (inp["Fd"] if inp["use_mvs"] else 0)
gx = g_x.cpu().numpy()
print("g_x pts", rel(gx[:, :P], gx_ref[:, :P]), "feat", rel(gx[:, P:P+F], gx_ref[:, P:P+F]) if F else None)
names = {slot: name for name, slot in zh._PARAM_SLOTS}
names[13] = {zh.HEAD_BLEND: "w_linear", zh.HEAD_DYNAMIC: "sf_linear"}.get(desc.head)
names[14] = "prob_linear" if desc.head == zh.HEAD_DYNAMIC else None
for slot in range(zh.P_COUNT):
    if tab[2*slot] is None or names.get(slot) is None: continue
    for j, kind in enumerate(("weight", "bias")):
        want = gp_ref["nerf.%s.%s" % (names[slot], kind)]
        got = grads[2*slot+j].cpu().numpy()
        print("%-22s %-6s rel %.4f  |want| %.3g |got| %.3g" % (names[slot], kind, rel(got, want), np.abs(want).max(), np.abs(got).max()))
