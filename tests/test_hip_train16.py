"""GPU: the bf16 training path of the MLP on the MFMA engine (zest_mlp_train16_*): forward with activation
stash, backward data / modulation / weight-gradient kernels, against the oracle's autograd in fp64.
Tolerances are those of bf16 operands with fp32 accumulation through 10 layers: a few per cent of each
gradient tensor's norm (the fp32 rocBLAS path, tests/test_hip_backward.py, stays the 1e-3 parity mode)."""
import numpy as np
import pytest
import torch

import golden_cases as gc
import oracle_run as orun
from test_hip_ops import G, _mlp_setup

pytestmark = pytest.mark.gpu
V0_CASES = ["mlp_static_mvs20", "mlp_static_nomvs", "mlp_static_sf_mvs40", "mlp_dynamic_mvs24", "mlp_dynamic_nomvs"]


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-30))


class _Bf16Operands:
    """The oracle's MLP with every GEMM operand rounded to bf16 (straight-through: identity in backward),
    fp32/fp64 accumulation and epilogues - the arithmetic of the engine.  ReLU is not differentiable at 0:
    a unit whose pre-activation is within bf16 noise of 0 is 'on' in one evaluation and 'off' in another, and
    each such flip moves a layer's gradient by ~1/sqrt(active units), i.e. ~10 % of its norm per 1 % of flipped
    units.  The gradient check therefore differentiates the SAME rounded network (same masks) instead of the
    fp64 one; the forward check below still compares with the un-rounded oracle."""

    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.real = F, F.linear
        r = lambda t: t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()
        F.linear = lambda x, w, b=None: self.real(r(x), r(w), b)
        return self

    def __exit__(self, *a):
        self.F.linear = self.real


def oracle_grads(inp, x, gw, bf16_operands=True):
    from oracle import zest_oracle as zo
    state = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in inp["state"].items()}
    xt = torch.from_numpy(x).double().requires_grad_(True)
    spec = orun.spec_of(inp["P"], inp["Fd"], inp["sceneflow"], inp["static"], inp["use_mvs"], inp["net_type"])
    if bf16_operands:
        with _Bf16Operands():
            y = zo.mlp_forward(state, xt, spec)
    else:
        y = zo.mlp_forward(state, xt, spec)
    (y * torch.from_numpy(gw).double()).sum().backward()
    return y.detach().numpy(), xt.grad.numpy(), {k: (v.grad.numpy() if v.grad is not None else None) for k, v in state.items()}


@pytest.mark.parametrize("M", [64, 300])
@pytest.mark.parametrize("case", V0_CASES)
def test_train16_forward_and_backward_match_the_oracle(hip, case, M):
    zh, inp, desc, tab = _mlp_setup(case)
    g = gc.zs.rng(900 + M)
    x = g.uniform(-1, 1, size=(M, desc.in_ch)).astype(np.float32)
    gw = g.standard_normal((M, desc.out_ch)).astype(np.float32)
    y_ref, gx_ref, gp_ref = oracle_grads(inp, x, gw)
    y64, _, _ = oracle_grads(inp, x, gw, bf16_operands=False)
    out, stash = zh.mlp_train16_fwd(desc, zh.mlp_pack(desc, zh.PREC_BF16, tab), G(x))
    assert rel(out.cpu().numpy(), y64) < 2e-2 and rel(out.cpu().numpy(), y_ref) < 2e-3
    g_x, grads, _ = zh.mlp_train16_bwd(desc, zh.mlp_train16_pack_bwd(desc, tab), tab, G(x), stash, out, G(gw))
    torch.cuda.synchronize()
    P, F = inp["P"], (inp["Fd"] if inp["use_mvs"] else 0)
    gx = g_x.cpu().numpy()
    assert rel(gx[:, :P], gx_ref[:, :P]) < 4e-2, "g_x points %.3g" % rel(gx[:, :P], gx_ref[:, :P])
    if F:
        assert rel(gx[:, P:P + F], gx_ref[:, P:P + F]) < 4e-2, "g_x features %.3g" % rel(gx[:, P:P + F], gx_ref[:, P:P + F])
    assert np.abs(gx[:, P + F:]).max() == 0.0                                   # directions are data
    names = {}
    for name, slot in zh._PARAM_SLOTS:
        names[slot] = name
    names[13] = {zh.HEAD_BLEND: "w_linear", zh.HEAD_DYNAMIC: "sf_linear"}.get(desc.head)
    names[14] = "prob_linear" if desc.head == zh.HEAD_DYNAMIC else None
    checked = 0
    for slot in range(zh.P_COUNT):
        if tab[2 * slot] is None or names.get(slot) is None:
            continue
        for j, kind in enumerate(("weight", "bias")):
            want = gp_ref["nerf.%s.%s" % (names[slot], kind)]
            got = grads[2 * slot + j].cpu().numpy()
            assert got.shape == want.shape
            e = rel(got, want)
            assert e < 5e-2, "%s.%s: relative L2 error %.3g" % (names[slot], kind, e)
            checked += 1
    assert checked >= 24
