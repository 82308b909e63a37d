"""CPU: the oracle's autograd reproduces the gradient digests of the reference's own
training-mode backward (tests/golden/grad_*.npz from tools/gen_golden.py), so the oracle is a
pinned checker for the HIP backward kernels too."""
import numpy as np
import pytest

import golden_cases as gc
import oracle_run


@pytest.mark.parametrize("case", ["grad_zest_5f", "grad_static", "grad_static_timecodes", "grad_static_d5w128"])
def test_oracle_gradients_match_reference(case):
    gold = gc.load_golden(case)
    loss, grads = oracle_run.oracle_render_grads(case)
    assert abs(loss - gold["__loss__"][0]) <= 1e-4 * max(1.0, abs(gold["__loss__"][0]))
    names = [k for k in gold if not k.startswith("__")]
    assert sorted(names) == sorted(k.replace(".nerf.", ".nerf.") for k in grads)
    for k in names:
        d = gc.grad_digest(grads[k])
        scale = max(gold[k][1], 1e-6)                       # L2 norm of the reference gradient
        assert np.all(np.abs(d - gold[k]) <= 2e-3 * scale * max(1.0, np.sqrt(grads[k].size) / 50)), \
            "%s: digest %s vs reference %s" % (k, d, gold[k])


def test_oracle_volume_cost_gradient_is_a_valid_reference():
    """The oracle's plane sweep (restatement of MVSNet.build_volume_cost, reference networks.py:1077-1140)
    is the gradient reference of zest_volume_cost_bwd: check its autograd against finite differences in
    fp64 at V = 3 with a padding ring (gradcheck), so the GPU test compares against a verified gradient."""
    import torch
    import golden_cases as gc
    from oracle import zest_oracle as zo
    inp = gc.cost_inputs(70, V=3, H=5, W=6, D=3, pad=2)
    t = lambda k: torch.from_numpy(inp[k])[0].double()
    imgs, proj, depth = t("imgs"), t("proj_mats"), t("depth_values")
    feats = t("feats")[:, :4].clone().requires_grad_(True)          # 4 of the 32 channels keep gradcheck small

    def f(x):
        out, _ = zo.volume_cost(imgs, x, proj, depth, 2)
        return out[-4:]                                              # the variance channels
    assert torch.autograd.gradcheck(f, (feats,), eps=1e-6, atol=1e-6, rtol=1e-4, nondet_tol=0.0)
