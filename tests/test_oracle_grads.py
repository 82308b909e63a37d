"""CPU: the oracle's autograd reproduces the gradient digests of the reference's own
training-mode backward (tests/golden/grad_*.npz from tools/gen_golden.py), so the oracle is a
pinned checker for the HIP backward kernels too."""
import numpy as np
import pytest

import golden_cases as gc
import oracle_run


@pytest.mark.parametrize("case", ["grad_zest_5f", "grad_static"])
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
