"""The two convolution stacks of the volume builder against a fixture generated from the REFERENCE's own classes
(tools/gen_golden.py run_builder_nets: networks.CostRegNet / networks.FeatureNet of /root/reference, constructed with
their `norm_act` argument = batch norm + leaky ReLU(0.01), the documented behaviour of the uninstallable
inplace_abn.InPlaceABN, and loaded with the seeded state dict under strict key matching).  Pinned by it: the wiring of
the stacks (layers, strides, paddings, transposed-convolution settings, skip additions) and the state-dict layout.
Not pinned: InPlaceABN's own arithmetic.  CPU: the modules' library path; GPU: the HIP kernels."""
import numpy as np
import pytest
import torch

import golden_cases as gc
import zest_networks as networks


def _nets(inp, device):
    cr, fn = networks.CostRegNet(41), networks.FeatureNet()
    for net, key in ((cr, "costreg_state"), (fn, "feature_state")):
        res = net.load_state_dict({k: torch.from_numpy(v) for k, v in inp[key].items()}, strict=False)
        assert not res.unexpected_keys and all(k.endswith("num_batches_tracked") for k in res.missing_keys)
    return cr.to(device), fn.to(device)


def _rel(got, want):
    want = torch.as_tensor(want)
    return float((got.detach().cpu() - want).abs().max() / want.abs().max())


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_library_path_matches_the_reference_stacks(mode):
    inp, gold = gc.build("builder_nets"), gc.load_golden("builder_nets")
    cr, fn = _nets(inp, "cpu")
    cr.train(mode == "train"), fn.train(mode == "train")
    with torch.no_grad():
        vol, levels = cr(torch.from_numpy(inp["cost"]))
        feats, stages = fn(torch.from_numpy(inp["imgs"]))
    assert len(levels) == 7 and len(stages) == 4                     # the reference returns its activation maps too
    assert _rel(vol, gold["costreg_" + mode]) < 2e-5
    assert _rel(feats, gold["feature_" + mode]) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("passes,tol", [(3, 2e-3), (1, 8e-2)])
def test_hip_kernels_match_the_reference_stacks(hip, mode, passes, tol):
    inp, gold = gc.build("builder_nets"), gc.load_golden("builder_nets")
    cr, fn = _nets(inp, "cuda:0")
    cr.train(mode == "train"), fn.train(mode == "train")
    cost = torch.zeros(16, 16, 24, 48, device="cuda:0")
    cost[..., :41] = torch.from_numpy(inp["cost"]).to("cuda:0")[0].permute(1, 2, 3, 0)
    with torch.no_grad():
        vol = cr.forward_hip(cost, passes=passes)
        feats = fn.forward_hip(torch.from_numpy(inp["imgs"]).to("cuda:0"), passes=passes)
    assert _rel(vol, gold["costreg_" + mode]) < tol
    assert _rel(feats.permute(0, 3, 1, 2), gold["feature_" + mode]) < tol
