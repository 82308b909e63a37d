"""GPU: the plane-sweep cost volume of the MVS volume builder.

Pinned part: the sampling half of the reference's homo_warp (given grid -> grid_sample) has a
reference-generated fixture (tests/golden/homo_warp.npz).  The grid construction and
build_volume_cost are checked against the oracle's restatement of the reference text ("parity
unpinned": the reference needs kornia.create_meshgrid / inplace_abn to run them, neither is
installed).  Tolerance: BASELINE's fp32 1e-4 abs + 1e-3 rel; sampling positions 2e-5.
"""
import numpy as np
import pytest
import torch

import golden_cases as gc
from test_hip_ops import G, close, ATOL, RTOL

pytestmark = pytest.mark.gpu


def _oracle_cost(inp):
    from oracle import zest_oracle as zo
    t = lambda k: torch.from_numpy(inp[k])[0]
    with torch.no_grad():
        return zo.volume_cost(t("imgs"), t("feats"), t("proj_mats"), t("depth_values"), inp["pad"])


def test_homo_warp_given_grid_matches_reference(hip):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import zest_utils as utils
    from oracle import zest_oracle as zo
    inp, gold = gc.build("homo_warp"), gc.load_golden("homo_warp")
    feats, imgs = G(inp["feats"]), G(inp["imgs"])
    H, W = feats.shape[-2:]
    g = zo.plane_grid(torch.from_numpy(inp["proj_mats"])[0, 1], torch.from_numpy(inp["depth_values"])[0], H, W, inp["pad"])
    D, Hp, Wp = g.shape[:3]
    grid = g.view(1, D, Wp, Hp, 2).cuda()
    with torch.no_grad():
        warped, _ = utils.homo_warp(feats[:, 1], G(inp["proj_mats"])[:, 1], G(inp["depth_values"]), src_grid=grid, pad=inp["pad"])
        img_lr = torch.nn.functional.interpolate(imgs[0], (H, W), mode="bilinear", align_corners=False)[1:2]
        img_warped, _ = utils.homo_warp(img_lr, None, None, src_grid=grid, pad=inp["pad"])
    close(warped[0], gold["warped"], name="warped")
    close(img_warped[0], gold["img_warped"], name="img_warped")


def test_homo_warp_builds_the_grid(hip):
    """Own grid construction (utils.py:57-89 restated) against the oracle, then the same warp."""
    import zest_utils as utils
    from oracle import zest_oracle as zo
    inp, gold = gc.build("homo_warp"), gc.load_golden("homo_warp")
    feats = G(inp["feats"])
    H, W = feats.shape[-2:]
    with torch.no_grad():
        warped, grid = utils.homo_warp(feats[:, 1], G(inp["proj_mats"])[:, 1], G(inp["depth_values"]), pad=inp["pad"])
    want = zo.plane_grid(torch.from_numpy(inp["proj_mats"])[0, 1], torch.from_numpy(inp["depth_values"])[0], H, W, inp["pad"])
    D, Hp, Wp = want.shape[:3]
    assert tuple(grid.shape) == (1, D, Wp, Hp, 2)
    close(grid.reshape(D, Hp, Wp, 2), want.numpy(), atol=2e-5, rtol=2e-5, name="grid")
    close(warped[0], gold["warped"], atol=3e-4, rtol=1e-3, name="warped (own grid)")


@pytest.mark.parametrize("V,pad,seed", [(3, 2, 61), (4, 0, 62), (2, 5, 64)])
def test_build_volume_cost(hip, V, pad, seed):
    import zest_networks as networks
    inp = gc.cost_inputs(seed, V=V, pad=pad)
    want_feat, want_masks = _oracle_cost(inp)
    net = networks.MVSNet.__new__(networks.MVSNet)          # the method needs no parameters
    torch.nn.Module.__init__(net)
    with torch.no_grad():
        img_feat, masks = net.build_volume_cost(G(inp["imgs"]), G(inp["feats"]), G(inp["proj_mats"]),
                                                G(inp["depth_values"]), pad=pad)
    assert tuple(img_feat.shape) == (1,) + tuple(want_feat.shape) and tuple(masks.shape) == (1,) + tuple(want_masks.shape)
    m, wm = masks[0].cpu().numpy(), want_masks.numpy()
    flips = (m != wm)
    assert flips.mean() < 1e-3, "mask flips %.4f" % flips.mean()      # |g| = 1 exactly: measure zero
    ok = ~flips.any(0)                                                  # voxels whose counts agree
    got, want = img_feat[0].cpu().numpy(), want_feat.numpy()
    # variance of 32 unit-normal features: |values| ~ 1; bilinear positions differ by ~1e-6 px
    assert np.all(np.abs(got - want)[:, ok] <= 3e-4 + 1e-3 * np.abs(want)[:, ok]), np.abs(got - want)[:, ok].max()
    assert 0.02 < wm[1:].mean() < 0.98                                  # the sweep leaves the frames


def test_volume_builder_end_to_end_shapes(hip):
    """MVSNet.forward with random weights: FeatureNet -> HIP plane sweep -> CostRegNet gives the
    8-channel encoding volume the renderer consumes (convolutions: parity unpinned, see module
    docstring)."""
    import zest_networks as networks
    torch.manual_seed(0)
    net = networks.MVSNet().cuda().eval()
    inp = gc.cost_inputs(65, V=3, H=16, W=24, pad=4)
    with torch.no_grad():
        vol, feats, depth = net(G(inp["imgs"]), G(inp["proj_mats"]), (2.0, 6.0), pad=4, return_color=True)
    assert tuple(vol.shape) == (1, 8, 128, 16 + 8, 24 + 8) and torch.isfinite(vol).all()
    assert tuple(feats.shape) == (1, 3, 4, 128, 24, 32) and tuple(depth.shape) == (1, 128)
    sd = net.state_dict()
    assert "cost_reg_2.conv7.1.running_var" in sd and "feature.conv0.0.bn.weight" in sd
