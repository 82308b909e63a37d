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


def _oracle_cost_grad(inp, gw):
    """Oracle forward + autograd: d sum(gw * img_feat) / d feats."""
    from oracle import zest_oracle as zo
    t = lambda k: torch.from_numpy(inp[k])[0]
    feats = t("feats").clone().requires_grad_(True)
    out, _ = zo.volume_cost(t("imgs"), feats, t("proj_mats"), t("depth_values"), inp["pad"])
    (out * torch.from_numpy(gw)).sum().backward()
    return feats.grad.numpy()


@pytest.mark.parametrize("V,pad,seed", [(3, 2, 61), (4, 0, 62), (2, 5, 64), (5, 1, 66)])
def test_build_volume_cost_backward(hip, V, pad, seed):
    """zest_volume_cost_bwd (scatter-add of d variance / d features through the bilinear taps, warped
    features gathered again) against the oracle's autograd (itself checked against finite differences in
    tests/test_oracle_grads.py).  Parity unpinned against the reference itself (build_volume_cost needs
    kornia / inplace_abn to run); the reference's graph has gradients exactly where this one has them."""
    import zest_networks as networks
    inp = gc.cost_inputs(seed, V=V, pad=pad)
    D, H, W = inp["depth_values"].shape[1], inp["feats"].shape[-2], inp["feats"].shape[-1]
    gw = gc.zs.rng(seed + 500).standard_normal((3 * V + 32, D, H + 2 * pad, W + 2 * pad)).astype(np.float32)
    want = _oracle_cost_grad(inp, gw)
    net = networks.MVSNet.__new__(networks.MVSNet)
    torch.nn.Module.__init__(net)
    feats = G(inp["feats"]).requires_grad_(True)
    img_feat, masks = net.build_volume_cost(G(inp["imgs"]), feats, G(inp["proj_mats"]), G(inp["depth_values"]), pad=pad)
    assert img_feat.requires_grad and not masks.requires_grad
    (img_feat[0] * G(gw)).sum().backward()
    got = feats.grad[0].cpu().numpy()
    scale = np.abs(want).max()
    # a sampling position that lands within ~1e-6 px of a pixel boundary picks other taps than the oracle
    # does: allow a handful of feature-map entries to differ, hold everything else to 1e-4 abs / 1e-3 rel
    bad = np.abs(got - want) > 1e-4 * scale + 1e-3 * np.abs(want)
    assert bad.mean() < 2e-3, "V=%d pad=%d: %.3f%% of the entries off, max err %.3g (scale %.3g)" % (
        V, pad, 100 * bad.mean(), np.abs(got - want).max(), scale)
    assert np.abs(got - want)[~bad].max() <= 1e-4 * scale + 1e-3 * scale


def test_homo_warp_backward(hip):
    import zest_utils as utils
    from oracle import zest_oracle as zo
    inp = gc.build("homo_warp")
    feats = torch.from_numpy(inp["feats"])[0, 1].clone().requires_grad_(True)
    H, W = feats.shape[-2:]
    g = zo.plane_grid(torch.from_numpy(inp["proj_mats"])[0, 1], torch.from_numpy(inp["depth_values"])[0], H, W, inp["pad"])
    gw = gc.zs.rng(5).standard_normal((feats.shape[0],) + tuple(g.shape[:3])).astype(np.float32)
    (zo.grid_warp(feats, g) * torch.from_numpy(gw)).sum().backward()
    D, Hp, Wp = g.shape[:3]
    src = G(inp["feats"])[:, 1].clone().requires_grad_(True)
    for grid in (g.view(1, D, Wp, Hp, 2).cuda(), None):            # given grid / own grid construction
        src.grad = None
        warped, _ = utils.homo_warp(src, G(inp["proj_mats"])[:, 1], G(inp["depth_values"]), src_grid=grid, pad=inp["pad"])
        (warped[0] * G(gw)).sum().backward()
        close(src.grad[0], feats.grad.numpy(), atol=1e-3, rtol=1e-3, name="homo_warp g_src (grid given: %s)" % (grid is not None))


def test_volume_builder_trains_end_to_end(hip):
    """MVSNet under autograd: FeatureNet -> HIP plane sweep (forward + backward kernels) -> CostRegNet.
    The parameter gradients equal those of the same modules with the oracle's (pure torch, CPU) plane
    sweep in the middle: library convolutions on both sides, only the sweep differs."""
    import copy
    import zest_networks as networks
    from oracle import zest_oracle as zo
    torch.manual_seed(3)
    net = networks.MVSNet().cuda().train()
    inp = gc.cost_inputs(67, V=3, H=8, W=16, pad=4)          # padded volume 128 x 16 x 24: the U-Net halves it three times
    imgs, proj = G(inp["imgs"]), G(inp["proj_mats"])
    ref = copy.deepcopy(net).cpu().double()

    def oracle_sweep(imgs_, feats_, proj_, depth_, pad=0):
        out, masks = zo.volume_cost(imgs_[0], feats_[0], proj_[0], depth_.reshape(depth_.shape[0], -1)[0], pad)
        return out[None], masks[None]
    ref.build_volume_cost = oracle_sweep
    vol, _, _ = net(imgs, proj, (2.0, 6.0), pad=4)
    gw = torch.randn(vol.shape, generator=torch.Generator().manual_seed(4))
    (vol * gw.cuda()).sum().backward()
    vol_r, _, _ = ref(imgs.cpu().double(), proj.cpu().double(), (2.0, 6.0), pad=4)
    (vol_r * gw.double()).sum().backward()
    close(vol[0], vol_r[0].detach().numpy(), atol=2e-3, rtol=2e-3, name="encoding volume")
    n = 0
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        g, w = p.grad.double().cpu().numpy(), q.grad.numpy()
        err = np.sqrt(((g - w) ** 2).sum()) / (np.sqrt((w ** 2).sum()) + 1e-12)
        assert err < 2e-2, "%s: relative L2 error %.3g" % (k, err)      # fp32 conv stacks (train-mode batch norm) vs fp64
        n += int(k.startswith("feature."))
    assert n > 10 and any(p.grad.abs().max() > 0 for k, p in net.named_parameters() if k.startswith("feature."))
