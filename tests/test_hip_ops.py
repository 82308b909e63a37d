"""GPU: each HIP operator against the reference-generated golden vectors and the oracle.

Tolerance for fp32 work is BASELINE.json's: 1e-4 abs + 1e-3 rel.  The bf16 MLP is a
throughput mode with bf16 operands (8 significant bits) and is held to 3e-2 abs + 3e-2 rel
of the output scale, stated per test.
"""
import numpy as np
import pytest
import torch

import golden_cases as gc
import oracle_run

pytestmark = pytest.mark.gpu
ATOL, RTOL = 1e-4, 1e-3
DEV = "cuda:0"


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(got, want, atol=ATOL, rtol=RTOL, name=""):
    got = got.detach().double().cpu().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    both_nan = np.isnan(got) & np.isnan(want)
    err = np.where(both_nan, 0, np.abs(got - want))
    tol = atol + rtol * np.abs(np.where(both_nan, 0, want))
    bad = err > tol
    assert not bad.any(), "%s: %d/%d outside tolerance, max err %.3g at |ref| %.3g" % (
        name, bad.sum(), bad.size, err.max(), np.abs(want).flat[np.nanargmax(err)])


@pytest.mark.parametrize("case", ["composite", "composite_white"])
def test_composite(hip, case):
    import zest_hip
    inp, gold = gc.build(case), gc.load_golden(case)
    r = zest_hip.composite(G(inp["raw"]), G(inp["z"]), G(inp["rays_dir"]),
                           white_bkgd=gc.CASES[case].get("white_bkgd", False))
    for n, v in zip(("rgb_map", "disp_map", "acc_map", "weights", "depth_map", "alpha"), r):
        close(v, gold[n], name=n)


def test_composite_noise_and_properties(hip):
    import zest_hip
    inp = gc.composite_inputs(101, R=37, S=150, dead_ray=True)     # ragged: 150 = 2*64 + 22
    inp["raw"][2, :, 3] = 1e5                                       # opaque at the first sample
    noise = gc.zs.rng(5).standard_normal((37, 150)).astype(np.float32)
    from oracle import zest_oracle as zo
    z, d, raw = (torch.from_numpy(inp[k]) for k in ("z", "rays_dir", "raw"))
    dists = zo.sample_dists(z, torch.linalg.vector_norm(d, dim=-1, keepdim=True))
    want = zo.composite(raw, z, dists, True, torch.from_numpy(noise) * 0.7)
    got = zest_hip.composite(G(inp["raw"]), G(inp["z"]), G(inp["rays_dir"]), noise=G(noise),
                             noise_std=0.7, white_bkgd=True)
    for n, g, w in zip(("rgb_map", "disp_map", "acc_map", "weights", "depth_map", "alpha"), got, want):
        close(g, w.numpy(), name=n)
    assert (got[3] >= 0).all() and (got[3].sum(-1) <= 1 + 1e-5).all()
    w = zest_hip.composite(G(inp["raw"]), G(inp["z"]), G(inp["rays_dir"]))[3].cpu().numpy()   # no noise
    assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-5).all()
    assert np.abs(w[1]).max() == 0.0                       # sigma <= 0 everywhere
    assert abs(w[2, 0] - 1.0) < 1e-6 and np.abs(w[2, 1:]).max() < 1e-9   # opaque first sample


def test_blend(hip):
    import zest_hip
    inp, gold = gc.build("blend"), gc.load_golden("blend")
    r = zest_hip.composite_blend(G(inp["raw_dy"]), G(inp["raw_st"]), G(inp["blend"]), G(inp["z"]),
                                 G(inp["rays_dir"]))
    for n, v in zip(("rgb_map", "depth_map", "rgb_map_fg", "depth_map_fg", "weights_fg", "weights_dy"), r):
        close(v, gold[n], name=n)
    close(r[6], gold["weights_dy"].sum(-1), name="weights_dd_sum")


@pytest.mark.parametrize("case", ["embed3x10", "embed4x10", "embed3x4"])
def test_embed(hip, case):
    import zest_hip
    inp, gold = gc.build(case), gc.load_golden(case)
    y = zest_hip.embed(G(inp["x"]), gc.CASES[case]["L"])
    close(y, gold["y"], atol=2e-6, rtol=0, name=case)      # encoder itself is held to 2e-6


def test_embed_large_arguments(hip):
    import zest_hip
    x = (gc.zs.rng(3).uniform(-60, 60, size=(4096, 3))).astype(np.float32)
    y = zest_hip.embed(G(x), 10).cpu().numpy()
    x64 = x.astype(np.float64)
    for k in range(10):
        assert np.abs(y[:, 3 + 6 * k:6 + 6 * k] - np.sin(x64 * 2 ** k)).max() < 3e-7
        assert np.abs(y[:, 6 + 6 * k:9 + 6 * k] - np.cos(x64 * 2 ** k)).max() < 3e-7


def test_volume_lookup(hip):
    import zest_hip
    inp, gold = gc.build("volume"), gc.load_golden("volume")
    vcl = zest_hip.volume_to_cl(G(inp["volume"]))
    # the kernels' copy is channels-last and depth-innermost: [H,W,D,8]; and the way back ([8,D,H,W]) is exact
    assert torch.equal(vcl, G(inp["volume"])[0].permute(2, 3, 1, 0).contiguous())
    assert torch.equal(zest_hip.volume_from_cl(vcl), G(inp["volume"]))
    odd = torch.randn(1, 8, 19, 7, 45, device="cuda:0")            # ragged against the 16 x 32 transpose tile
    assert torch.equal(zest_hip.volume_to_cl(odd), odd[0].permute(2, 3, 1, 0).contiguous())
    assert torch.equal(zest_hip.volume_from_cl(zest_hip.volume_to_cl(odd)), odd)
    close(zest_hip.volume_lookup(vcl, G(inp["ndc"])[0]), gold["feat"], name="volume")


def test_color_lookup(hip):
    import zest_hip
    inp, gold = gc.build("color"), gc.load_golden("color")
    icl = zest_hip.images_to_cl(G(inp["imgs"]))
    got = zest_hip.color_lookup(icl, G(inp["w2cs"])[0], G(inp["intrinsics"])[0], G(inp["pts"])[0])
    close(got, gold["colors"], name="colors")
    m = got.cpu().numpy()[..., 3::4]
    assert set(np.unique(m)) <= {0.0, 1.0}


def _mlp_setup(case):
    import zest_hip
    inp = gc.build(case)
    head = zest_hip.HEAD_NONE
    if inp["sceneflow"] and inp["net_type"] == "v0":
        head = zest_hip.HEAD_BLEND if inp["static"] else zest_hip.HEAD_DYNAMIC
    use_feat = inp["use_mvs"] or inp["net_type"] == "v2"
    shape = ()
    if (inp["D"], inp["W"], tuple(inp["skips"])) != (8, 256, (4,)):
        shape = (inp["D"], inp["W"], sum(1 << i for i in inp["skips"]))
    desc = zest_hip.MlpDesc(inp["P"], inp["Fd"], gc.PE_DIR, int(use_feat),
                            2 if inp["net_type"] == "v2" else 0, head, *shape)
    state = {k: G(v) for k, v in inp["state"].items()}
    return zest_hip, inp, desc, zest_hip.param_table(state, desc)


def _mlp_module(inp):
    """zest_networks.MVSNeRF of an MLP case (any depth / width / skips) with the case's weights, on the GPU."""
    import zest_networks as networks
    net = networks.MVSNeRF(D=inp["D"], W=inp["W"], skips=list(inp["skips"]), input_ch_pts=inp["P"],
                           input_ch_views=gc.PE_DIR, input_ch_feat=inp["Fd"], net_type=inp["net_type"],
                           sceneflow=inp["sceneflow"], static=inp["static"], use_mvs=inp["use_mvs"])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in inp["state"].items()})
    return net.cuda()


# every MLP case runs in the fp32 kernel and the fp32 training path; the MFMA engine (bf16 / fp16 / split
# fp16) is written for the shipped shape D=8, W=256, skips=[4]
ALL_MLP_CASES = [c for c in gc.CASES if gc.CASES[c]["kind"] == "mlp"]
SHAPE_MLP_CASES = [c for c in ALL_MLP_CASES if gc.mlp_shape(gc.CASES[c]["variant"]) != (8, 256, (4,))]
MLP_CASES = [c for c in ALL_MLP_CASES if c not in SHAPE_MLP_CASES]


@pytest.mark.parametrize("case", ALL_MLP_CASES)
def test_mlp_f32(hip, case):
    zh, inp, desc, tab = _mlp_setup(case)
    packed = zh.mlp_pack(desc, zh.PREC_F32, tab)
    y = zh.mlp_fwd(desc, zh.PREC_F32, packed, G(inp["x"])[0])
    close(y, gc.load_golden(case)["y"], name=case)


@pytest.mark.parametrize("case", MLP_CASES)
def test_mlp_bf16(hip, case):
    zh, inp, desc, tab = _mlp_setup(case)
    packed = zh.mlp_pack(desc, zh.PREC_BF16, tab)
    y = zh.mlp_fwd(desc, zh.PREC_BF16, packed, G(inp["x"])[0])
    gold = gc.load_golden(case)["y"]
    scale = np.abs(gold).max()
    close(y, gold, atol=3e-2 * scale, rtol=3e-2, name=case)


@pytest.mark.parametrize("case", SHAPE_MLP_CASES)
def test_mlp_other_shapes_refuse_the_engine_and_run_fp32_from_the_module(hip, case, monkeypatch):
    """Depths / widths / skips other than 8 / 256 / [4] (reference networks.py:93-100): the engine precisions
    refuse them at the C ABI with a message, and the module routes every mode to the fp32 kernel."""
    zh, inp, desc, tab = _mlp_setup(case)
    for prec in (zh.PREC_BF16, zh.PREC_F16, zh.PREC_F16X3):
        with pytest.raises(RuntimeError, match="depth 8 / width 256"):
            zh.mlp_pack(desc, prec, tab)
    gold = gc.load_golden(case)
    for mode in ("f32", "bf16", "f16"):
        monkeypatch.setenv("ZEST_PRECISION", mode)
        net = _mlp_module(inp)
        with torch.no_grad():
            y = net(G(inp["x"]))
            close(y[0], gold["y"], name="%s/%s" % (case, mode))
            if "alpha_only" in gold:
                a = net.forward_alpha(G(inp["x"])[..., :inp["P"] + inp["Fd"]])
                close(a[0], gold["alpha_only"], name="%s/%s/forward_alpha" % (case, mode))


def test_mlp_ragged_batch(hip):
    """M not a multiple of the 32-sample tile, and M smaller than one tile."""
    zh, inp, desc, tab = _mlp_setup("mlp_static_mvs20")
    gold = gc.load_golden("mlp_static_mvs20")["y"]
    for prec, tol in ((zh.PREC_F32, (ATOL, RTOL)), (zh.PREC_BF16, (3e-2 * np.abs(gold).max(), 3e-2))):
        packed = zh.mlp_pack(desc, prec, tab)
        for M in (1, 31, 45):
            y = zh.mlp_fwd(desc, prec, packed, G(inp["x"])[0, :M])
            close(y, gold[:M], atol=tol[0], rtol=tol[1], name="M=%d" % M)


@pytest.mark.parametrize("V", [1, 2, 5, 6, 7, 11, 14])
def test_mlp_view_counts(hip, V):
    """Feature operand layouts other than the fixtures' (V = 3, 4, 8): one k-tile up to 6 source
    views (all four lane groups used from V = 5 on), two k-tiles up to 14.  Checked against the
    oracle (pinned by the fixtures) on seeded inputs; both MFMA kernels."""
    import torch
    import zest_hip as zh
    from oracle import zest_oracle as zo
    import oracle_run as orun
    import zest_synth as zs
    Fd, M = 8 + 4 * V, 80
    lay = zs.mlp_layout(gc.PE_PTS, gc.PE_DIR, Fd, False, True, True)
    state = zs.fill_mlp_state(lay, 700 + V)
    x = zs.rng(800 + V).uniform(-1, 1, size=(M, gc.PE_PTS + Fd + gc.PE_DIR)).astype(np.float32)
    with torch.no_grad():
        want = zo.mlp_forward(orun.state_t(state, torch.float32), torch.from_numpy(x),
                              orun.spec_of(gc.PE_PTS, Fd, False, True, True)).numpy()
    desc = zh.MlpDesc(gc.PE_PTS, Fd, gc.PE_DIR, 1, 0, zh.HEAD_NONE)
    tab = zh.param_table({k: G(v) for k, v in state.items()}, desc)
    y32 = zh.mlp_fwd(desc, zh.PREC_F32, zh.mlp_pack(desc, zh.PREC_F32, tab), G(x))
    close(y32, want, name="fp32 V=%d" % V)
    y16 = zh.mlp_fwd(desc, zh.PREC_BF16, zh.mlp_pack(desc, zh.PREC_BF16, tab), G(x))
    close(y16, want, atol=3e-2 * np.abs(want).max(), rtol=3e-2, name="bf16 V=%d" % V)


def test_channels_last_cache_hits_on_fresh_views(hip, monkeypatch):
    """The generators hand `imgs[:, :-1]` (a NEW view object per chunk) to the renderer: the channels-last
    copy must be made once per image, not once per chunk; an in-place update or a different slice misses."""
    import zest_hip
    import zest_utils
    calls = []
    real = zest_hip.images_to_cl
    monkeypatch.setattr(zest_hip, "images_to_cl", lambda t: (calls.append(1), real(t))[1])
    zest_utils._CL_CACHE.clear()
    imgs = torch.rand(1, 4, 3, 12, 16, device="cuda")
    a = zest_utils.images_channels_last(imgs[:, :-1])
    b = zest_utils.images_channels_last(imgs[:, :-1])
    assert a is b and len(calls) == 1
    zest_utils.images_channels_last(imgs[:, 1:])              # another slice of the same memory
    assert len(calls) == 2
    imgs.mul_(0.5)                                            # new content: version counter moved
    c = zest_utils.images_channels_last(imgs[:, :-1])
    assert len(calls) == 3 and torch.equal(c[..., :3], imgs[0, :-1].permute(0, 2, 3, 1))
    del imgs, a, b, c
    fresh = torch.rand(1, 4, 3, 12, 16, device="cuda")        # may reuse the freed address: must not hit
    d = zest_utils.images_channels_last(fresh[:, :-1])
    assert torch.equal(d[..., :3], fresh[0, :-1].permute(0, 2, 3, 1))


@pytest.mark.parametrize("case", ["mlp_static_mvs20", "mlp_static_sf_mvs40", "mlp_dynamic_mvs24", "mlp_v2_mvs20"])
@pytest.mark.parametrize("exact", [False, True], ids=["f16x3", "exact"])
def test_forward_alpha_matches_reference(hip, case, exact, monkeypatch):
    """Renderer.forward_alpha / Renderer_linear.forward_alpha (reference networks.py:134-147, 266-280)
    against the reference's own outputs (fixture key `alpha_only`), both fp32-mode kernels."""
    import zest_networks as networks
    monkeypatch.setenv("ZEST_FP32_EXACT", "1" if exact else "0")
    inp, gold = gc.build(case), gc.load_golden(case)
    net = _mlp_module(inp)
    with torch.no_grad():
        a = net.forward_alpha(G(inp["x"])[..., :inp["P"] + inp["Fd"]])
        y = net(G(inp["x"]))                        # the full forward still packs and runs its own variant
    assert a.shape == (1, 64, 1)
    close(a[0], gold["alpha_only"], name=case + "/forward_alpha")
    close(y[0], gold["y"], name=case + "/forward after forward_alpha")


def test_sample_pdf_inverse_cdf_properties(hip):
    """sample_pdf (named by BASELINE.json's north_star; the reference has no counterpart: parity unpinned):
    against a float64 numpy inverse CDF, and by its properties - samples ascending for ascending quantiles,
    inside the bins, concentrated where the weight is, uniform weights give back evenly spaced depths."""
    import zest_hip
    import zest_utils
    g = gc.zs.rng(321)
    for R, Nb, Ns in ((37, 127, 64), (5, 63, 200), (3, 1, 9), (2, 300, 128)):
        edges = np.sort(g.uniform(2.0, 6.0, size=(R, Nb + 1)).astype(np.float32), -1)
        w = g.uniform(0, 1, size=(R, Nb)).astype(np.float32) ** 4
        w[0, : Nb // 2] = 0.0                                                  # a dead half
        u = np.sort(g.uniform(0, 1, size=(R, Ns)).astype(np.float32), -1)
        got = zest_hip.sample_pdf(G(edges), G(w), u=G(u)).cpu().numpy()
        pdf = (w.astype(np.float64) + 1e-5) / (w.astype(np.float64) + 1e-5).sum(-1, keepdims=True)
        cdf = np.concatenate([np.zeros((R, 1)), np.cumsum(pdf, -1)], -1)
        want = np.empty_like(got, dtype=np.float64)
        for r in range(R):
            idx = np.searchsorted(cdf[r], u[r].astype(np.float64), side="right")
            below, above = np.maximum(idx - 1, 0), np.minimum(idx, Nb)
            den = cdf[r, above] - cdf[r, below]
            den = np.where(den < 1e-5, 1.0, den)
            want[r] = edges[r, below] + (u[r] - cdf[r, below]) / den * (edges[r, above] - edges[r, below])
        # a quantile that falls within fp32 rounding of a cdf knot may pick the neighbouring bin: continuous there
        assert np.abs(got - want).max() < 2e-3 * (edges.max() - edges.min()), np.abs(got - want).max()
        assert (np.diff(got, axis=-1) >= -1e-5).all()
        assert (got >= edges[:, :1] - 1e-6).all() and (got <= edges[:, -1:] + 1e-6).all()
        if Nb > 1:
            assert (got[0] >= edges[0, Nb // 2] - 1e-3).mean() > 0.97                # mass only in the live half
    z = np.linspace(2, 6, 65, dtype=np.float32)[None]
    det = zest_utils.sample_pdf(G(z), torch.ones(1, 64, device="cuda"), 33, det=True)
    close(det[0], np.linspace(2, 6, 33), atol=1e-4, rtol=0, name="uniform weights")


@pytest.mark.parametrize("D,H,W", [(128, 120, 176), (128, 208, 288)])       # NSFF and LLFF encoding volumes
def test_volume_layout_change_at_full_size(hip, D, H, W):
    """The tile transpose [8,D,H,W] <-> [H,W,D,8] at the reference's real volume sizes (86.5 / 245 MB): exact both
    ways, and the lookup on the copy equals the lookup on a permuted view made by torch."""
    import zest_hip
    g = torch.Generator(device="cuda").manual_seed(D + H + W)
    vol = torch.randn(1, 8, D, H, W, device="cuda:0", generator=g)
    cl = zest_hip.volume_to_cl(vol)
    assert cl.shape == (H, W, D, 8) and torch.equal(cl, vol[0].permute(2, 3, 1, 0))
    assert torch.equal(zest_hip.volume_from_cl(cl), vol)
    ndc = torch.rand(4096, 3, device="cuda:0", generator=g) * 1.1 - 0.05          # some samples outside the volume
    got = zest_hip.volume_lookup(cl, ndc)
    want = torch.nn.functional.grid_sample(vol, (ndc * 2 - 1).view(1, 1, 1, -1, 3), mode="bilinear",
                                           padding_mode="zeros", align_corners=True)[0, :, 0, 0].t()
    assert (got - want).abs().max().item() < 2e-5
