"""GPU: `renderer.rendering` (the drop-in boundary) against the reference's golden outputs,
plus the module-level API (`MVSNeRF`, `Embedding`, `utils.*`, `raw2outputs*`).

fp32 tolerance is BASELINE.json's 1e-4 abs + 1e-3 rel.  Outputs that pass through the
scene-flow chain twice (t+-2: raw_pts_pp, rgb_map_pp_dy) are held to 3e-4 abs: the reference
itself sits 5e-5 from an fp64 evaluation there because a 1e-7 perturbation of a displaced
point is amplified by sin(512 x) (see tests/test_oracle_golden.py).
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import golden_cases as gc
import oracle_run
from test_hip_ops import G, close, ATOL, RTOL

pytestmark = pytest.mark.gpu
RENDER_CASES = [c for c in gc.CASES if gc.CASES[c]["kind"] == "render"]
CHAIN2 = ("raw_pts_pp", "rgb_map_pp_dy")


def build_nets(sc, net_type="v0"):
    import zest_networks as networks
    sf = sc["scene_flow"]
    D, W, skips = sc.get("static_shape", (8, 256, (4,)))
    ns = networks.MVSNeRF(D=D, W=W, input_ch_pts=gc.PE_PTS + sc.get("time_dim", 0), output_ch=4,
                          input_ch_views=gc.PE_DIR, input_ch_feat=sc["feat_dim"], skips=list(skips), net_type=net_type,
                          sceneflow=sf, static=True, use_mvs=sc["use_mvs"])
    ns.load_state_dict({k: torch.from_numpy(v) for k, v in sc["state_static"].items()})
    nd = None
    if sf:
        nd = networks.MVSNeRF(D=8, W=256, input_ch_pts=gc.PE_XYZT, output_ch=4,
                              input_ch_views=gc.PE_DIR, input_ch_feat=24, skips=[4], net_type="v0",
                              sceneflow=True, static=False, use_mvs=sc["use_mvs_dy"])
        nd.load_state_dict({k: torch.from_numpy(v) for k, v in sc["state_dynamic"].items()})
        nd = nd.to("cuda:0")
    return ns.to("cuda:0"), nd


def render_scene(sc, c, precision=32, monkeypatch=None, maps_only=False, dtype16="bf16", net_type="v0",
                 fp32_exact=False, nets=None):
    """renderer.rendering on a scene dict of golden_cases.render_inputs with the flags of case dict c.
    nets: (static, dynamic) modules to reuse (their packed-weight caches with them)."""
    import zest_networks as networks
    import zest_renderer as renderer
    sf = sc["scene_flow"]
    ns, nd = nets if nets is not None else build_nets(sc, net_type)
    args = SimpleNamespace(netchunk=1024, feat_dim=sc["feat_dim"], feat_dim_dy=24, img_downscale=1.0,
                           use_color_volume=False, net_type=net_type, precision=precision,
                           zest_maps_only=maps_only, zest_dtype16=dtype16, zest_fp32_exact=fp32_exact)
    cam = {"w2cs": G(sc["w2cs"]), "intrinsics": G(sc["intrinsics"])}
    dy = sf and sc["use_mvs_dy"]
    nb_cam = {"w2cs": G(sc["nb_w2cs"]), "intrinsics": G(sc["nb_intrinsics"])} if dy else None
    if sf and monkeypatch is not None:
        queue = [G(sc["noise_static"])[None], G(sc["noise_blend"])[None]]
        monkeypatch.setattr(renderer, "_draw_noise", lambda shape, device: queue.pop(0))
    with torch.no_grad():
        return renderer.rendering(
            args, G(sc["rays_pts"]), G(sc["rays_ndc"]), G(sc["depth_candidates"]), G(sc["rays_dir"]),
            volume_feature_static=G(sc["vol_static"]) if sc["use_mvs"] else None,
            volume_feature_dynamic=G(sc["vol_dynamic"]) if dy else None,
            imgs=G(sc["imgs"]) if sc["use_mvs"] else None,
            neighbour_frames=G(sc["nb_imgs"]) if dy else None,
            im_cam_mat=cam, nb_cam_mat=nb_cam, network_fn=ns, network_fn_dy=nd,
            embedding_pts=networks.Embedding(3, 10), embedding_xyzt=networks.Embedding(4, 10),
            embedding_dir=networks.Embedding(3, 4),
            chain_bwd=c.get("chain_bwd", False), chain_5frames=c.get("chain_5frames", False),
            ref_frame_idx=gc.REF_FRAME_IDX, num_frames=gc.NUM_FRAMES,
            white_bkgd=c.get("white_bkgd", False), scene_flow=sf, val=c.get("val", False),
            raw_noise_std=c.get("raw_noise_std", 0),
            time_codes=G(sc["time_codes"]) if sc.get("time_dim", 0) else None)


def call_rendering(case, precision=32, monkeypatch=None, maps_only=False, dtype16="bf16", fp32_exact=False):
    return render_scene(gc.build(case), gc.CASES[case], precision, monkeypatch, maps_only, dtype16,
                        fp32_exact=fp32_exact)


@pytest.mark.parametrize("exact", [False, True], ids=["f16x3", "exact"])
@pytest.mark.parametrize("case", RENDER_CASES)
def test_rendering_matches_reference_fp32(hip, case, exact, monkeypatch):
    """Every key of the reference's result dict, all ten scenes, both fp32-mode MLP kernels: split-fp16
    pairs (the default) and exact fp32 products (args.zest_fp32_exact)."""
    ret = call_rendering(case, 32, monkeypatch, fp32_exact=exact)
    gold = gc.load_golden(case)
    assert sorted(ret.keys()) == gold["__keys__"].tolist()
    none_keys = set(x for x in gold["__none_keys__"].tolist() if x)
    assert set(k for k, v in ret.items() if v is None) == none_keys
    for k, v in ret.items():
        if v is None:
            continue
        assert v.shape[0] == 1 and v.is_cuda
        close(v[0], gold[k], atol=3e-4 if k in CHAIN2 else ATOL, rtol=RTOL, name="%s/%s" % (case, k))


@pytest.mark.parametrize("case", ["render_static_mvs", "render_static_nomvs", "render_zest_val"])
def test_rendering_bf16_mode(hip, case, monkeypatch):
    """args.precision=16 selects the bf16 MFMA engine: per-ray maps within 2e-2 of the reference
    (bf16 operands carry 8 significant bits through 10 layers)."""
    ret = call_rendering(case, 16, monkeypatch)
    gold = gc.load_golden(case)
    for k in ("rgb_map", "rgb_map_ref", "rgb_map_ref_dy"):
        if k in gold:
            close(ret[k][0], gold[k], atol=2e-2, rtol=0, name=k)
    for k in ("depth_map", "depth_map_ref"):
        if k in gold:
            close(ret[k][0], gold[k], atol=6e-2, rtol=0, name=k)


@pytest.mark.parametrize("maps_only", [False, True], ids=["per-op", "fused"])
def test_time_code_of_the_next_frame_is_not_served_from_the_cache(hip, maps_only):
    """The reference's callers hand rendering() a NEW `time_codes[frame].to(device)` tensor per batch
    (train.py:849, 1058); the allocator gives the next frame's code the block the last one just freed, at
    version 0.  The packed weights carry the code in the biases of layers 0 and 5, so the cache must tell two
    codes apart by value: frame A, then frame B in recycled memory, then A again - each against the oracle
    evaluated with that frame's code."""
    case = "render_static_timecodes"
    sc, c = gc.build(case), gc.CASES[case]
    nets = build_nets(sc, "v0")
    code_a = np.array(sc["time_codes"], dtype=np.float32)
    code_b = (code_a[..., ::-1] * 1.7 + 0.9).astype(np.float32).copy()
    for code in (code_a, code_b, code_a, code_b):
        sc_f = dict(sc, time_codes=code)
        want = oracle_run.oracle_render(c, sc_f, explicit=False)
        ret = render_scene(sc_f, c, 32, maps_only=maps_only, nets=nets)   # the code tensor dies with the call
        for k in ("rgb_map", "depth_map"):
            close(ret[k][0], want[k].numpy(), atol=ATOL, rtol=RTOL, name="%s/%s" % (case, k))
    a = render_scene(dict(sc, time_codes=code_a), c, 32, maps_only=maps_only, nets=nets)["rgb_map"]
    b = render_scene(dict(sc, time_codes=code_b), c, 32, maps_only=maps_only, nets=nets)["rgb_map"]
    assert (a - b).abs().max() > 1e-3, "the two codes must render differently for this test to mean anything"


FUSED_CASES = ["render_static_mvs", "render_static_nomvs", "render_static_white", "render_zest_val",
               "render_zest_nomvsdy"]      # (the time-code scene: tests/test_hip_precision.py, all three operand types)


@pytest.mark.parametrize("case", FUSED_CASES)
def test_fused_renderer(hip, case):
    """The fused inference path (bf16 engine, nothing per-sample in HBM): agrees with the per-op
    bf16 path to a few bf16 ulps of the operands (the fused encoder uses the hardware sine, whose
    1e-6 error flips an occasional bf16 rounding), and is within the bf16 tolerance of the
    reference's golden maps."""
    fused = call_rendering(case, 16, maps_only=True)
    perop = call_rendering(case, 16)
    gold = gc.load_golden(case)
    keys = [k for k in fused if k not in ("acc_map", "zest_packed_maps")]
    assert set(keys) == {k for k in ("rgb_map", "depth_map", "rgb_map_ref", "depth_map_ref",
                                     "rgb_map_ref_dy", "depth_map_ref_dy", "weights_map_dd") if k in gold}
    for k in keys:
        close(fused[k][0], perop[k][0].cpu().numpy(), atol=3e-2 if "depth" in k else 4e-3, rtol=0,
              name="fused~perop/" + k)
        close(fused[k][0], gold[k], atol=6e-2 if "depth" in k else 2e-2, rtol=0, name="fused~ref/" + k)


def test_module_api(hip):
    import zest_networks as networks
    import zest_renderer as renderer
    import zest_utils as utils
    # MVSNeRF.forward on [1, M, C] like the reference's batchify call
    inp, gold = gc.build("mlp_static_sf_mvs40"), gc.load_golden("mlp_static_sf_mvs40")
    net = networks.MVSNeRF(D=8, W=256, input_ch_pts=63, input_ch_views=27, input_ch_feat=40,
                           net_type="v0", sceneflow=True, static=True, use_mvs=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in inp["state"].items()})
    net = net.cuda()
    with torch.no_grad():
        y = net(G(inp["x"]))
    assert y.shape == (1, 64, 5)
    close(y[0], gold["y"], name="MVSNeRF.forward")
    # parameter update invalidates the packed-weight cache
    with torch.no_grad():
        net.nerf.alpha_linear.bias.add_(1.0)
        y2 = net(G(inp["x"]))
    close(y2[0, :, 3], gold["y"][:, 3] + 1.0, name="repack after update")
    yt = net(G(inp["x"]))                                       # grad mode: the HIP training path
    assert yt.requires_grad and yt.grad_fn is not None
    close(yt[0], y2[0].cpu().numpy(), name="training forward == inference forward")
    # Embedding / utils
    e = networks.Embedding(3, 10)
    assert e.out_channels == 63
    xi = gc.build("embed3x10")
    close(e(G(xi["x"])), gc.load_golden("embed3x10")["y"], atol=2e-6, rtol=0, name="Embedding")
    vi = gc.build("volume")
    close(utils.index_point_feature(G(vi["volume"]), G(vi["ndc"]))[0], gc.load_golden("volume")["feat"],
          name="index_point_feature")
    ci = gc.build("color")
    poses = {"w2cs": G(ci["w2cs"]), "intrinsics": G(ci["intrinsics"])}
    col = utils.build_color_volume(G(ci["pts"]), poses, G(ci["imgs"]), with_mask=True)
    close(col[0], gc.load_golden("color")["colors"], name="build_color_volume")
    col3 = utils.build_color_volume(G(ci["pts"]), poses, G(ci["imgs"]))
    assert col3.shape[-1] == 9
    # raw2outputs with caller-provided dists
    comp, cg = gc.build("composite"), gc.load_golden("composite")
    z = G(comp["z"])[None]
    dists = renderer.depth2dist(z, torch.norm(G(comp["rays_dir"])[None], dim=-1, keepdim=True))
    close(dists[0], cg["dists"], name="depth2dist")
    with torch.no_grad():
        r = renderer.raw2outputs(G(comp["raw"])[None], z, dists)
    for n, v in zip(("rgb_map", "disp_map", "acc_map", "weights", "depth_map", "alpha"), r):
        close(v[0], cg[n], name="raw2outputs/" + n)
    sig = torch.relu(G(comp["raw"])[None, ..., 3])
    a, w = renderer.raw2alpha(sig, dists)
    close(a[0], cg["alpha"], name="raw2alpha/alpha")
    close(w[0], cg["weights"], name="raw2alpha/weights")


RAYS_CASES = [c for c in gc.CASES if gc.CASES[c]["kind"] == "rays"]


@pytest.mark.parametrize("case", RAYS_CASES)
def test_build_rays_matches_reference(hip, case):
    """utils.build_rays_dy (ray sampling, the step in front of the renderer) against the
    reference's outputs for the same seeded pixels and injected stratified jitter."""
    import os
    import sys
    sys.path.insert(0, os.path.join(gc.ROOT, "tools"))
    import gen_golden
    import zest_utils as utils
    got = gen_golden.run_rays(utils, gc.CASES[case], gc.build(case), wrap=G)
    gold = gc.load_golden(case)
    assert sorted(got) == sorted(gold)
    for k, v in gold.items():
        close(got[k], v, atol=2e-6, rtol=2e-6, name="%s/%s" % (case, k))


def test_get_ndc_coordinate(hip):
    import zest_utils as utils
    from oracle import zest_oracle as zo
    inp = gc.build("rays_random")
    pts = gc.zs.rng(9).uniform(-2, 2, size=(1, 5, 7, 3)).astype(np.float32) + np.array([0, 0, 4], np.float32)
    w2c, K = G(inp["w2cs"])[:, 1], G(inp["intrinsics"])[:, 1]
    got = utils.get_ndc_coordinate(w2c, K, G(pts), torch.tensor([31, 23]), near=2.0, far=6.0, pad=3)
    want = zo.ndc_coordinate(torch.from_numpy(pts)[0], torch.from_numpy(inp["w2cs"])[0, 1],
                             torch.from_numpy(inp["intrinsics"])[0, 1], 31, 23, 2.0, 6.0, 3)
    close(got[0], want.numpy(), atol=2e-6, rtol=2e-6, name="get_ndc_coordinate")


def test_ray_permutation_commutes(hip):
    """Rays are independent: rendering a permuted batch permutes the outputs (basis of the
    multi-GPU ray sharding)."""
    import zest_hip
    inp = gc.composite_inputs(77, R=64, S=48, dead_ray=False)
    perm = np.random.default_rng(0).permutation(64)
    a = zest_hip.composite(G(inp["raw"]), G(inp["z"]), G(inp["rays_dir"]))
    b = zest_hip.composite(G(inp["raw"][perm]), G(inp["z"][perm]), G(inp["rays_dir"][perm]))
    for x, y in zip(a, b):
        assert torch.equal(x[torch.from_numpy(perm).cuda()], y)


@pytest.mark.parametrize("V", [1, 6, 7, 14])
def test_fused_view_counts(hip, V):
    """Fused renderer with feature operands of one k-tile (V <= 6) and two (V <= 14), against the
    per-op bf16 path and the oracle on a seeded scene with V source views."""
    import zest_networks as networks
    import zest_renderer as renderer
    import oracle_run as orun
    sc = gc.render_inputs(900 + V, R=24, S=40, V=V, use_mvs=True)
    ns, _ = build_nets(sc)
    cam = {"w2cs": G(sc["w2cs"]), "intrinsics": G(sc["intrinsics"])}

    def call(maps_only):
        args = SimpleNamespace(netchunk=1024, feat_dim=sc["feat_dim"], feat_dim_dy=24, img_downscale=1.0,
                               use_color_volume=False, net_type="v0", precision=16, zest_maps_only=maps_only)
        with torch.no_grad():
            return renderer.rendering(
                args, G(sc["rays_pts"]), G(sc["rays_ndc"]), G(sc["depth_candidates"]), G(sc["rays_dir"]),
                volume_feature_static=G(sc["vol_static"]), imgs=G(sc["imgs"]), im_cam_mat=cam,
                network_fn=ns, embedding_pts=networks.Embedding(3, 10),
                embedding_xyzt=networks.Embedding(4, 10), embedding_dir=networks.Embedding(3, 4),
                ref_frame_idx=gc.REF_FRAME_IDX, num_frames=gc.NUM_FRAMES, scene_flow=False, val=True)
    fused, perop = call(True), call(False)
    want = orun.oracle_render(dict(val=True), sc)
    for k in ("rgb_map", "depth_map"):
        close(fused[k][0], perop[k][0].cpu().numpy(), atol=3e-2 if "depth" in k else 4e-3, rtol=0,
              name="fused~perop/%s V=%d" % (k, V))
        close(fused[k][0], want[k].numpy(), atol=6e-2 if "depth" in k else 2e-2, rtol=0,
              name="fused~oracle/%s V=%d" % (k, V))


def test_raw2outputs_uses_the_callers_dists(hip):
    """raw2outputs / raw2outputs_blending / raw2alpha take `dists` as the reference does (renderer.py:115,
    166, 91): spacings that did NOT come from depth2dist (random, no 1e10 tail) give the oracle's result
    for those spacings, forward and backward."""
    import zest_renderer as renderer
    from oracle import zest_oracle as zo
    comp = gc.composite_inputs(123, R=20, S=70, dead_ray=False)
    g = gc.zs.rng(124)
    dists = g.uniform(0.01, 0.2, size=comp["z"].shape).astype(np.float32)
    raw, z = torch.from_numpy(comp["raw"]), torch.from_numpy(comp["z"])
    want = zo.composite(raw, z, torch.from_numpy(dists), white_bkgd=True)
    with torch.no_grad():
        got = renderer.raw2outputs(G(comp["raw"])[None], G(comp["z"])[None], G(dists)[None], white_bkgd=True)
    for n, a, b in zip(("rgb_map", "disp_map", "acc_map", "weights", "depth_map", "alpha"), got, want):
        close(a[0], b.numpy(), name="raw2outputs(dists)/" + n)
    a, w = renderer.raw2alpha(torch.relu(G(comp["raw"])[None, ..., 3]), G(dists)[None])
    close(a[0], want[5].numpy(), name="raw2alpha(dists)/alpha")
    close(w[0], want[3].numpy(), name="raw2alpha(dists)/weights")
    # backward through the same spacings
    rg = G(comp["raw"])[None].requires_grad_(True)
    out = renderer.raw2outputs(rg, G(comp["z"])[None], G(dists)[None])
    (out[0].sum() + (out[3] * G(comp["z"])[None]).sum()).backward()
    rc = raw.clone().requires_grad_(True)
    o2 = zo.composite(rc, z, torch.from_numpy(dists))
    (o2[0].sum() + (o2[3] * z).sum()).backward()
    close(rg.grad[0], rc.grad.numpy(), atol=1e-4, rtol=1e-3, name="raw2outputs(dists) grad")
    # blending
    bl = gc.build("blend")
    d2 = g.uniform(0.01, 0.2, size=bl["z"].shape).astype(np.float32)
    wantb = zo.composite_blend(*[torch.from_numpy(bl[k]) for k in ("raw_dy", "raw_st", "blend", "z")], torch.from_numpy(d2))
    with torch.no_grad():
        gotb = renderer.raw2outputs_blending(G(bl["raw_dy"])[None], G(bl["raw_st"])[None], G(bl["blend"])[None],
                                             G(bl["z"])[None], G(d2)[None])
    for n, a, b in zip(("rgb_map", "depth_map", "rgb_map_fg", "depth_map_fg", "weights_fg", "weights_dy"), gotb, wantb):
        close(a[0], b.numpy(), name="raw2outputs_blending(dists)/" + n)


def test_raw2alpha_under_autograd(hip):
    """raw2alpha (reference renderer.py:91-113) with a density that wants a gradient: weights through CompositeFn
    (HIP backward), alpha through its defining ops - against the oracle's autograd on the same numbers."""
    import zest_renderer as renderer
    from oracle import zest_oracle as zo
    comp = gc.composite_inputs(31, R=12, S=20, dead_ray=False)
    z, d = torch.from_numpy(comp["z"]), torch.from_numpy(comp["rays_dir"])
    dists = zo.sample_dists(z, torch.linalg.vector_norm(d, dim=-1, keepdim=True))
    sig0 = torch.relu(torch.from_numpy(comp["raw"])[..., 3]) + 0.05
    ga = gc.zs.rng(3).standard_normal(tuple(sig0.shape)).astype(np.float32)
    gw = gc.zs.rng(4).standard_normal(tuple(sig0.shape)).astype(np.float32)
    so = sig0.clone().requires_grad_(True)
    alpha_o = 1.0 - torch.exp(-so * dists)
    w_o = alpha_o * torch.cumprod(torch.cat([torch.ones_like(alpha_o[:, :1]), 1.0 - alpha_o + 1e-10], -1), -1)[:, :-1]
    ((torch.from_numpy(ga) * alpha_o).sum() + (torch.from_numpy(gw) * w_o).sum()).backward()
    sg = G(sig0.numpy()).requires_grad_(True)
    a, w = renderer.raw2alpha(sg[None], G(dists.numpy())[None])
    ((G(ga) * a[0]).sum() + (G(gw) * w[0]).sum()).backward()
    close(a[0], alpha_o.detach().numpy(), name="raw2alpha/alpha (autograd)")
    close(w[0], w_o.detach().numpy(), name="raw2alpha/weights (autograd)")
    close(sg.grad, so.grad.numpy(), atol=1e-4, rtol=1e-3, name="raw2alpha/d sigma")
    with pytest.raises(NotImplementedError, match="sample spacings"):
        renderer.raw2alpha(sg[None], G(dists.numpy())[None].requires_grad_(True))
