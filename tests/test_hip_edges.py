"""GPU: edge cases of the C ABI as the reference's tensor code meets them - empty batches (torch ops on
empty tensors return empty tensors: a chunked whole-image loop ends on one), a single sample per ray, a
single ray, and arguments the kernels refuse (they must say so, not fault)."""
import numpy as np
import pytest
import torch

import golden_cases as gc
from test_hip_ops import G, _mlp_setup

pytestmark = pytest.mark.gpu


def _z(*shape):
    return torch.zeros(*shape, device="cuda:0")


def test_empty_batches_return_empty_results(hip):
    import zest_hip as zh
    S = 16
    raw, z, d = _z(0, S, 4), _z(0, S), _z(0, 3)
    out = zh.composite(raw, z, d)
    assert [tuple(t.shape) for t in out] == [(0, 3), (0,), (0,), (0, S), (0,), (0, S)]
    out = zh.composite_blend(raw, raw, _z(0, S), z, d)
    assert out[0].shape == (0, 3) and out[4].shape == (0, S)
    assert zh.weighted_complement_sum(_z(0, S), _z(0, S)).shape == (0,)
    assert zh.embed(_z(0, S, 3), 10).shape == (0, S, 63)
    vol = zh.volume_to_cl(torch.rand(1, 8, 4, 6, 5, device="cuda:0"))
    assert zh.volume_lookup(vol, _z(0, S, 3)).shape == (0, S, 8)
    imgs = zh.images_to_cl(torch.rand(1, 3, 3, 12, 16, device="cuda:0"))
    cams = torch.eye(4, device="cuda:0").repeat(4, 1, 1)
    k = torch.eye(3, device="cuda:0").repeat(4, 1, 1)
    assert zh.color_lookup(imgs, cams, k, _z(0, S, 3)).shape[:2] == (0, S)
    x = zh.encode(_z(0, S, 3), _z(0, S, 3), _z(0, 3), vol_cl=vol, imgs_cl=imgs, w2cs=cams, intrinsics=k)
    assert x.shape == (0, S, 63 + 8 + 4 * 3 + 27)
    zhh, inp, desc, tab = _mlp_setup("mlp_static_mvs20")
    for prec in (zh.PREC_F32, zh.PREC_BF16, zh.PREC_F16X3):
        assert zh.mlp_fwd(desc, prec, zh.mlp_pack(desc, prec, tab), _z(0, desc.in_ch)).shape == (0, 4)
    loss, grad = zh.distortion(_z(0, S), torch.linspace(0, 1, S, device="cuda:0")[None])
    assert loss.shape == (0,) and grad.shape == (0, S)
    assert zh.sample_pdf(_z(0, S + 1), _z(0, S), n_samples=8).shape == (0, 8)
    torch.cuda.synchronize()


def test_empty_batch_through_rendering(hip):
    """rendering() on zero rays, fused and per-operator plans: every key of the result dict, empty."""
    from test_hip_render import render_scene
    sc = gc.render_inputs(77, R=4, S=16, V=3, use_mvs=True, scene_flow=True, use_mvs_dy=True)
    for k in ("rays_pts", "rays_ndc", "depth_candidates", "rays_dir"):
        sc[k] = sc[k][:, :0]
    sc["noise_static"], sc["noise_blend"] = sc["noise_static"][:0], sc["noise_blend"][:0]
    for maps_only in (True, False):
        ret = render_scene(sc, dict(val=True), precision=16, maps_only=maps_only)
        assert ret["rgb_map"].shape == (1, 0, 3) and ret["depth_map_ref"].shape == (1, 0)
    torch.cuda.synchronize()


def test_single_sample_rays_and_a_single_ray(hip):
    """S = 1 (the lone sample takes the reference's 1e10 interval: alpha is 0 or 1) and R = 1 against the oracle."""
    import zest_hip as zh
    from oracle import zest_oracle as zo
    g = gc.zs.rng(5)
    for R, S in ((5, 1), (1, 7), (1, 1)):
        raw = torch.from_numpy(g.standard_normal((R, S, 4)).astype(np.float32))
        z = torch.from_numpy(np.sort(g.uniform(2, 6, size=(R, S)).astype(np.float32), -1))
        d = torch.from_numpy(g.standard_normal((R, 3)).astype(np.float32))
        want = zo.composite(raw, z, zo.sample_dists(z, torch.norm(d, dim=-1, keepdim=True)))
        got = zh.composite(raw.cuda(), z.cuda(), d.cuda())
        for a, b, name in zip(got, want, ("rgb", "disp", "acc", "weights", "depth", "alpha")):
            if name == "disp":
                continue                        # 1 / max(1e-10, depth / acc): undefined where acc == 0
            np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-3, atol=1e-4, err_msg="%dx%d %s" % (R, S, name))


def test_refused_arguments_raise_with_a_message(hip):
    import zest_hip as zh
    with pytest.raises(RuntimeError, match="channels"):
        zh.mlp_fwd(*_bad_mlp_call(zh))                                        # one input column too many
    with pytest.raises(RuntimeError):
        zh.composite(_z(4, 8, 3), _z(4, 8), _z(4, 3))                       # raw must carry 4 channels
    with pytest.raises(RuntimeError):
        zh.composite(_z(4, 8, 4), _z(4, 9), _z(4, 3))                       # z of another length
    with pytest.raises(RuntimeError):
        zh.composite(torch.zeros(4, 8, 4), torch.zeros(4, 8), torch.zeros(4, 3))   # host tensors: no CPU path
    desc = zh.MlpDesc(63, 20, 27, 1, 0, zh.HEAD_NONE, 9, 256, 0)             # depth 9
    with pytest.raises(RuntimeError, match="depth"):
        zh.mlp_pack(desc, zh.PREC_F32, [_z(1)] * (2 * zh.P_COUNT))
    _, inp, good, tab = _mlp_setup("mlp_static_mvs20")
    with pytest.raises(RuntimeError, match="precision"):
        zh.render_fused(_z(2, 4, 3), None, _z(2, 4), _z(2, 3), good, _z(16), zh.make_view_set(), precision=zh.PREC_F32)
    with pytest.raises(RuntimeError, match="packed"):                          # weights packed for another precision
        zh.mlp_fwd(good, zh.PREC_F16X3, zh.mlp_pack(good, zh.PREC_BF16, tab), _z(8, good.in_ch))
    wide = list(tab)
    wide[2] = _z(256, 300)                                                   # pts_linears.1 of another architecture
    with pytest.raises(RuntimeError, match="slot 1 weight"):
        zh.mlp_pack(good, zh.PREC_BF16, wide)
    with pytest.raises(RuntimeError, match="rays_dir"):
        zh.encode(_z(4, 8, 3), _z(4, 8, 3), _z(5, 3))
    with pytest.raises(RuntimeError, match="ndc"):
        zh.volume_lookup(zh.volume_to_cl(torch.rand(1, 8, 4, 6, 5, device="cuda:0")), _z(4, 8, 2))


def _bad_mlp_call(zh):
    _, inp, desc, tab = _mlp_setup("mlp_static_mvs20")
    return desc, zh.PREC_F32, zh.mlp_pack(desc, zh.PREC_F32, tab), _z(8, desc.in_ch + 1)


def test_second_device_context(hip):
    """include/zest_render.h, "Devices": the library's own device-side state (gather tables of zest_mlp_pack, train16
    tables, rocBLAS handle) is keyed on the current device, so the same net renders on another device of the process.
    One-GPU boxes (the driver's test box) skip; the keying itself is also exercised by every test on device 0."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    import zest_hip as zh
    zhh, inp, desc, tab = _mlp_setup("mlp_static_mvs20")
    x0 = torch.rand(96, desc.in_ch, device="cuda:0")
    y0 = zh.mlp_fwd(desc, zh.PREC_BF16, zh.mlp_pack(desc, zh.PREC_BF16, tab), x0)
    with torch.cuda.device(1):
        tab1 = [t.to("cuda:1") if t is not None else None for t in tab]
        y1 = zh.mlp_fwd(desc, zh.PREC_BF16, zh.mlp_pack(desc, zh.PREC_BF16, tab1), x0.to("cuda:1"))
    assert torch.equal(y0.cpu(), y1.cpu())
