#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference on CPU.

Runs only in the build container (needs /root/reference).  The reference modules
`renderer`, `utils`, `networks` import four packages that are absent here and are
not used on the rendering path (cv2, torchvision, kornia, inplace_abn; SURVEY.md
section 8(c)); empty placeholder modules are registered for them before import.
Inputs and weights come from the build's own seeded generator
(tests/golden_cases.py + zest-nerf_amd/zest_synth.py); only the reference's OUTPUTS
are written.  Nothing from the reference's source is copied.

    python tools/gen_golden.py            # all cases
    python tools/gen_golden.py render_    # cases whose name contains the substring
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import golden_cases as gc  # noqa: E402

REF = "/root/reference"


def _placeholders():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    mod("cv2").COLORMAP_JET = 2                        # default arg of an off-path viz helper
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms")
    tv.utils = mod("torchvision.utils")
    k = mod("kornia")
    k.utils = mod("kornia.utils")
    k.utils.create_meshgrid = lambda *a, **kw: None   # only homo_warp uses it (off-path)
    abn = mod("inplace_abn")

    class InPlaceABN(torch.nn.Module):                # default arg of off-path CNN blocks
        pass
    abn.InPlaceABN = InPlaceABN


def import_reference():
    _placeholders()
    sys.path.insert(0, REF)
    mods = {}
    for n in ("utils", "renderer", "networks", "losses"):
        sys.modules.pop(n, None)
    import renderer as r
    import utils as u
    import networks as nw
    import losses as ls
    mods.update(renderer=r, utils=u, networks=nw, losses=ls)
    sys.path.remove(REF)
    return SimpleNamespace(**mods)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def ref_net(ref, state, P, Fd, sceneflow, static, use_mvs, net_type="v0", D=8, W=256, skips=(4,)):
    net = ref.networks.MVSNeRF(D=D, W=W, input_ch_pts=P, output_ch=4, input_ch_views=gc.PE_DIR,
                               input_ch_feat=Fd, skips=list(skips), net_type=net_type,
                               sceneflow=sceneflow, static=static, use_mvs=use_mvs)
    missing = net.load_state_dict({k: T(v) for k, v in state.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.eval()


def run_case(ref, name):
    c = gc.CASES[name]
    inp = gc.build(name)
    k = c["kind"]
    out = {}
    with torch.set_grad_enabled(k == "render_grad"):
        if k == "composite":
            z, d = T(inp["z"])[None], T(inp["rays_dir"])[None]
            dists = ref.renderer.depth2dist(z, torch.norm(d, dim=-1, keepdim=True))
            r = ref.renderer.raw2outputs(T(inp["raw"])[None], z, dists,
                                         white_bkgd=c.get("white_bkgd", False))
            for n, v in zip(("rgb_map", "disp_map", "acc_map", "weights", "depth_map", "alpha"), r):
                out[n] = v[0].numpy()
            out["dists"] = dists[0].numpy()
        elif k == "blend":
            z, d = T(inp["z"])[None], T(inp["rays_dir"])[None]
            dists = ref.renderer.depth2dist(z, torch.norm(d, dim=-1, keepdim=True))
            r = ref.renderer.raw2outputs_blending(T(inp["raw_dy"])[None], T(inp["raw_st"])[None],
                                                  T(inp["blend"])[None], z, dists)
            for n, v in zip(("rgb_map", "depth_map", "rgb_map_fg", "depth_map_fg", "weights_fg",
                             "weights_dy"), r):
                out[n] = v[0].numpy()
        elif k == "embed":
            e = ref.networks.Embedding(c["C"], c["L"])
            out["y"] = e(T(inp["x"])).numpy()
            out["freq_bands"] = e.freq_bands.numpy()
        elif k == "volume":
            out["feat"] = ref.utils.index_point_feature(T(inp["volume"]), T(inp["ndc"])).numpy()[0]
        elif k == "color":
            poses = {"w2cs": T(inp["w2cs"]), "intrinsics": T(inp["intrinsics"])}
            out["colors"] = ref.utils.build_color_volume(T(inp["pts"]), poses, T(inp["imgs"]),
                                                         with_mask=True).numpy()[0]
        elif k == "mlp":
            net = ref_net(ref, inp["state"], inp["P"], inp["Fd"], inp["sceneflow"], inp["static"],
                          inp["use_mvs"], inp["net_type"], inp["D"], inp["W"], inp["skips"])
            out["y"] = net(T(inp["x"])).numpy()[0]
            if inp["use_mvs"] or inp["net_type"] == "v2":          # x carries the feature columns forward_alpha reads
                out["alpha_only"] = net.forward_alpha(T(inp["x"])[..., :inp["P"] + inp["Fd"]]).numpy()[0]
        elif k == "loss_side":
            pass
        elif k == "homo_warp":
            out = run_homo_warp(ref.utils, inp)
        elif k == "builder_nets":
            out = run_builder_nets(ref.networks, inp)
        elif k == "rays":
            out = run_rays(ref.utils, c, inp)
        elif k == "render":
            out = run_render(ref, c, inp)
    if k == "render_grad":
        out = run_render_grad(ref, c, inp)
    if k == "loss_side":
        out = run_loss_side(ref, inp)
    return out


def run_loss_side(ref, inp):
    """distortion_loss (losses.py:53-87) and projection_from_ndc (utils.py:527-539) with the
    reference's own autograd for the gradients."""
    w = T(inp["weights"]).requires_grad_(True)
    loss = ref.losses.distortion_loss(w, T(inp["t_vals"]))
    loss.backward()
    out = dict(distortion=np.array([float(loss)]), distortion_dw=w.grad[0].numpy().copy())
    w2, pts = T(inp["weights"]).requires_grad_(True), T(inp["pts"]).requires_grad_(True)
    uv = ref.utils.projection_from_ndc(T(inp["w2c"]), inp["H"], inp["W"], inp["f"], w2, pts)
    (uv * T(inp["gw"])).sum().backward()
    out.update(uv=uv[0].detach().numpy(), uv_dw=w2.grad[0].numpy(), uv_dpts=pts.grad[0].numpy())
    return out


class AbnStandIn(torch.nn.Module):
    """What is passed to the reference's CostRegNet / FeatureNet as their `norm_act` constructor argument (the classes
    take the norm as a parameter, networks.py:938-1034; their default, inplace_abn.InPlaceABN, is a CUDA extension that
    is not installed): batch norm followed by leaky ReLU(0.01), with InPlaceABN's constructor signature and parameter
    names (inplace_abn 1.1.0: eps 1e-5, momentum 0.1, activation "leaky_relu", activation_param 0.01).  The fixture
    therefore pins the reference's WIRING of the two stacks - layers, strides, paddings, skip additions, state-dict
    keys - under this reading of the norm; the norm's own arithmetic stays parity unpinned."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu", activation_param=0.01):
        super().__init__()
        self.eps, self.momentum, self.slope = eps, momentum, activation_param
        self.weight, self.bias = torch.nn.Parameter(torch.ones(num_features)), torch.nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def forward(self, x):
        y = torch.nn.functional.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training,
                                           self.momentum, self.eps)
        return torch.nn.functional.leaky_relu(y, self.slope)


def run_builder_nets(N, inp):
    """Reference CostRegNet(41) (networks.py:1003-1059) and FeatureNet (networks.py:962-1001), norm_act = AbnStandIn, on
    the seeded state dicts, in evaluation mode (running estimates) and in training mode (batch statistics)."""
    out = {}
    for name, cls, args, x in (("costreg", N.CostRegNet, (41,), T(inp["cost"])), ("feature", N.FeatureNet, (), T(inp["imgs"]))):
        for mode in ("eval", "train"):
            net = cls(*args, norm_act=AbnStandIn)
            missing = net.load_state_dict({k: T(v) for k, v in inp[name + "_state"].items()}, strict=True)
            assert not missing.missing_keys and not missing.unexpected_keys
            net.train(mode == "train")
            y = net(x)
            y = y[0] if isinstance(y, (tuple, list)) else y
            out["%s_%s" % (name, mode)] = y.numpy()
    return out


def run_homo_warp(U, inp):
    """The sampling half of utils.homo_warp (utils.py:91-98): source view 1's feature map and its
    resized image warped with a GIVEN grid.  The grid is an input here (computed by the oracle's
    plane_grid): the reference's own grid construction needs kornia.create_meshgrid, which is
    not installed, so that half stays unpinned (oracle/zest_oracle.py).  Also used by the GPU
    test against our utils.homo_warp."""
    from oracle import zest_oracle as zo
    pad = inp["pad"]
    feats, imgs = T(inp["feats"]), T(inp["imgs"])
    H, W = feats.shape[-2:]
    g = zo.plane_grid(T(inp["proj_mats"])[0, 1], T(inp["depth_values"])[0], H, W, pad)
    D, Hp, Wp = g.shape[:3]
    grid = g.view(1, D, Wp, Hp, 2)                   # the reference's shape label, memory order [D][y][x]
    warped, _ = U.homo_warp(feats[:, 1], T(inp["proj_mats"])[:, 1], T(inp["depth_values"]), src_grid=grid, pad=pad)
    img_lr = torch.nn.functional.interpolate(imgs[0], (H, W), mode="bilinear", align_corners=False)[1:2]
    img_warped, _ = U.homo_warp(img_lr, T(inp["proj_mats"])[:, 1], T(inp["depth_values"]), src_grid=grid, pad=pad)
    return dict(warped=warped[0].numpy(), img_warped=img_warped[0].numpy())


def run_render_grad(ref, c, sc):
    """Reference training-mode call with autograd: digests of dLoss/d(parameter | volume)."""
    grads = {}
    ret, leaves = run_render(ref, c, sc, want_grad=True)
    W = gc.loss_weights(c["seed"], {k: tuple(v.shape[1:]) for k, v in ret.items() if v is not None})
    loss = sum((T(W[k]) * ret[k][0]).sum() for k in W)
    loss.backward()
    for name, t in leaves.items():
        grads[name] = gc.grad_digest(t.grad.numpy())
    grads["__loss__"] = np.array([float(loss)])
    return grads


RAYS_OUT = ("point_samples", "rays_d", "color", "points_ndc", "depth_candidate", "rays_depth_gt", "t_vals",
            "rays_flow_fwd_gt", "rays_flow_bwd_gt", "rays_mask_fwd_gt", "rays_mask_bwd_gt")


def run_rays(U, c, inp, wrap=T):
    """Call build_rays_dy of module U (the reference, or the build's own utils in the tests) with
    the stratified jitter injected and the pixel RNG seeded."""
    sf = c.get("scene_flow", False)
    real_rand = torch.rand
    used = {}

    def fake_rand(shape, *a, **kw):
        n = wrap(inp["t_rand"][:shape[0]])
        assert tuple(n.shape) == tuple(shape), (n.shape, shape)
        used["n"] = shape[0]
        return n
    torch.manual_seed(c.get("torch_seed", 0))
    hook = getattr(U, "_draw_uniform", None)
    if hook is not None:
        U._draw_uniform = lambda shape, device: fake_rand(shape)
    else:
        torch.rand = fake_rand
    try:
        r = U.build_rays_dy(wrap(inp["imgs"]), wrap(inp["depths"]), wrap(inp["w2cs"]), wrap(inp["c2ws"]),
                            wrap(inp["intrinsics"]), wrap(inp["near_fars"]), inp["S"], N_rays=inp["R"],
                            stratified=c.get("stratified", True), pad=c.get("pad", 0),
                            chunk=c.get("chunk", -1), idx=c.get("idx", -1), val=not c.get("isRandom", True),
                            isRandom=c.get("isRandom", True), patch_size=c.get("patch_size", -1),
                            variable_patches=c.get("variable_patches", False), scale_anneal=c.get("scale_anneal", -1),
                            step=c.get("step", 0), scene_flow=sf,
                            flow_fwd=wrap(inp["flow_fwd"]) if sf else None, flow_bwd=wrap(inp["flow_bwd"]) if sf else None,
                            mask_fwd=wrap(inp["mask_fwd"]) if sf else None, mask_bwd=wrap(inp["mask_bwd"]) if sf else None,
                            num_extra_samples=c.get("num_extra_samples", 0),
                            motion_coords=torch.from_numpy(inp["motion_coords"]) if c.get("num_extra_samples", 0) else None)
    finally:
        torch.rand = real_rand
        if hook is not None:
            U._draw_uniform = hook
    return {n: v.detach().cpu().numpy() for n, v in zip(RAYS_OUT, r) if v is not None}


def run_render(ref, c, sc, want_grad=False):
    sf = c.get("scene_flow", False)
    args = SimpleNamespace(netchunk=1024, feat_dim=sc["feat_dim"], feat_dim_dy=sc["feat_dim_dy"],
                           img_downscale=1.0, use_color_volume=False, net_type="v0")
    e_pts = ref.networks.Embedding(3, 10)
    e_xyzt = ref.networks.Embedding(4, 10)
    e_dir = ref.networks.Embedding(3, 4)
    net_s = ref_net(ref, sc["state_static"], gc.PE_PTS + sc.get("time_dim", 0), sc["feat_dim"], sf, True, sc["use_mvs"],
                    "v0", *sc["static_shape"])
    tc = T(sc["time_codes"]).requires_grad_(want_grad) if sc.get("time_dim", 0) else None
    net_d = None
    if sf:
        net_d = ref_net(ref, sc["state_dynamic"], gc.PE_XYZT, 24, True, False, sc["use_mvs_dy"])
    leaves = {}
    if want_grad:
        for tag, net in (("static", net_s), ("dynamic", net_d)):
            if net is not None:
                net.train()
                for n_, p_ in net.named_parameters():
                    if tag == "static" and not sc["use_mvs"] and "pts_bias" in n_:
                        continue
                    leaves["%s.%s" % (tag, n_)] = p_
    vol_s = T(sc["vol_static"]).requires_grad_(want_grad) if sc["use_mvs"] else None
    vol_d = T(sc["vol_dynamic"]).requires_grad_(want_grad) if (sf and sc["use_mvs_dy"]) else None
    if want_grad:
        if tc is not None:
            leaves["time_codes"] = tc
        if vol_s is not None:
            leaves["vol_static"] = vol_s
        if vol_d is not None:
            leaves["vol_dynamic"] = vol_d
    cam = {"w2cs": T(sc["w2cs"]), "intrinsics": T(sc["intrinsics"])}
    nb_cam = None
    if sf and sc["use_mvs_dy"]:
        nb_cam = {"w2cs": T(sc["nb_w2cs"]), "intrinsics": T(sc["nb_intrinsics"])}
    std = c.get("raw_noise_std", 0)
    queue = [T(sc["noise_static"])[None], T(sc["noise_blend"])[None]] if sf else []
    real_randn = torch.randn

    def fake_randn(shape, *a, **kw):       # inject the two density-noise draws, in call order
        n = queue.pop(0)
        assert tuple(n.shape) == tuple(shape), (n.shape, shape)
        return n
    torch.randn = fake_randn
    try:
        ret = ref.renderer.rendering(
            args, T(sc["rays_pts"]), T(sc["rays_ndc"]), T(sc["depth_candidates"]), T(sc["rays_dir"]),
            volume_feature_static=vol_s, volume_feature_dynamic=vol_d,
            imgs=T(sc["imgs"]) if sc["use_mvs"] else None,
            neighbour_frames=T(sc["nb_imgs"]) if (sf and sc["use_mvs_dy"]) else None,
            im_cam_mat=cam, nb_cam_mat=nb_cam, network_fn=net_s, network_fn_dy=net_d,
            embedding_pts=e_pts, embedding_xyzt=e_xyzt, embedding_dir=e_dir,
            chain_bwd=c.get("chain_bwd", False), chain_5frames=c.get("chain_5frames", False),
            ref_frame_idx=gc.REF_FRAME_IDX, num_frames=gc.NUM_FRAMES,
            white_bkgd=c.get("white_bkgd", False), scene_flow=sf, val=c.get("val", False),
            raw_noise_std=std, time_codes=tc)
    finally:
        torch.randn = real_randn
    if want_grad:
        return ret, leaves
    out = {}
    for kk, v in ret.items():
        if v is not None:
            out[kk] = v[0].numpy()
    out["__keys__"] = np.array(sorted(ret.keys()))
    out["__none_keys__"] = np.array(sorted(kk for kk, v in ret.items() if v is None) or [""])
    return out


def main():
    sel = sys.argv[1] if len(sys.argv) > 1 else ""
    ref = import_reference()
    os.makedirs(gc.GOLDEN_DIR, exist_ok=True)
    for name in gc.CASES:
        if sel not in name:
            continue
        out = run_case(ref, name)
        path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-24s %3d arrays %7.1f KB" % (name, len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
