"""Run the CPU oracle on a golden case -> dict with the golden file's key names."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import golden_cases as gc  # noqa: E402
from oracle import zest_oracle as zo  # noqa: E402


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def state_t(state, dtype=torch.float32):
    return {k: T(v, dtype) for k, v in state.items()}


def spec_of(P, Fd, sceneflow, static, use_mvs, net_type="v0", D=8, W=256, skips=(4,)):
    return zo.MlpSpec(P, gc.PE_DIR, Fd, sceneflow=sceneflow, static=static, use_mvs=use_mvs,
                      net_type=net_type, D=D, W=W, skips=skips)


def render_nets(sc, dtype=torch.float32):
    sf = sc["scene_flow"]
    ns = zo.Net(state_t(sc["state_static"], dtype),
                spec_of(gc.PE_PTS + sc.get("time_dim", 0), sc["feat_dim"], sf, True, sc["use_mvs"], "v0",
                        *sc.get("static_shape", (8, 256, (4,)))))
    nd = None
    if sf:
        nd = zo.Net(state_t(sc["state_dynamic"], dtype),
                    spec_of(gc.PE_XYZT, 24, True, False, sc["use_mvs_dy"]))
    return ns, nd


def oracle_render(c, sc, dtype=torch.float32, explicit=True):
    sf = sc["scene_flow"]
    ns, nd = render_nets(sc, dtype)
    t = lambda k: T(sc[k], dtype)[0]
    cams = (t("w2cs"), t("intrinsics"))
    nb_cams = (t("nb_w2cs"), t("nb_intrinsics")) if (sf and sc["use_mvs_dy"]) else None
    noise = dict(static=T(sc["noise_static"], dtype), blend=T(sc["noise_blend"], dtype)) if sf else None
    return zo.rendering(
        t("rays_pts"), t("rays_ndc"), t("depth_candidates"), t("rays_dir"), ns, nd,
        vol_static=t("vol_static") if sc["use_mvs"] else None,
        vol_dynamic=t("vol_dynamic") if (sf and sc["use_mvs_dy"]) else None,
        imgs=t("imgs") if sc["use_mvs"] else None,
        nb_imgs=t("nb_imgs") if (sf and sc["use_mvs_dy"]) else None,
        cams=cams, nb_cams=nb_cams, scene_flow=sf, val=c.get("val", False),
        chain_bwd=c.get("chain_bwd", False), chain_5frames=c.get("chain_5frames", False),
        ref_frame_idx=gc.REF_FRAME_IDX, num_frames=gc.NUM_FRAMES,
        white_bkgd=c.get("white_bkgd", False), raw_noise_std=c.get("raw_noise_std", 0),
        noise=noise, explicit=explicit, time_codes=T(sc["time_codes"], dtype) if sc.get("time_dim", 0) else None)


def rays_pixels(c, inp):
    """Pixel selection with the reference's RNG call order (utils.py:172-212) for the test cases."""
    H, W, R = inp["H"], inp["W"], inp["R"]
    torch.manual_seed(c.get("torch_seed", 0))
    ps = c.get("patch_size", -1)
    if c.get("variable_patches", False):
        xs, ys = zo.graf_patch_pixels(H, W, ps, c.get("step", 0), c.get("scale_anneal", -1))
        xs, ys = xs.float(), ys.float()
    elif ps > 0:
        n = R // (ps * ps)
        xb, yb = torch.randint(0, W - ps, (n,)), torch.randint(0, H - ps, (n,))
        ar = torch.arange(ps, dtype=torch.float32)
        ys = (yb.float()[:, None, None] + ar[None, :, None]).expand(-1, -1, ps).reshape(-1)
        xs = (xb.float()[:, None, None] + ar[None, None, :]).expand(-1, ps, -1).reshape(-1)
    elif c.get("isRandom", True):
        xs, ys = torch.randint(0, W, (R,)).float(), torch.randint(0, H, (R,)).float()
    else:
        lin = torch.arange(c["idx"] * c["chunk"], min((c["idx"] + 1) * c["chunk"], H * W))
        ys, xs = (lin // W).float(), (lin % W).float()
    ne = c.get("num_extra_samples", 0)
    if ne:
        hard = torch.from_numpy(inp["motion_coords"])[torch.randint(0, inp["motion_coords"].shape[0], (ne,))]
        xs, ys = torch.cat([xs, hard[:, 1].float()]), torch.cat([ys, hard[:, 0].float()])
    return xs, ys


def run(case, dtype=torch.float32, explicit=True):
    c = gc.CASES[case]
    inp = gc.build(case)
    k = c["kind"]
    out = {}
    with torch.no_grad():
        if k == "composite":
            z, d = T(inp["z"], dtype), T(inp["rays_dir"], dtype)
            dists = zo.sample_dists(z, torch.linalg.vector_norm(d, dim=-1, keepdim=True))
            r = zo.composite(T(inp["raw"], dtype), z, dists, c.get("white_bkgd", False))
            out = dict(zip(("rgb_map", "disp_map", "acc_map", "weights", "depth_map", "alpha"), r))
            out["dists"] = dists
        elif k == "blend":
            z, d = T(inp["z"], dtype), T(inp["rays_dir"], dtype)
            dists = zo.sample_dists(z, torch.linalg.vector_norm(d, dim=-1, keepdim=True))
            r = zo.composite_blend(T(inp["raw_dy"], dtype), T(inp["raw_st"], dtype),
                                   T(inp["blend"], dtype), z, dists)
            out = dict(zip(("rgb_map", "depth_map", "rgb_map_fg", "depth_map_fg", "weights_fg",
                            "weights_dy"), r))
        elif k == "embed":
            out["y"] = zo.embed(T(inp["x"], dtype), c["L"])
        elif k == "volume":
            out["feat"] = zo.volume_lookup(T(inp["volume"], dtype)[0], T(inp["ndc"], dtype)[0], explicit)
        elif k == "color":
            out["colors"] = zo.color_lookup(T(inp["pts"], dtype)[0], T(inp["w2cs"], dtype)[0],
                                            T(inp["intrinsics"], dtype)[0], T(inp["imgs"], dtype)[0],
                                            explicit)
        elif k == "mlp":
            spec = spec_of(inp["P"], inp["Fd"], inp["sceneflow"], inp["static"], inp["use_mvs"],
                           inp["net_type"], inp["D"], inp["W"], inp["skips"])
            out["y"] = zo.mlp_forward(state_t(inp["state"], dtype), T(inp["x"], dtype)[0], spec)
            if inp["use_mvs"] or inp["net_type"] == "v2":
                out["alpha_only"] = zo.mlp_forward_alpha(state_t(inp["state"], dtype),
                                                         T(inp["x"], dtype)[0, :, :inp["P"] + inp["Fd"]], spec)
        elif k == "loss_side":
            pass                                    # needs autograd: below
        elif k == "homo_warp":
            feats, imgs = T(inp["feats"], dtype)[0], T(inp["imgs"], dtype)[0]
            H, W = feats.shape[-2:]
            g = zo.plane_grid(T(inp["proj_mats"], dtype)[0, 1], T(inp["depth_values"], dtype)[0], H, W, inp["pad"])
            img_lr = torch.nn.functional.interpolate(imgs, (H, W), mode="bilinear", align_corners=False)
            out = dict(warped=zo.grid_warp(feats[1], g), img_warped=zo.grid_warp(img_lr[1], g))
        elif k == "builder_nets":
            for name, fn, x in (("costreg", zo.cost_reg_net, inp["cost"]), ("feature", zo.feature_net, inp["imgs"])):
                st = {kk: T(v, dtype) for kk, v in inp[name + "_state"].items()}
                for mode in ("eval", "train"):
                    out["%s_%s" % (name, mode)] = fn(st, T(x, dtype), training=mode == "train")
        elif k == "rays":
            xs, ys = rays_pixels(c, inp)
            t = lambda a: T(a, dtype)[0]
            nf = inp["near_fars"][0]
            tr = T(inp["t_rand"], dtype)[:xs.shape[0]] if c.get("stratified", True) else None
            d, z, pts, ndc = zo.sample_rays(xs.to(dtype), ys.to(dtype), t(inp["intrinsics"])[-1], t(inp["c2ws"])[-1],
                                            t(inp["w2cs"])[0], t(inp["intrinsics"])[0], float(nf[-1, 0]),
                                            float(nf[-1, 1]), float(nf[0, 0]), float(nf[0, 1]), inp["S"], tr,
                                            c.get("pad", 0), inp["W"], inp["H"])
            yi, xi = ys.long(), xs.long()
            out = dict(point_samples=pts[None], rays_d=d[None], points_ndc=ndc[None], depth_candidate=z[None],
                       color=T(inp["imgs"], dtype)[:, -1, :, yi, xi].permute(0, 2, 1),
                       rays_depth_gt=T(inp["depths"], dtype)[:, -1, yi, xi],
                       t_vals=torch.linspace(0., 1., inp["S"], dtype=dtype)[None])
            if c.get("scene_flow"):
                out.update(rays_flow_fwd_gt=T(inp["flow_fwd"], dtype)[:, -1, :, yi, xi].permute(0, 2, 1),
                           rays_flow_bwd_gt=T(inp["flow_bwd"], dtype)[:, -1, :, yi, xi].permute(0, 2, 1),
                           rays_mask_fwd_gt=T(inp["mask_fwd"], dtype)[:, -1, yi, xi],
                           rays_mask_bwd_gt=T(inp["mask_bwd"], dtype)[:, -1, yi, xi])
        elif k == "render":
            out = oracle_render(c, inp, dtype, explicit)
    if k == "loss_side":
        out = loss_side(inp, dtype)
    return {kk: (v.double().numpy() if v is not None else None) for kk, v in out.items()}


def loss_side(inp, dtype=torch.float32):
    """Oracle distortion loss / ray projection with autograd gradients (fixture key names)."""
    w = T(inp["weights"], dtype)[0].requires_grad_(True)
    loss = zo.distortion_loss(w, T(inp["t_vals"], dtype))
    loss.backward()
    out = dict(distortion=loss.detach().reshape(1), distortion_dw=w.grad.clone())
    w2, pts = T(inp["weights"], dtype)[0].requires_grad_(True), T(inp["pts"], dtype)[0].requires_grad_(True)
    uv = zo.projection_from_ndc(T(inp["w2c"], dtype)[0], inp["H"], inp["W"], inp["f"], w2, pts)
    (uv * T(inp["gw"], dtype)[0]).sum().backward()
    out.update(uv=uv.detach(), uv_dw=w2.grad, uv_dpts=pts.grad)
    return out


def oracle_render_grads(case, dtype=torch.float32):
    """Train-mode oracle call with autograd.  -> (ret dict of tensors, {leaf name: grad ndarray})
    with the leaf naming of tools/gen_golden.py (static.nerf.<param>, dynamic.nerf.<param>,
    vol_static, vol_dynamic) and the loss of golden_cases.loss_weights."""
    c, sc = gc.CASES[case], gc.build(case)
    sf = sc["scene_flow"]
    ns, nd = render_nets(sc, dtype)
    leaves = {}
    for tag, net in (("static", ns), ("dynamic", nd)):
        if net is None:
            continue
        for k in list(net.state):
            if tag == "static" and not sc["use_mvs"] and "pts_bias" in k:
                continue
            net.state[k] = net.state[k].clone().requires_grad_(True)
            leaves["%s.%s" % (tag, k)] = net.state[k]
    t = lambda k: T(sc[k], dtype)[0]
    vol_s = t("vol_static").requires_grad_(True) if sc["use_mvs"] else None
    vol_d = t("vol_dynamic").requires_grad_(True) if (sf and sc["use_mvs_dy"]) else None
    tc = T(sc["time_codes"], dtype).requires_grad_(True) if sc.get("time_dim", 0) else None
    if tc is not None:
        leaves["time_codes"] = tc
    if vol_s is not None:
        leaves["vol_static"] = vol_s
    if vol_d is not None:
        leaves["vol_dynamic"] = vol_d
    cams = (t("w2cs"), t("intrinsics"))
    nb_cams = (t("nb_w2cs"), t("nb_intrinsics")) if (sf and sc["use_mvs_dy"]) else None
    ret = zo.rendering(t("rays_pts"), t("rays_ndc"), t("depth_candidates"), t("rays_dir"), ns, nd,
                       vol_static=vol_s, vol_dynamic=vol_d, imgs=t("imgs") if sc["use_mvs"] else None,
                       nb_imgs=t("nb_imgs") if (sf and sc["use_mvs_dy"]) else None, cams=cams, nb_cams=nb_cams,
                       scene_flow=sf, val=False, chain_bwd=c.get("chain_bwd", False),
                       chain_5frames=c.get("chain_5frames", False), ref_frame_idx=gc.REF_FRAME_IDX,
                       num_frames=gc.NUM_FRAMES, white_bkgd=c.get("white_bkgd", False), explicit=False,
                       time_codes=tc)
    W = gc.loss_weights(c["seed"], {k: tuple(v.shape) for k, v in ret.items() if v is not None})
    loss = sum((T(W[k], dtype) * ret[k]).sum() for k in W)
    loss.backward()
    return float(loss.detach()), {k: v.grad.double().numpy() for k, v in leaves.items()}
