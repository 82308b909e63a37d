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


def spec_of(P, Fd, sceneflow, static, use_mvs, net_type="v0"):
    return zo.MlpSpec(P, gc.PE_DIR, Fd, sceneflow=sceneflow, static=static, use_mvs=use_mvs,
                      net_type=net_type)


def render_nets(sc, dtype=torch.float32):
    sf = sc["scene_flow"]
    ns = zo.Net(state_t(sc["state_static"], dtype),
                spec_of(gc.PE_PTS, sc["feat_dim"], sf, True, sc["use_mvs"]))
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
        noise=noise, explicit=explicit)


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
                           inp["net_type"])
            out["y"] = zo.mlp_forward(state_t(inp["state"], dtype), T(inp["x"], dtype)[0], spec)
        elif k == "render":
            out = oracle_render(c, inp, dtype, explicit)
    return {kk: (v.double().numpy() if v is not None else None) for kk, v in out.items()}
