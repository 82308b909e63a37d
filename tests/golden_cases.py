"""Golden-vector case table shared by tools/gen_golden.py (which runs the unmodified
reference on CPU in the build container) and the parity tests (which run the oracle
and the HIP path on the same regenerated inputs).

Inputs are never stored: every case is rebuilt from its seed with the build's own
generator (zest-nerf_amd/zest_synth.py).  Only expected OUTPUTS are committed, as
tests/golden/<case>.npz.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "zest-nerf_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

import zest_synth as zs  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

PE_PTS, PE_XYZT, PE_DIR = 63, 84, 27


# ----------------------------------------------------------------- op-level cases
def composite_inputs(seed, R=6, S=16, dead_ray=True):
    g = zs.rng(seed)
    raw = (g.standard_normal((R, S, 4)) * 2.0).astype(np.float32)
    if dead_ray:
        raw[1, :, 3] = -1.0          # sigma <= 0 everywhere: zero weights
        raw[2, :, 3] = 40.0          # opaque at the first sample
    z = np.sort(g.uniform(2, 6, size=(R, S)).astype(np.float32), -1)
    d = np.stack([g.uniform(-.5, .5, R), g.uniform(-.5, .5, R), np.ones(R)], -1).astype(np.float32)
    return dict(raw=raw, z=z, rays_dir=d)


def blend_inputs(seed, R=5, S=24):
    g = zs.rng(seed)
    raw_dy = (g.standard_normal((R, S, 4)) * 2.0).astype(np.float32)
    raw_st = (g.standard_normal((R, S, 4)) * 2.0).astype(np.float32)
    blend = g.uniform(0, 1, size=(R, S)).astype(np.float32)
    blend[0] = 0.0
    blend[1] = 1.0
    z = np.sort(g.uniform(2, 6, size=(R, S)).astype(np.float32), -1)
    d = np.stack([g.uniform(-.5, .5, R), g.uniform(-.5, .5, R), np.ones(R)], -1).astype(np.float32)
    return dict(raw_dy=raw_dy, raw_st=raw_st, blend=blend, z=z, rays_dir=d)


def embed_inputs(seed, C, M=40):
    g = zs.rng(seed)
    x = g.uniform(-0.4, 1.3, size=(M, C)).astype(np.float32)
    x[0] = 0.0
    x[1] = 1.0
    x[2] = 0.999
    x[3] = -0.3
    return dict(x=x)


def volume_inputs(seed, dims=(6, 7, 9), M=200):
    g = zs.rng(seed)
    vol = g.standard_normal((1, 8) + tuple(dims), dtype=np.float32)
    ndc = g.uniform(-0.15, 1.15, size=(1, 4, M // 4, 3)).astype(np.float32)
    ndc[0, 0, 0] = (0.0, 0.0, 0.0)
    ndc[0, 0, 1] = (1.0, 1.0, 1.0)
    ndc[0, 0, 2] = (0.5, 0.5, 0.5)
    ndc[0, 0, 3] = (1.0, 0.0, 0.25)
    return dict(volume=vol, ndc=ndc)


def color_inputs(seed, V=3, H=10, W=14, R=6, S=20):
    g = zs.rng(seed)
    imgs = g.uniform(0, 1, size=(1, V, 3, H, W)).astype(np.float32)
    w2cs, intr = zs.make_cameras(V + 1, H, W, focal=12.0, spread=0.3)
    pts = np.empty((1, R, S, 3), np.float32)
    pts[..., 0] = g.uniform(-3.5, 3.5, size=(1, R, S))
    pts[..., 1] = g.uniform(-2.5, 2.5, size=(1, R, S))
    pts[..., 2] = g.uniform(1.5, 6.0, size=(1, R, S))
    pts[0, 0, 0] = (0.0, 0.0, -2.0)      # behind the cameras
    pts[0, 0, 1] = (40.0, 0.0, 3.0)      # far outside the frame (border clamp, mask 0)
    return dict(imgs=imgs, w2cs=w2cs, intrinsics=intr, pts=pts)


MLP_VARIANTS = {
    # name: (in_ch_pts, in_ch_feat, sceneflow, static, use_mvs, net_type)
    "static_mvs20": (PE_PTS, 20, False, True, True, "v0"),
    "static_nomvs": (PE_PTS, 20, False, True, False, "v0"),
    "static_sf_mvs40": (PE_PTS, 40, True, True, True, "v0"),
    "dynamic_mvs24": (PE_XYZT, 24, True, False, True, "v0"),
    "dynamic_nomvs": (PE_XYZT, 24, True, False, False, "v0"),
    "v2_mvs20": (PE_PTS, 20, False, True, True, "v2"),
    # other depths / widths / skips (reference networks.py:93-100, opt.py:54-57): trailing (D, W, skips)
    "d4w128_static_mvs20": (PE_PTS, 20, False, True, True, "v0", 4, 128, (1,)),
    "d6w64_dynamic_mvs24": (PE_XYZT, 24, True, False, True, "v0", 6, 64, (0, 3)),
    "d3w192_static_sf_nomvs": (PE_PTS, 20, True, True, False, "v0", 3, 192, ()),
    "d5w128_v2_mvs20": (PE_PTS, 20, False, True, True, "v2", 5, 128, (2,)),
}


def mlp_shape(variant):
    v = MLP_VARIANTS[variant]
    return (v[6], v[7], tuple(v[8])) if len(v) > 6 else (8, 256, (4,))


def mlp_inputs(seed, variant, M=64):
    P, Fd, sf, st, mvs, nt = MLP_VARIANTS[variant][:6]
    D, W, skips = mlp_shape(variant)
    lay = zs.mlp_layout(P, PE_DIR, Fd, sf and nt == "v0", st, mvs, D=D, W=W, skips=skips)
    state = zs.fill_mlp_state(lay, seed)
    g = zs.rng(seed + 1000)
    in_ch = P + (Fd if (mvs or nt == "v2") else 0) + PE_DIR
    x = g.uniform(-1, 1, size=(1, M, in_ch)).astype(np.float32)
    return dict(state=state, x=x, P=P, Fd=Fd, sceneflow=sf, static=st, use_mvs=mvs, net_type=nt,
                D=D, W=W, skips=skips)


# --------------------------------------------------------------- loss-side cases
def loss_inputs(seed, R=12, S=40, jitter=False):
    """Compositing-like weights, sorted normalised sample positions, NDC points around the
    frustum (some beyond the clamp of NDC2Euclidean) and a neighbour camera."""
    g = zs.rng(seed)
    w = g.uniform(0, 1, size=(1, R, S)).astype(np.float32)
    w = (w / w.sum(-1, keepdims=True) * g.uniform(0.3, 1.0, size=(1, R, 1))).astype(np.float32)
    t = np.linspace(0, 1, S, dtype=np.float32)[None]
    if jitter:
        t = np.sort(np.clip(t + g.uniform(-0.4, 0.4, size=(R, S)).astype(np.float32) / S, 0, 1), -1)
    pts = g.uniform(-1.0, 1.0, size=(1, R, S, 3)).astype(np.float32)
    pts[..., 2] = g.uniform(-1.2, 1.05, size=(1, R, S)).astype(np.float32)
    w2cs, _ = zs.make_cameras(3, 24, 32, focal=30.0)
    gw = g.standard_normal((1, R, 2)).astype(np.float32)
    return dict(weights=w, t_vals=t.astype(np.float32), pts=pts, w2c=w2cs[:, 2], H=24, W=32, f=30.0, gw=gw)


# --------------------------------------------------------------- plane-sweep cases
def cost_inputs(seed, V=3, H=18, W=24, D=6, pad=2, spread=0.35):
    """Feature maps, images and homographies src_proj @ ref_proj_inv at feature resolution for
    MVSNet.build_volume_cost / utils.homo_warp.  The spread moves part of the sweep out of the
    source frames so the masks, the zero padding and the counts are exercised."""
    g = zs.rng(seed)
    feats = g.standard_normal((1, V, 32, H, W)).astype(np.float32)
    imgs = g.uniform(0, 1, size=(1, V, 3, 4 * H, 4 * W)).astype(np.float32)
    w2cs, intr = zs.make_cameras(V, H, W, focal=0.9 * W, spread=spread)
    projs = []
    for v in range(V):
        P = np.eye(4)
        P[:3, :4] = intr[0, v].astype(np.float64) @ w2cs[0, v, :3, :4].astype(np.float64)
        projs.append(P)
    ref_inv = np.linalg.inv(projs[0])
    proj_mats = np.stack([(P @ ref_inv)[:3, :4] for P in projs]).astype(np.float32)[None]
    proj_mats[0, 0] = np.eye(4, dtype=np.float32)[:3]
    depth_values = np.linspace(1.5, 5.0, D, dtype=np.float32)[None]
    return dict(feats=feats, imgs=imgs, proj_mats=proj_mats, depth_values=depth_values, pad=pad)


def builder_net_inputs(seed, D=16, H=16, W=24):
    """Seeded state dicts (reference key names) and inputs for the two convolution stacks of the volume builder:
    CostRegNet(32 + 9) on a [1,41,D,H,W] cost volume and FeatureNet on three [3,4H,4W]... images.  The keys and shapes come
    from this build's modules (the generator loads the same dicts into the REFERENCE's classes with strict key
    matching, which pins the state-dict layout too)."""
    import zest_networks as networks
    g = zs.rng(seed)

    def fill(mod):
        st = {}
        for k, v in mod.state_dict().items():
            shp = tuple(v.shape)
            if k.endswith("num_batches_tracked"):
                continue
            if k.endswith("running_mean"):
                a = g.standard_normal(shp) * 0.1
            elif k.endswith("running_var"):
                a = g.uniform(0.5, 2.0, size=shp)
            elif len(shp) == 1 and k.endswith("weight"):
                a = g.uniform(0.5, 1.5, size=shp)
            elif len(shp) == 1:
                a = g.standard_normal(shp) * 0.2
            else:
                a = g.standard_normal(shp) / np.sqrt(np.prod(shp[1:]))
            st[k] = a.astype(np.float32)
        return st
    return dict(costreg_state=fill(networks.CostRegNet(41)), feature_state=fill(networks.FeatureNet()),
                cost=g.standard_normal((1, 41, D, H, W)).astype(np.float32),
                imgs=g.uniform(-2, 2, size=(3, 3, 4 * H, 4 * W)).astype(np.float32))


# --------------------------------------------------------------- rendering cases
def render_inputs(seed, R=32, S=16, V=3, use_mvs=True, scene_flow=False, use_mvs_dy=True,
                  lively=True, time_dim=0, static_shape=None):
    """Small scene (24x32 images, 8x10x12 volume) + seeded MLP weights.  time_dim > 0: the static net
    takes that many time-code channels after the encoded point (Neural3D video mode, reference
    train.py:91-113, renderer.py:269-273) and the scene carries one latent code `time_codes` [1,T].
    static_shape: (D, W, skips) of the static net when not the shipped 8 / 256 / (4,)."""
    sc = zs.make_scene(seed, R, S, H=24, W=32, V=V, V_dy=4, pad=2, vol_depth=8, focal=30.0,
                       static_volume=use_mvs, dynamic=scene_flow)
    feat_dim = 8 + 4 * V
    D, W, skips = static_shape or (8, 256, (4,))
    lay_s = zs.mlp_layout(PE_PTS + time_dim, PE_DIR, feat_dim, scene_flow, True, use_mvs, D=D, W=W, skips=skips)
    sc["state_static"] = zs.fill_mlp_state(lay_s, seed + 1, lively=lively)
    sc.update(feat_dim=feat_dim, feat_dim_dy=24, use_mvs=use_mvs, use_mvs_dy=use_mvs_dy,
              scene_flow=scene_flow, time_dim=time_dim, static_shape=(D, W, tuple(skips)))
    if time_dim:
        sc["time_codes"] = zs.rng(seed + 4).standard_normal((1, time_dim)).astype(np.float32)
    if scene_flow:
        lay_d = zs.mlp_layout(PE_XYZT, PE_DIR, 24, True, False, use_mvs_dy)
        sc["state_dynamic"] = zs.fill_mlp_state(lay_d, seed + 2, lively=lively)
        g = zs.rng(seed + 3)
        sc["noise_static"] = g.standard_normal((R, S)).astype(np.float32)
        sc["noise_blend"] = g.standard_normal((R, S)).astype(np.float32)
        if not use_mvs_dy:
            sc.pop("vol_dynamic")
    return sc


def rays_inputs(seed, R=48, S=12, V=3, H=24, W=32):
    """A batch dict as the generators hand it to build_rays[_dy]."""
    g = zs.rng(seed)
    w2cs, intr = zs.make_cameras(V + 1, H, W, focal=30.0)
    c2ws = np.linalg.inv(w2cs[0].astype(np.float64)).astype(np.float32)[None]
    nf = np.tile(np.array([2.0, 6.0], np.float32), (1, V + 1, 1))
    nf[0, -1] = (2.25, 5.5)                       # target planes differ from the reference view's
    return dict(imgs=g.uniform(0, 1, size=(1, V + 1, 3, H, W)).astype(np.float32),
                depths=g.uniform(0.1, 1, size=(1, V + 1, H, W)).astype(np.float32), w2cs=w2cs, c2ws=c2ws,
                intrinsics=intr, near_fars=nf, t_rand=g.uniform(0, 1, size=(R + 8, S)).astype(np.float32),
                flow_fwd=g.standard_normal((1, 1, 2, H, W)).astype(np.float32),
                flow_bwd=g.standard_normal((1, 1, 2, H, W)).astype(np.float32),
                mask_fwd=(g.uniform(size=(1, 1, H, W)) > 0.5).astype(np.float32),
                mask_bwd=(g.uniform(size=(1, 1, H, W)) > 0.5).astype(np.float32),
                motion_coords=np.stack([g.integers(0, H, 20), g.integers(0, W, 20)], -1).astype(np.int64),
                R=R, S=S, H=H, W=W)


CASES = {
    "composite": dict(kind="composite", seed=11),
    "composite_white": dict(kind="composite", seed=12, white_bkgd=True),
    "blend": dict(kind="blend", seed=13),
    "embed3x10": dict(kind="embed", seed=14, C=3, L=10),
    "embed4x10": dict(kind="embed", seed=15, C=4, L=10),
    "embed3x4": dict(kind="embed", seed=16, C=3, L=4),
    "volume": dict(kind="volume", seed=17),
    "color": dict(kind="color", seed=18),
    **{"mlp_" + k: dict(kind="mlp", seed=20 + i, variant=k) for i, k in enumerate(MLP_VARIANTS)},
    "rays_random": dict(kind="rays", seed=51, pad=2, stratified=True, torch_seed=7),
    "rays_grid_chunk": dict(kind="rays", seed=52, pad=0, stratified=False, isRandom=False, chunk=40, idx=3),
    "rays_dy_motion": dict(kind="rays", seed=53, pad=2, stratified=True, torch_seed=11, scene_flow=True,
                           num_extra_samples=8),
    "rays_patches": dict(kind="rays", seed=54, pad=2, stratified=True, torch_seed=5, patch_size=4),
    "rays_graf_patches": dict(kind="rays", seed=55, pad=2, stratified=True, torch_seed=9, patch_size=4, R=16,
                              variable_patches=True, scale_anneal=0.0025, step=3000),      # GRAF: N_rays = patch_size^2
    "homo_warp": dict(kind="homo_warp", seed=63, pad=3),
    "builder_nets": dict(kind="builder_nets", seed=64),
    "loss_side": dict(kind="loss_side", seed=71),
    "render_static_mvs": dict(kind="render", seed=31, use_mvs=True),
    "render_static_nomvs": dict(kind="render", seed=32, use_mvs=False),
    "render_static_white": dict(kind="render", seed=33, use_mvs=True, white_bkgd=True),
    "render_zest_val": dict(kind="render", seed=34, scene_flow=True, val=True),
    "render_zest_train": dict(kind="render", seed=35, scene_flow=True),
    "render_zest_bwd": dict(kind="render", seed=36, scene_flow=True, chain_bwd=True),
    "render_zest_5f": dict(kind="render", seed=37, scene_flow=True, chain_5frames=True),
    "render_zest_bwd5f": dict(kind="render", seed=38, scene_flow=True, chain_bwd=True,
                              chain_5frames=True),
    "render_zest_noise": dict(kind="render", seed=39, scene_flow=True, chain_5frames=True,
                              raw_noise_std=1.0),
    "render_zest_nomvsdy": dict(kind="render", seed=40, scene_flow=True, val=True,
                                use_mvs_dy=False),
    "grad_zest_5f": dict(kind="render_grad", seed=41, scene_flow=True, chain_5frames=True),
    "grad_static": dict(kind="render_grad", seed=42, use_mvs=True, white_bkgd=True),
    "render_static_timecodes": dict(kind="render", seed=43, use_mvs=True, time_dim=8),
    "grad_static_timecodes": dict(kind="render_grad", seed=44, use_mvs=True, time_dim=8),
    # static net of another depth / width / skips (reference opt.py:54-57), with time codes through both skip layers
    "render_static_d5w128": dict(kind="render", seed=45, use_mvs=True, time_dim=4, static_shape=(5, 128, (1, 3))),
    "grad_static_d5w128": dict(kind="render_grad", seed=46, use_mvs=True, time_dim=4, static_shape=(5, 128, (1, 3))),
}

REF_FRAME_IDX, NUM_FRAMES = 0.1, 24

# outputs of a train-mode rendering() call that carry gradients; the gradient tests use
# loss = sum_k <W_k, out_k> with W_k ~ N(0,1) drawn in this order from rng(seed + 777)
LOSS_KEYS = ["rgb_map", "depth_map", "weights", "raw_blend_w", "rgb_map_ref", "depth_map_ref", "rgb_map_ref_dy",
             "depth_map_ref_dy", "raw_sf_ref2prev", "raw_sf_ref2post", "weights_ref_dy", "raw_prob_ref2prev",
             "raw_prob_ref2post", "raw_pts_prev", "raw_sf_prev2ref", "rgb_map_prev_dy", "raw_pts_post",
             "raw_sf_post2ref", "rgb_map_post_dy", "prob_map_prev", "prob_map_post", "raw_pts_pp", "rgb_map_pp_dy"]


def loss_weights(seed, shapes):
    """{key: ndarray} of fixed loss weights for the keys of `shapes` (dict key -> shape, no batch dim)."""
    g = zs.rng(seed + 777)
    return {k: g.standard_normal(shapes[k]).astype(np.float32) for k in LOSS_KEYS if k in shapes}


def grad_digest(t):
    """Compact fingerprint of a gradient tensor: (sum, L2 norm, dot with a fixed N(0,1) vector)."""
    a = np.asarray(t, np.float64).reshape(-1)
    r = zs.rng(a.size).standard_normal(a.size)
    return np.array([a.sum(), np.sqrt((a * a).sum()), (a * r).sum()])


def build(case):
    c = CASES[case]
    k = c["kind"]
    if k == "composite":
        return composite_inputs(c["seed"])
    if k == "blend":
        return blend_inputs(c["seed"])
    if k == "embed":
        return embed_inputs(c["seed"], c["C"])
    if k == "volume":
        return volume_inputs(c["seed"])
    if k == "color":
        return color_inputs(c["seed"])
    if k == "mlp":
        return mlp_inputs(c["seed"], c["variant"])
    if k == "loss_side":
        return loss_inputs(c["seed"], jitter=c.get("jitter", False))
    if k == "homo_warp":
        return cost_inputs(c["seed"], V=c.get("V", 3), pad=c["pad"])
    if k == "builder_nets":
        return builder_net_inputs(c["seed"])
    if k == "rays":
        return rays_inputs(c["seed"], R=c.get("R", 48))
    if k in ("render", "render_grad"):
        return render_inputs(c["seed"], use_mvs=c.get("use_mvs", True),
                             scene_flow=c.get("scene_flow", False),
                             use_mvs_dy=c.get("use_mvs_dy", True), time_dim=c.get("time_dim", 0),
                             static_shape=c.get("static_shape"))
    raise KeyError(k)


def load_golden(case):
    with np.load(os.path.join(GOLDEN_DIR, case + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}
