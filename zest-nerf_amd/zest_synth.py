"""Deterministic synthetic inputs for the ZeST-NeRF rendering hot path.

One seeded generator (numpy PCG64) shared by the golden-vector script
(tools/gen_golden.py), the parity tests, smoke() and bench.py, so that only seeds
and expected outputs ever need to be committed.  Shapes and value ranges follow
SURVEY.md section 8(d): un-normalised ray directions, depth samples between the
near/far planes, an 8-channel encoding volume, source images in [0,1], pinhole
cameras translated along x.

Nothing here is taken from the reference; the reference only defines the tensor
shapes the renderer consumes (/root/reference/renderer.py:579-587).
"""
from collections import OrderedDict

import numpy as np


def rng(seed):
    return np.random.Generator(np.random.PCG64(int(seed)))


# --------------------------------------------------------------------------- MLP
def mlp_layout(in_ch_pts, in_ch_views, in_ch_feat, sceneflow, static, use_mvs=True,
               D=8, W=256, skips=(4,)):
    """Parameter names and shapes of the width-W NeRF MLP as an ordered dict.

    Names match the state-dict contract of /root/reference/networks.py:93-123
    (``pts_linears.N``, ``pts_bias``, ``views_linears.0``, ``feature_linear``,
    ``alpha_linear``, ``rgb_linear``, ``w_linear`` | ``sf_linear``+``prob_linear``).
    The reference constructor yields D linear layers; the one following a skip
    index takes W + in_ch_pts inputs.
    """
    lay = OrderedDict()
    n = 0
    for i in range(D - 1):
        if i == 0:
            lay["pts_linears.%d" % n] = (W, in_ch_pts)
            n += 1
        fan_in = W + in_ch_pts if i in skips else W
        lay["pts_linears.%d" % n] = (W, fan_in)
        n += 1
    lay["pts_bias"] = (W, in_ch_feat)
    lay["views_linears.0"] = (W // 2, W + in_ch_views)
    lay["feature_linear"] = (W, W)
    lay["alpha_linear"] = (1, W)
    lay["rgb_linear"] = (3, W // 2)
    if sceneflow:
        if static:
            lay["w_linear"] = (1, W)
        else:
            lay["sf_linear"] = (6, W)
            lay["prob_linear"] = (2, W)
    return lay


def fill_mlp_state(layout, seed, prefix="nerf.", lively=True):
    """Seeded weights for ``layout`` -> {state_dict_key: float32 ndarray}.

    ``lively`` keeps activations O(1) through the 8 modulated ReLU layers
    (He-uniform weights; the modulation layer ``pts_bias`` gets a bias around 1)
    so that a numerical error anywhere upstream is visible at the outputs.
    With ``lively=False`` the scale is nn.Linear's default U(+-1/sqrt(fan_in)).
    """
    g = rng(seed)
    out = OrderedDict()
    for name, (fo, fi) in layout.items():
        bound_w = np.sqrt(6.0 / fi) if lively else 1.0 / np.sqrt(fi)
        bound_b = 1.0 / np.sqrt(fi)
        w = g.uniform(-bound_w, bound_w, size=(fo, fi)).astype(np.float32)
        b = g.uniform(-bound_b, bound_b, size=(fo,)).astype(np.float32)
        if name == "pts_bias" and lively:
            w = (w * 0.25).astype(np.float32)
            b = (1.0 + 0.5 * b).astype(np.float32)
        if name in ("sf_linear",) and lively:
            # keep predicted scene flow small so displaced points stay near the volume
            w = (w * 0.05).astype(np.float32)
            b = (b * 0.05).astype(np.float32)
        out[prefix + name + ".weight"] = w
        out[prefix + name + ".bias"] = b
    return out


# ----------------------------------------------------------------------- cameras
def make_cameras(n_views, H, W, focal, spread=0.2):
    """w2c [1,n,4,4] (identity rotation, x-translation) and K [1,n,3,3]."""
    w2cs = np.tile(np.eye(4, dtype=np.float32), (n_views, 1, 1))
    tx = np.linspace(-spread, spread, n_views).astype(np.float32) if n_views > 1 \
        else np.zeros(1, np.float32)
    # a small fixed rotation about y for odd views so R is not the identity
    for v in range(n_views):
        a = np.float32(0.02 * ((v % 3) - 1))
        c, s = np.cos(a), np.sin(a)
        w2cs[v, :3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
        w2cs[v, 0, 3] = tx[v]
    K = np.array([[focal, 0, W / 2.0], [0, focal, H / 2.0], [0, 0, 1]], np.float32)
    intr = np.tile(K, (n_views, 1, 1))
    return w2cs[None], intr[None]


def project_ndc(pts, w2c, K, W, H, near, far, pad):
    """World points -> [0,1]^3 volume coordinates of a reference view.

    Same geometry the reference applies before the renderer
    (/root/reference/utils.py:262-283): rigid transform, pinhole projection,
    divide by (W-1, H-1), depth normalised to [near, far], then the padded-volume
    rescale with feature-map size ((W-1)+1)/4 x ((H-1)+1)/4.
    """
    p = pts.reshape(-1, 3).astype(np.float64)
    pc = p @ w2c[:3, :3].astype(np.float64).T + w2c[:3, 3].astype(np.float64)
    q = pc @ K.astype(np.float64).T
    u = q[:, 0] / q[:, 2] / (W - 1)
    v = q[:, 1] / q[:, 2] / (H - 1)
    z = (q[:, 2] - near) / (far - near)
    if pad > 0:
        wf, hf = W / 4.0, H / 4.0
        u = u * wf / (wf + 2 * pad) + pad / (wf + 2 * pad)
        v = v * hf / (hf + 2 * pad) + pad / (hf + 2 * pad)
    return np.stack([u, v, z], -1).reshape(pts.shape).astype(np.float32)


# ------------------------------------------------------------------------- scene
def make_scene(seed, R, S, H=288, W=512, V=8, V_dy=4, pad=24, vol_depth=128,
               vol_hw=None, focal=400.0, near=2.0, far=6.0, static_volume=True,
               dynamic=False, stratified=True, ndc_mode="project", ray_mode="random", grid_start=0):
    """All tensors one ``rendering`` call consumes, as float32 numpy arrays.

    Returns a dict with rays_pts [1,R,S,3], rays_ndc [1,R,S,3],
    depth_candidates [1,R,S], rays_dir [1,R,3], and (optionally) vol_static
    [1,8,D,h,w], imgs [1,V,3,H,W], w2cs/intrinsics [1,V+1,...]; with ``dynamic``
    also vol_dynamic, nb_imgs [1,V_dy,3,H,W] and nb_w2cs/nb_intrinsics.

    ray_mode "random": one random direction per ray (the batch a training step draws); "grid": the rays of
    R consecutive pixels of the H x W target image in row-major order starting at pixel ``grid_start`` - the
    chunk a whole-image evaluation loop renders (/root/reference/networks.py:660-673), with its depth samples
    unjittered as there (``stratified`` is ignored).
    """
    g = rng(seed)
    sc = {"R": R, "S": S, "H": H, "W": W, "V": V, "V_dy": V_dy, "pad": pad}
    if vol_hw is None:
        vol_hw = (H // 4 + 2 * pad, W // 4 + 2 * pad)
    d = np.empty((1, R, 3), np.float32)
    if ray_mode == "grid":
        pix = (grid_start + np.arange(R)) % (H * W)
        d[0, :, 0] = ((pix % W) + 0.5 - W / 2.0) / focal
        d[0, :, 1] = ((pix // W) + 0.5 - H / 2.0) / focal
        stratified = False
    else:
        d[..., 0] = g.uniform(-0.5, 0.5, size=(1, R))
        d[..., 1] = g.uniform(-0.5, 0.5, size=(1, R))
    d[..., 2] = 1.0
    z = np.linspace(near, far, S, dtype=np.float32)[None, None, :].repeat(R, 1)
    if stratified and S > 1:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = np.concatenate([mids, z[..., -1:]], -1)
        lower = np.concatenate([z[..., :1], mids], -1)
        z = (lower + (upper - lower) * g.uniform(0, 1, size=z.shape)).astype(np.float32)
    o = np.array([0.05, -0.03, 0.0], np.float32)
    pts = (o[None, None, None, :] + z[..., None] * d[:, :, None, :]).astype(np.float32)
    w2cs, intr = make_cameras(V + 1, H, W, focal)
    if ndc_mode == "project":
        ndc = project_ndc(pts, w2cs[0, 0], intr[0, 0], W, H, near, far, pad)
    else:
        ndc = g.uniform(0, 1, size=pts.shape).astype(np.float32)
    sc.update(rays_pts=pts, rays_ndc=ndc, depth_candidates=z.astype(np.float32),
              rays_dir=d, w2cs=w2cs, intrinsics=intr)
    if static_volume:
        sc["vol_static"] = g.standard_normal((1, 8, vol_depth) + tuple(vol_hw),
                                             dtype=np.float32)
        sc["imgs"] = g.uniform(0, 1, size=(1, V, 3, H, W)).astype(np.float32)
    if dynamic:
        sc["vol_dynamic"] = g.standard_normal((1, 8, vol_depth) + tuple(vol_hw),
                                              dtype=np.float32)
        sc["nb_imgs"] = g.uniform(0, 1, size=(1, V_dy, 3, H, W)).astype(np.float32)
        nb_w2cs, nb_intr = make_cameras(V_dy, H, W, focal, spread=0.1)
        sc["nb_w2cs"], sc["nb_intrinsics"] = nb_w2cs, nb_intr
    return sc
