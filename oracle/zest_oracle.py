"""ORACLE — test infrastructure, NOT product code.

CPU restatement (PyTorch CPU ops, fp32 by default, fp64 on request) of the
ZeST-NeRF volume-rendering hot path, written from the behaviour of the reference
and pinned against it by tests/golden/*.npz, which tools/gen_golden.py produced
by running the unmodified reference on CPU in the build container
(tests/test_oracle_golden.py re-checks every vector).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (zest-nerf_amd/) never does and fails loudly when
its HIP library is missing.

Each function cites the reference lines whose behaviour it restates.  Internally
everything is flat: R rays, S samples, no leading image-batch dimension (the
reference always runs with N=1, train.py:307); `rendering` adds it back.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------- encoding (a4)
def embed(x, n_freqs):
    """[..., C] -> [..., C*(2*n_freqs+1)]: x, then sin/cos at 2^k, k=0..n_freqs-1.

    Restates Embedding.forward, /root/reference/networks.py:48-65 with the
    log-scale bands of :44 (exact powers of two).
    """
    parts = [x]
    for k in range(n_freqs):
        xs = x * float(2 ** k)
        parts.append(torch.sin(xs))
        parts.append(torch.cos(xs))
    return torch.cat(parts, -1)


# ------------------------------------------------------- encoding volume (a7)
def volume_lookup(volume, ndc, explicit=True):
    """Trilinear lookup with zero padding.  volume [C,D,H,W], ndc [...,3] in
    volume coordinates (x->W, y->H, z->D, 0..1 spans first..last voxel centre).

    Restates index_point_feature, /root/reference/utils.py:433-459
    (grid = ndc*2-1, grid_sample bilinear, align_corners=True, zeros padding).
    """
    C, D, H, W = volume.shape
    lead = ndc.shape[:-1]
    p = ndc.reshape(-1, 3)
    if not explicit:
        grid = (p * 2 - 1.0).view(1, 1, 1, -1, 3)
        out = F.grid_sample(volume[None], grid, mode="bilinear", align_corners=True)
        return out[0, :, 0, 0].t().reshape(*lead, C)
    g = p * 2 - 1.0                                    # same rounding as the reference
    fx = (g[:, 0] + 1) / 2 * (W - 1)
    fy = (g[:, 1] + 1) / 2 * (H - 1)
    fz = (g[:, 2] + 1) / 2 * (D - 1)
    x0, y0, z0 = torch.floor(fx), torch.floor(fy), torch.floor(fz)
    tx, ty, tz = fx - x0, fy - y0, fz - z0
    flat = volume.reshape(C, -1)
    acc = torch.zeros(p.shape[0], C, dtype=volume.dtype)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xi, yi, zi = x0 + dx, y0 + dy, z0 + dz
                w = (tx if dx else 1 - tx) * (ty if dy else 1 - ty) * (tz if dz else 1 - tz)
                ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1) & \
                     (zi >= 0) & (zi <= D - 1)
                idx = (zi.clamp(0, D - 1) * H + yi.clamp(0, H - 1)) * W + xi.clamp(0, W - 1)
                vals = flat[:, idx.long()].t()
                acc = acc + vals * (w * ok.to(w.dtype))[:, None]
    return acc.reshape(*lead, C)


# ----------------------------------------------------------- colour gather (a8/a9)
def color_lookup(pts, w2cs, intrinsics, imgs, explicit=True):
    """Per source view: project, bilinear RGB with border clamp, strict in-frame mask.

    pts [...,3] world; w2cs [V+,4,4]; intrinsics [V+,3,3]; imgs [V,3,H,W].
    Returns [..., 4V] laid out (r,g,b,mask) per view.
    Restates build_color_volume(with_mask=True), /root/reference/utils.py:461-505,
    and the projection of get_ndc_coordinate, utils.py:262-269 (pad=0).
    """
    V, _, H, W = imgs.shape
    lead = pts.shape[:-1]
    p = pts.reshape(-1, 3)
    out = torch.empty(p.shape[0], 4 * V, dtype=imgs.dtype)
    for v in range(V):
        Rm, T = w2cs[v, :3, :3], w2cs[v, :3, 3]
        pc = p @ Rm.t() + T
        q = pc @ intrinsics[v].t()
        u = (q[:, 0] / q[:, 2] + 0.0) / (W - 1)
        w_ = (q[:, 1] / q[:, 2] + 0.0) / (H - 1)
        gx, gy = u * 2.0 - 1.0, w_ * 2.0 - 1.0
        if explicit:
            fx = ((gx + 1) / 2 * (W - 1)).clamp(0, W - 1)
            fy = ((gy + 1) / 2 * (H - 1)).clamp(0, H - 1)
            x0, y0 = torch.floor(fx), torch.floor(fy)
            tx, ty = fx - x0, fy - y0
            x1, y1 = (x0 + 1).clamp(max=W - 1), (y0 + 1).clamp(max=H - 1)
            im = imgs[v].reshape(3, -1)

            def tap(xx, yy):
                return im[:, (yy * W + xx).long()].t()
            rgb = tap(x0, y0) * ((1 - tx) * (1 - ty))[:, None] + tap(x1, y0) * (tx * (1 - ty))[:, None] \
                + tap(x0, y1) * ((1 - tx) * ty)[:, None] + tap(x1, y1) * (tx * ty)[:, None]
        else:
            grid = torch.stack([gx, gy], -1).view(1, 1, -1, 2)
            rgb = F.grid_sample(imgs[v:v + 1], grid, mode="bilinear", align_corners=True,
                                padding_mode="border")[0, :, 0].t()
        mask = ((gx > -1.0) & (gx < 1.0) & (gy > -1.0) & (gy < 1.0)).to(imgs.dtype)
        out[:, 4 * v:4 * v + 3] = rgb
        out[:, 4 * v + 3] = mask
    return out.reshape(*lead, 4 * V)


# -------------------------------------------------------------------- MLP (a11)
class MlpSpec:
    """Static description of one width-W MLP (what the reference passes to
    MVSNeRF.__init__, /root/reference/networks.py:322-346)."""

    def __init__(self, in_ch_pts, in_ch_views, in_ch_feat, sceneflow=False, static=True,
                 use_mvs=True, net_type="v0", D=8, W=256, skips=(4,)):
        self.in_ch_pts, self.in_ch_views, self.in_ch_feat = in_ch_pts, in_ch_views, in_ch_feat
        self.sceneflow, self.static, self.use_mvs = sceneflow, static, use_mvs
        self.net_type, self.D, self.W, self.skips = net_type, D, W, tuple(skips)

    @property
    def in_ch(self):
        return self.in_ch_pts + (self.in_ch_feat if self._has_feat() else 0) + self.in_ch_views

    def _has_feat(self):
        return self.use_mvs or self.net_type == "v2"

    @property
    def out_ch(self):
        if self.net_type == "v2" or not self.sceneflow:
            return 4
        return 5 if self.static else 12


def mlp_forward(state, x, spec, prefix="nerf."):
    """x [M, in_ch] -> [M, out_ch] = (rgb raw, sigma raw, extras).

    v0 restates Renderer.forward (/root/reference/networks.py:150-221): trunk layers
    ``relu(Linear(h) * m)`` with m = pts_bias(feats) when use_mvs, skip re-concat
    of the encoded point after layer 4, heads sigmoid(w) | tanh(sf), sigmoid(prob),
    raw alpha, feature->views->rgb.  v2 restates Renderer_linear.forward
    (networks.py:283-319): additive modulation, relu on alpha, sigmoid on rgb.
    """
    def lin(name, h):
        return F.linear(h, state[prefix + name + ".weight"], state[prefix + name + ".bias"])

    P, Fd, Vw = spec.in_ch_pts, spec.in_ch_feat, spec.in_ch_views
    if spec._has_feat():
        pts, feats, views = x[:, :P], x[:, P:P + Fd], x[:, P + Fd:P + Fd + Vw]
        m = lin("pts_bias", feats)
    else:
        pts, views = x[:, :P], x[:, P:P + Vw]
        m = None
    h = pts
    for i in range(spec.D):
        h = lin("pts_linears.%d" % i, h)
        if m is not None:
            h = h * m if spec.net_type == "v0" else h + m
        h = torch.relu(h)
        if i in spec.skips:
            h = torch.cat([pts, h], -1)
    extras = []
    if spec.net_type == "v0" and spec.sceneflow:
        if spec.static:
            extras = [torch.sigmoid(lin("w_linear", h))]
        else:
            extras = [torch.tanh(lin("sf_linear", h)), torch.sigmoid(lin("prob_linear", h))]
    alpha = lin("alpha_linear", h)
    if spec.net_type == "v2":
        alpha = torch.relu(alpha)
    g = torch.cat([lin("feature_linear", h), views], -1)
    g = torch.relu(lin("views_linears.0", g))
    rgb = lin("rgb_linear", g)
    if spec.net_type == "v2":
        rgb = torch.sigmoid(rgb)
    return torch.cat([rgb, alpha] + extras, -1)


def mlp_forward_alpha(state, x, spec, prefix="nerf."):
    """Density-only pass, x [M, in_ch_pts + in_ch_feat] -> [M, 1].  Restates Renderer.forward_alpha
    (/root/reference/networks.py:134-147: multiplicative modulation, relu on alpha) and
    Renderer_linear.forward_alpha (networks.py:266-280: additive modulation, raw alpha); both always
    modulate with pts_bias(features), whatever use_mvs says."""
    def lin(name, h):
        return F.linear(h, state[prefix + name + ".weight"], state[prefix + name + ".bias"])

    P = spec.in_ch_pts
    pts, feats = x[:, :P], x[:, P:P + spec.in_ch_feat]
    m = lin("pts_bias", feats)
    h = pts
    for i in range(spec.D):
        h = lin("pts_linears.%d" % i, h)
        h = torch.relu(h * m if spec.net_type == "v0" else h + m)
        if i in spec.skips:
            h = torch.cat([pts, h], -1)
    alpha = lin("alpha_linear", h)
    return torch.relu(alpha) if spec.net_type == "v0" else alpha


# ------------------------------------------------------------- compositing (a2,a12,a13)
def sample_dists(z, dir_norm):
    """z [R,S], dir_norm [R,1] -> [R,S]; last interval 1e10, all scaled by |d|.
    Restates depth2dist, /root/reference/renderer.py:74-89."""
    d = z[:, 1:] - z[:, :-1]
    d = torch.cat([d, torch.full_like(z[:, :1], 1e10)], -1)
    return d * dir_norm


def _excl_cumprod(x):
    one = torch.ones_like(x[:, :1])
    return torch.cumprod(torch.cat([one, x], -1), -1)[:, :-1]


def composite(raw, z, dists, white_bkgd=False, noise=None):
    """raw [R,S,4] -> rgb_map [R,3], disp [R], acc [R], weights [R,S], depth [R], alpha [R,S].
    Restates raw2outputs + raw2alpha, /root/reference/renderer.py:91-164
    (transmittance uses 1 - alpha + 1e-10)."""
    rgb = torch.sigmoid(raw[..., :3])
    sig = raw[..., 3] if noise is None else raw[..., 3] + noise
    sig = torch.relu(sig)
    alpha = 1.0 - torch.exp(-sig * dists)
    w = alpha * _excl_cumprod(1.0 - alpha + 1e-10)
    rgb_map = (w[..., None] * rgb).sum(-2)
    depth = (w * z).sum(-1)
    acc = w.sum(-1)
    disp = 1.0 / torch.maximum(torch.full_like(depth, 1e-10), depth / acc)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc[..., None])
    return rgb_map, disp, acc, w, depth, alpha


def composite_blend(raw_dy, raw_st, blend, z, dists, noise=None):
    """Static/dynamic blended compositing plus the dynamic-only render.
    Returns rgb_map, depth_map, rgb_map_fg, depth_map_fg, weights_fg, weights_dy.
    Restates raw2outputs_blending, /root/reference/renderer.py:166-219."""
    rgb_d, rgb_s = torch.sigmoid(raw_dy[..., :3]), torch.sigmoid(raw_st[..., :3])
    n = 0.0 if noise is None else noise
    sd, ss = torch.relu(raw_dy[..., 3] + n), torch.relu(raw_st[..., 3] + n)
    a_fg = 1.0 - torch.exp(-sd * dists)
    a_d = a_fg * blend
    a_s = (1.0 - torch.exp(-ss * dists)) * (1.0 - blend)
    T = _excl_cumprod((1.0 - a_d) * (1.0 - a_s) + 1e-10)
    w_d, w_s = T * a_d, T * a_s
    rgb_map = (w_d[..., None] * rgb_d + w_s[..., None] * rgb_s).sum(-2)
    depth = ((w_d + w_s) * z).sum(-1)
    w_fg = a_fg * _excl_cumprod(1.0 - a_fg + 1e-10)
    return rgb_map, depth, (w_fg[..., None] * rgb_d).sum(-2), (w_fg * z).sum(-1), w_fg, w_d


# -------------------------------------------------------------- orchestration (a1,a3,a5,a14,a15)
class Net:
    """An MLP (state dict + spec) with its encoders' frequency counts."""

    def __init__(self, state, spec, n_freq_pts=10, n_freq_dir=4):
        self.state, self.spec = state, spec
        self.n_freq_pts, self.n_freq_dir = n_freq_pts, n_freq_dir


def build_mlp_input(net, pts_world, ndc, view_dir, volume=None, imgs=None, cams=None,
                    frame_idx=None, explicit=True, time_codes=None):
    """Assemble the per-sample MLP input [R,S,in_ch] = PE(point[,t]) | features | PE(dir).

    pts_world, ndc [R,S,3]; view_dir [R,3] already rotated into the reference camera.
    Restates prepare_pts / prepare_dynamic_pts / gen_pts_feats,
    /root/reference/renderer.py:51-72,246-318: the time index is appended as a 4th
    coordinate before encoding; the volume is looked up at ndc (without t) and the
    colours at the un-displaced world points.  time_codes [1,T] or [T] (Neural3D video mode,
    renderer.py:269-273): sigmoid of the frame's latent code, the same for every sample, right
    after the encoded point.
    """
    R, S, _ = ndc.shape
    p = ndc
    if frame_idx is not None:
        p = torch.cat([ndc, torch.full_like(ndc[..., :1], frame_idx)], -1)
    cols = [embed(p, net.n_freq_pts)]
    if time_codes is not None:
        cols.append(torch.sigmoid(time_codes).reshape(1, 1, -1).expand(R, S, -1))
    feats = None
    if volume is not None:
        f8 = volume_lookup(volume, ndc, explicit)
        fc = color_lookup(pts_world, cams[0], cams[1], imgs, explicit)
        feats = torch.cat([f8, fc], -1)
        cols.append(feats)
    cols.append(embed(view_dir[:, None, :].expand(R, S, 3), net.n_freq_dir))
    return torch.cat(cols, -1), p, feats


def run_mlp(net, x):
    R, S, C = x.shape
    return mlp_forward(net.state, x.reshape(R * S, C), net.spec).reshape(R, S, -1)


def rendering(rays_pts, rays_ndc, z, rays_dir, net_static, net_dynamic=None,
              vol_static=None, vol_dynamic=None, imgs=None, nb_imgs=None,
              cams=None, nb_cams=None, scene_flow=False, val=False, chain_bwd=False,
              chain_5frames=False, ref_frame_idx=None, num_frames=None,
              white_bkgd=False, raw_noise_std=0.0, noise=None, explicit=True, time_codes=None):
    """Flat-tensor restatement of rendering(), /root/reference/renderer.py:579-626.

    rays_pts, rays_ndc [R,S,3]; z [R,S]; rays_dir [R,3] (un-normalised).
    cams / nb_cams = (w2cs [V+1,4,4], intrinsics [V+1,3,3]); view 0 is the reference
    camera whose rotation defines the view-direction feature (renderer.py:256-258).
    ``noise`` optionally injects the density noise the reference draws with randn:
    dict with keys 'static' and 'blend', each [R,S]; it is only used where the
    reference would draw (raw_noise_std > 0).
    Returns the same keys as the reference, without the leading N=1 dimension.
    """
    dn = torch.linalg.vector_norm(rays_dir, dim=-1, keepdim=True)
    dists = sample_dists(z, dn)
    unit = rays_dir / dn

    def vdir(c):
        return unit @ c[0][0, :3, :3].t() if c is not None else unit

    use_noise = raw_noise_std > 0
    x, _, feats = build_mlp_input(net_static, rays_pts, rays_ndc, vdir(cams), vol_static, imgs,
                                  cams, None, explicit, time_codes)          # static net only (renderer.py:351)
    raw_s = run_mlp(net_static, x)
    raw_rgba = raw_s[..., :4]
    blend = raw_s[..., 4] if scene_flow else None
    rgb_map, _, _, w, depth, alpha = composite(
        raw_rgba, z, dists, white_bkgd, noise["static"] * raw_noise_std if use_noise else None)
    ret = dict(rgb_map=rgb_map, depth_map=depth, raw_rgba=raw_rgba, input_feat=feats,
               weights=w, raw_blend_w=blend, alpha=alpha)
    if not scene_flow:
        return ret

    def dyn_pass(ndc3, t):
        xi, p4, _ = build_mlp_input(net_dynamic, rays_pts, ndc3, vdir(nb_cams), vol_dynamic,
                                    nb_imgs, nb_cams, t, explicit)
        return run_mlp(net_dynamic, xi), p4

    raw_ref, p_ref = dyn_pass(rays_ndc, ref_frame_idx)
    sf_prev, sf_post = raw_ref[..., 4:7], raw_ref[..., 7:10]
    prob_prev, prob_post = raw_ref[..., 10], raw_ref[..., 11]
    rgb_ref, depth_ref, rgb_fg, depth_fg, w_fg, w_dd = composite_blend(
        raw_ref[..., :4], raw_rgba, blend, z, dists,
        noise["blend"] * raw_noise_std if use_noise else None)
    ret.update(rgb_map_ref=rgb_ref, depth_map_ref=depth_ref, rgb_map_ref_dy=rgb_fg,
               depth_map_ref_dy=depth_fg, weights_map_dd=w_dd.sum(-1).detach())   # renderer.py:436
    if val:
        return ret
    ret.update(raw_sf_ref2prev=sf_prev, raw_sf_ref2post=sf_post, raw_pts_ref=p_ref[..., :3],
               weights_ref_dy=w_fg, raw_blend_w=blend, raw_prob_ref2prev=prob_prev,
               raw_prob_ref2post=prob_post)
    # Neighbour-frame renders.  Reference quirk (renderer.py:478-479,505-506,541-543,
    # 570-572): raw_noise_std lands in the white_bkgd slot, so these composite onto
    # white whenever raw_noise_std is non-zero and never receive noise.
    white_nb = bool(raw_noise_std)
    step = 1.0 / num_frames * 2.0
    raw_prev, p_prev = dyn_pass(rays_ndc + sf_prev, ref_frame_idx - step)
    rgb_prev, _, _, w_prev, _, _ = composite(raw_prev[..., :4], z, dists, white_nb)
    raw_post, p_post = dyn_pass(rays_ndc + sf_post, ref_frame_idx + step)
    rgb_post, _, _, w_post, _, _ = composite(raw_post[..., :4], z, dists, white_nb)
    ret.update(raw_pts_prev=p_prev[..., :3], raw_sf_prev2ref=raw_prev[..., 7:10],
               rgb_map_prev_dy=rgb_prev, raw_pts_post=p_post[..., :3],
               raw_sf_post2ref=raw_post[..., 4:7], rgb_map_post_dy=rgb_post,
               # compute_2d_prob detaches the weights (renderer.py:31)
               prob_map_prev=(w_prev.detach() * (1.0 - prob_prev)).sum(-1),
               prob_map_post=(w_post.detach() * (1.0 - prob_post)).sum(-1))
    step2 = 2.0 / num_frames * 2.0
    if chain_bwd:
        ndc_pp, t_pp = p_prev[..., :3] + raw_prev[..., 4:7], ref_frame_idx - step2
    else:
        ndc_pp, t_pp = p_post[..., :3] + raw_post[..., 7:10], ref_frame_idx + step2
    ret["raw_pts_pp"] = ndc_pp
    if chain_5frames:
        raw_pp, _ = dyn_pass(ndc_pp, t_pp)
        ret["rgb_map_pp_dy"] = composite(raw_pp[..., :4], z, dists, white_nb)[0]
    return ret


# ------------------------------------------------------------ ray sampling (8(f) next-1)
def ndc_coordinate(pts, w2c, K, inv_w, inv_h, near, far, pad=0):
    """pts [...,3] world -> (u, v, z) of a view, normalised and pad-rescaled.
    Restates get_ndc_coordinate, /root/reference/utils.py:257-285 (projection branch)."""
    p = pts.reshape(-1, 3)
    pc = p @ w2c[:3, :3].t() + w2c[:3, 3]
    q = pc @ K.t()
    u = (q[:, 0] / q[:, 2] + 0.0) / inv_w
    v = (q[:, 1] / q[:, 2] + 0.0) / inv_h
    zn = (q[:, 2] - near) / (far - near)
    if pad > 0:
        wf, hf = (inv_w + 1) / 4.0, (inv_h + 1) / 4.0
        v = v * hf / (hf + pad * 2) + pad / (hf + pad * 2)
        u = u * wf / (wf + pad * 2) + pad / (wf + pad * 2)
    return torch.stack([u, v, zn], -1).reshape(pts.shape)


def sample_rays(xs, ys, K_tgt, c2w_tgt, w2c_ref, K_ref, near_tgt, far_tgt, near_ref, far_ref, S,
                t_rand=None, pad=0, W=None, H=None):
    """Pixels (xs, ys) [R] of the target camera -> rays_dir [R,3], depth [R,S], pts [R,S,3],
    ndc [R,S,3].  Restates the per-sample part of build_rays_base and the direction part of
    get_rays_mvs, /root/reference/utils.py:215-223, 361-387."""
    dirs = torch.stack([(xs - K_tgt[0, 2]) / K_tgt[0, 0], (ys - K_tgt[1, 2]) / K_tgt[1, 1],
                        torch.ones_like(xs)], -1)
    d = dirs @ c2w_tgt[:3, :3].t()
    o = c2w_tgt[:3, 3]
    t = torch.linspace(0., 1., steps=S, dtype=xs.dtype)
    z = (near_tgt * (1. - t) + far_tgt * t)[None].expand(xs.shape[0], S)
    if t_rand is not None:
        mids = .5 * (z[:, 1:] + z[:, :-1])
        upper = torch.cat([mids, z[:, -1:]], -1)
        lower = torch.cat([z[:, :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    pts = o[None, None, :] + z[..., None] * d[:, None, :]
    ndc = ndc_coordinate(pts, w2c_ref, K_ref, W - 1, H - 1, near_ref, far_ref, pad)
    return d, z, pts, ndc


# ----------------------------------------------------- plane sweep (8(f) row 3)
# Pinning: the reference's homo_warp builds its pixel grid with kornia.create_meshgrid, which
# is not installed here, so only its sampling half (a given src_grid -> F.grid_sample, zero
# padding, align_corners; utils.py:91-98) is pinned by a reference-generated fixture
# (tests/golden/homo_warp.npz).  plane_grid and volume_cost below restate utils.py:57-89 and
# networks.py:1077-1140 from the source text: PARITY UNPINNED for those two.
def plane_grid(proj, depth, H, W, pad=0):
    """proj [3,4] = src_proj @ ref_proj_inv, depth [D] -> normalised source positions
    [D, H+2pad, W+2pad, 2] of every reference pixel (x - pad, y - pad) on every depth plane.

    Restates homo_warp, /root/reference/utils.py:57-89 (create_meshgrid(normalized=False) is
    the integer pixel grid, x fastest)."""
    Hp, Wp = H + 2 * pad, W + 2 * pad
    ys, xs = torch.meshgrid(torch.arange(Hp, dtype=proj.dtype), torch.arange(Wp, dtype=proj.dtype), indexing="ij")
    ref = torch.stack([xs.reshape(-1) - pad, ys.reshape(-1) - pad, torch.ones(Hp * Wp, dtype=proj.dtype)])  # [3, HW]
    R, T = proj[:, :3], proj[:, 3:]
    ref_d = ref.repeat(1, depth.shape[0])                                         # [3, D*HW]
    dv = depth[:, None].expand(-1, Hp * Wp).reshape(1, -1)
    src = R @ ref_d + T / dv
    g = src[:2] / src[2:]
    gx = g[0] / ((W - 1) / 2) - 1
    gy = g[1] / ((H - 1) / 2) - 1
    return torch.stack([gx, gy], -1).view(depth.shape[0], Hp, Wp, 2)


def grid_warp(src, grid):
    """src [C,H,W], grid [D,Hp,Wp,2] -> [C,D,Hp,Wp]: bilinear, zero padding, align_corners.

    Restates the sampling half of homo_warp, /root/reference/utils.py:91-98."""
    D, Hp, Wp = grid.shape[:3]
    out = F.grid_sample(src[None], grid.reshape(1, D, Hp * Wp, 2), mode="bilinear", padding_mode="zeros",
                        align_corners=True)
    return out.view(src.shape[0], D, Hp, Wp)


def volume_cost(imgs, feats, proj_mats, depth, pad=0):
    """imgs [V,3,Hi,Wi], feats [V,C,H,W], proj_mats [V,3,4], depth [D] ->
    (img_feat [3V+C, D, Hp, Wp], in_masks [V, D, Hp, Wp]).

    Restates MVSNet.build_volume_cost, /root/reference/networks.py:1077-1140 (inference branch;
    the training branch is the same arithmetic out of place).  Channels 0-2 of the padding
    ring, which the reference leaves uninitialised, are 0."""
    V, C, H, W = feats.shape
    D, Hp, Wp = depth.shape[0], H + 2 * pad, W + 2 * pad
    ref = F.pad(feats[0], (pad, pad, pad, pad)) if pad > 0 else feats[0]
    img_feat = torch.zeros(3 * V + C, D, Hp, Wp, dtype=feats.dtype)
    imgs_lr = F.interpolate(imgs, (H, W), mode="bilinear", align_corners=False)
    img_feat[:3, :, pad:H + pad, pad:W + pad] = imgs_lr[0][:, None].expand(-1, D, -1, -1)
    vol_sum = ref[:, None].repeat(1, D, 1, 1)
    vol_sq = vol_sum ** 2
    masks = torch.ones(V, D, Hp, Wp, dtype=feats.dtype)
    for i in range(1, V):
        grid = plane_grid(proj_mats[i], depth, H, W, pad)
        warped = grid_warp(feats[i], grid)
        img_feat[3 * i:3 * i + 3] = grid_warp(imgs_lr[i], grid)
        inside = (grid > -1.0) & (grid < 1.0)
        masks[i] = (inside[..., 0] & inside[..., 1]).to(feats.dtype)
        vol_sum = vol_sum + warped
        vol_sq = vol_sq + warped ** 2
    count = 1.0 / masks.sum(0, keepdim=True)
    img_feat[-C:] = vol_sq * count - (vol_sum * count) ** 2
    return img_feat, masks


def _abn(x, st, prefix, training, eps=1e-5, slope=0.01):
    """Batch norm (batch statistics in training mode, the running estimates otherwise) + leaky ReLU(0.01): the stated
    reading of inplace_abn.InPlaceABN (reference networks.py:938-960 passes it as `norm_act`; not installable here)."""
    y = F.batch_norm(x, None if training else st[prefix + ".running_mean"], None if training else st[prefix + ".running_var"],
                     st[prefix + ".weight"], st[prefix + ".bias"], True if training else False, 0.0, eps)
    return F.leaky_relu(y, slope)


def cost_reg_net(st, x, training=False):
    """3-D regularisation net of the volume builder (reference networks.py:1003-1059): x [1,C,D,H,W] -> [1,8,D,H,W].
    st: state dict with the reference's keys."""
    def cbr(x, name, stride=1):
        return _abn(F.conv3d(x, st[name + ".conv.weight"], stride=stride, padding=1), st, name + ".bn", training)

    def up(x, name):
        return _abn(F.conv_transpose3d(x, st[name + ".0.weight"], stride=2, padding=1, output_padding=1), st, name + ".1", training)
    c0 = cbr(x, "conv0")
    c2 = cbr(cbr(c0, "conv1", 2), "conv2")
    c4 = cbr(cbr(c2, "conv3", 2), "conv4")
    x = cbr(cbr(c4, "conv5", 2), "conv6")
    x = c4 + up(x, "conv7")
    x = c2 + up(x, "conv9")
    return c0 + up(x, "conv11")


def feature_net(st, x, training=False):
    """2-D feature pyramid of the volume builder (reference networks.py:962-1001): x [N,3,H,W] -> [N,32,H/4,W/4]."""
    for name, k, stride in (("conv0.0", 3, 1), ("conv0.1", 3, 1), ("conv1.0", 5, 2), ("conv1.1", 3, 1), ("conv1.2", 3, 1),
                            ("conv2.0", 5, 2), ("conv2.1", 3, 1), ("conv2.2", 3, 1)):
        x = _abn(F.conv2d(x, st[name + ".conv.weight"], stride=stride, padding=k // 2), st, name + ".bn", training)
    return F.conv2d(x, st["toplayer.weight"], st["toplayer.bias"])


def graf_patch_pixels(H, W, patch_size, step, scale_anneal=-1, min_scale=0.25, max_scale=1.0):
    """Pixel (x, y) indices of GRAF's variable patch.  Restates patch_ray_sampler and the variable_patches branch
    of get_rays_mvs, /root/reference/utils.py:102-131, 157-170: a patch_size^2 lattice over [-1, 1]^2 scaled by
    s ~ U(min_scale', max_scale) (min_scale' annealed with the step) and shifted by +-U(0, 1 - s) per axis - draws in
    that order from torch's CPU generator - then mapped to pixels by bilinear sampling of the coordinate ramps
    and truncation to int."""
    import math
    lin = torch.linspace(-1, 1, patch_size)
    if scale_anneal > 0:
        min_scale = min(0.9, max(min_scale, max_scale * math.exp(-(step // 1000 * 3) * scale_anneal)))
    s = torch.Tensor(1).uniform_(min_scale, max_scale)
    gx = (lin[None, :] * s).expand(patch_size, patch_size)
    gy = (lin[:, None] * s).expand(patch_size, patch_size)
    room = 1 - s.item()
    gx = gx + torch.Tensor(1).uniform_(0, room) * (torch.randint(2, (1,)).float() - 0.5) * 2
    gy = gy + torch.Tensor(1).uniform_(0, room) * (torch.randint(2, (1,)).float() - 0.5) * 2
    grid = torch.stack([gx, gy], -1)[None]
    ramp_x = torch.arange(W, dtype=torch.float32)[None, :].expand(H, W)[None, None]
    ramp_y = torch.arange(H, dtype=torch.float32)[:, None].expand(H, W)[None, None]
    xs = F.grid_sample(ramp_x, grid, mode="bilinear", align_corners=True).reshape(-1).int()
    ys = F.grid_sample(ramp_y, grid, mode="bilinear", align_corners=True).reshape(-1).int()
    return xs, ys


# ------------------------------------------------ loss-side reductions (8(f) row 4)
def distortion_loss(ray_weights, t_vals):
    """ray_weights [R,S], t_vals [1,S] or [R,S] -> scalar.

    Restates distortion_loss, /root/reference/losses.py:53-87 (pairs over the S-1 intervals)."""
    w = ray_weights[..., :-1]
    mids = 0.5 * (t_vals[..., :-1] + t_vals[..., 1:])
    pair = (w[..., :, None] * w[..., None, :]) * (mids[..., :, None] - mids[..., None, :]).abs()
    inter = (1.0 / 3.0) * (w * w * (t_vals[..., 1:] - t_vals[..., :-1])).sum(-1)
    return (0.5 * pair.sum((-1, -2)) + inter).sum()


def projection_from_ndc(w2c, H, W, f, weights, pts):
    """weights [R,S], pts [R,S,3], w2c [4,4] -> [R,2].

    Restates projection_from_ndc with NDC2Euclidean, se3_transform_points and
    perspective_projection, /root/reference/utils.py:507-539."""
    p = (weights[..., None] * pts).sum(-2)
    ze = 2.0 / (p[..., 2:3].clamp(-1.0, 0.99) - 1.0)
    xe = -p[..., 0:1] * ze * W / (2.0 * f)
    ye = -p[..., 1:2] * ze * H / (2.0 * f)
    e = torch.cat([xe, ye, ze], -1)
    loc = e @ w2c[:3, :3].T + w2c[:3, 3]
    return torch.cat([loc[..., 0:1] * f / -loc[..., 2:3] + W / 2.0,
                      -loc[..., 1:2] * f / -loc[..., 2:3] + H / 2.0], -1)


def num_threads():
    return torch.get_num_threads()


def mlp_flops_per_sample(spec):
    """2 x sum(in*out) over the linear layers one forward executes."""
    P, Fd, Vw, W = spec.in_ch_pts, spec.in_ch_feat, spec.in_ch_views, spec.W
    macs = P * W + (spec.D - 2) * W * W + (W + P) * W
    if spec._has_feat():
        macs += Fd * W
    macs += W * W + W + (W + Vw) * (W // 2) + (W // 2) * 3
    if spec.net_type == "v0" and spec.sceneflow:
        macs += W if spec.static else 8 * W
    return 2 * macs
