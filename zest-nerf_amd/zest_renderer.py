"""Drop-in `renderer` module: ZeST-NeRF volume rendering on MI355X (HIP kernels).

`rendering` keeps the reference's signature and return-dict keys
(/root/reference/renderer.py:579-626); so do the helpers other reference modules import
(`raw2outputs`, `raw2outputs_blending`, `raw2alpha`, `depth2dist`, `compute_2d_prob`).
All device work goes through the C ABI (include/zest_render.h); there is no PyTorch
restatement in this package.  When autograd is recording and a parameter, a volume or the
sample coordinates want a gradient, every stage runs as an autograd.Function whose forward
and backward are HIP kernels (zest_autograd.py, fp32); otherwise the inference kernels run.

Two execution plans:
  * complete  - per-sample tensors in HBM (encode -> MLP -> composite kernels); returns
                every key of the reference dict.  Used for train-mode calls.
  * fused     - one launch, per-ray maps only (zest_render_fused_fwd); chosen for
                inference-shaped calls when `args.zest_maps_only` is set (the whole-image
                loops in the generators set it: they read 2 / 7 per-ray maps only,
                reference networks.py:697-704, train.py:898-899).  Operand type by
                networks.resolve_precision: bf16 / fp16, or split-fp16 pairs in fp32 mode.
"""
import torch

import zest_hip
from zest_networks import MVSNeRF, inference_precision, resolve_precision
from zest_utils import images_channels_last, volume_channels_last

__all__ = ["rendering", "render_hierarchical", "raw2outputs", "raw2outputs_blending", "raw2alpha", "depth2dist",
           "compute_2d_prob"]


def _draw_noise(shape, device):
    """Density noise ~ N(0,1); one draw per compositing call that uses it, like the reference
    (renderer.py:140,189).  Tests replace this hook to inject known noise."""
    return torch.randn(shape, device=device)


# ------------------------------------------------------------------ reference helper surface
def compute_2d_prob(weights_p_mix, raw_prob_ref2p):
    """sum_s w * (1 - prob)  (reference renderer.py:22-32)."""
    lead = weights_p_mix.shape[:-1]
    S = weights_p_mix.shape[-1]
    out = zest_hip.weighted_complement_sum(weights_p_mix.reshape(-1, S), raw_prob_ref2p.reshape(-1, S))
    return out.view(*lead)


def depth2dist(z_vals, cos_angle):
    """Sample spacing along the ray, last = 1e10, scaled by |dir| (reference renderer.py:74-89).
    For callers of the module-level raw2outputs*; `rendering` itself lets the compositing kernels
    rebuild the spacing from z and rays_dir."""
    d = z_vals[..., 1:] - z_vals[..., :-1]
    d = torch.cat([d, torch.full_like(z_vals[..., :1], 1e10)], -1)
    return d * cos_angle


def raw2alpha(sigma, dist):
    """alpha = 1 - exp(-sigma*dist), weights = alpha * exclusive-prod(1 - alpha + 1e-10)
    (reference renderer.py:91-113), on the compositing kernel with the caller's spacings; sigma < 0
    is clamped to 0, which every reference caller has already done.  Under autograd the weights come from
    CompositeFn (HIP forward and backward) and alpha from the two elementwise ops that define it; the spacings
    are data, as in every call site of the reference."""
    lead, S = sigma.shape[:-1], sigma.shape[-1]
    d = dist.reshape(-1, S).float()
    sg = sigma.reshape(-1, S).float()
    if torch.is_grad_enabled() and dist.requires_grad:
        raise NotImplementedError("zest raw2alpha: a gradient with respect to the sample spacings is not provided "
                                  "(no caller of the reference differentiates them: renderer.py:74-89 builds them "
                                  "from the depth samples)")
    if torch.is_grad_enabled() and sigma.requires_grad:
        import zest_autograd as za
        raw = torch.cat([torch.zeros(sg.shape[0], S, 3, device=sg.device), sg[..., None]], -1)
        w = za.CompositeFn.apply(raw, torch.zeros_like(d), d.contiguous(), None, 0.0, False, True)[3]
        a = 1.0 - torch.exp(-torch.relu(sg) * d)
        return a.view(*lead, S), w.view(*lead, S)
    raw = torch.zeros(d.shape[0], S, 4, device=d.device)
    raw[..., 3] = sg
    _, _, _, w, _, a = zest_hip.composite(raw, torch.zeros_like(d), None, dists=d)
    return a.view(*lead, S), w.view(*lead, S)


def raw2outputs(raw, z_vals, dists, white_bkgd=False, raw_noise_std=0):
    """[N,R,S,4] -> rgb_map, disp_map, acc_map, weights, depth_map, alpha
    (reference renderer.py:115-164).  `dists` is used as given, whatever produced it."""
    lead, S = z_vals.shape[:-1], z_vals.shape[-1]
    z, d = z_vals.reshape(-1, S), dists.reshape(-1, S)
    noise = _draw_noise(z_vals.shape, z_vals.device).reshape(-1, S) if raw_noise_std > 0 else None
    if torch.is_grad_enabled() and raw.requires_grad:
        import zest_autograd as za
        rgb, disp, acc, w, depth, a = za.CompositeFn.apply(raw.reshape(-1, S, 4).contiguous(), z, d.contiguous(), noise,
                                                           float(raw_noise_std), bool(white_bkgd), True)
    else:
        rgb, disp, acc, w, depth, a = zest_hip.composite(raw.reshape(-1, S, 4), z, None, noise,
                                                         float(raw_noise_std), bool(white_bkgd), dists=d)
    return (rgb.view(*lead, 3), disp.view(*lead), acc.view(*lead), w.view(*lead, S),
            depth.view(*lead), a.view(*lead, S))


def raw2outputs_blending(raw_dy, raw_rigid, raw_blend_w, z_vals, dists, raw_noise_std=0):
    """-> rgb_map, depth_map, rgb_map_fg, depth_map_fg, weights_fg, weights_dy
    (reference renderer.py:166-219).  `dists` is used as given."""
    lead, S = z_vals.shape[:-1], z_vals.shape[-1]
    z, d = z_vals.reshape(-1, S), dists.reshape(-1, S)
    noise = _draw_noise(z_vals.shape, z_vals.device).reshape(-1, S) if raw_noise_std > 0 else None
    args3 = (raw_dy.reshape(-1, S, 4), raw_rigid.reshape(-1, S, 4), raw_blend_w.reshape(-1, S))
    if torch.is_grad_enabled() and any(t.requires_grad for t in args3):
        import zest_autograd as za
        rgb, depth, rgb_fg, depth_fg, w_fg, w_dy, _ = za.BlendFn.apply(
            *[t.contiguous() for t in args3], z, d.contiguous(), noise, float(raw_noise_std), True)
    else:
        rgb, depth, rgb_fg, depth_fg, w_fg, w_dy, _ = zest_hip.composite_blend(*args3, z, None, noise,
                                                                               float(raw_noise_std), dists=d)
    return (rgb.view(*lead, 3), depth.view(*lead), rgb_fg.view(*lead, 3), depth_fg.view(*lead),
            w_fg.view(*lead, S), w_dy.view(*lead, S))


# ------------------------------------------------------------------------------ view sets
class _Views:
    """Volume + source images + cameras of one feature lookup, in kernel layout."""

    def __init__(self, volume, imgs, cam_mat):
        self.vol_cl = self.imgs_cl = self.w2cs = self.intr = None
        if cam_mat is not None:
            self.w2cs = cam_mat['w2cs'][0].float().contiguous()
            self.intr = cam_mat['intrinsics'][0].float().contiguous()
        if volume is not None:
            if not torch.is_tensor(volume):
                raise NotImplementedError("zest renderer: callable volumes are not supported")
            if imgs is None or cam_mat is None:
                raise RuntimeError("zest renderer: a feature volume needs source images and cameras")
            self.vol_cl = volume_channels_last(volume)
            self.imgs_cl = images_channels_last(imgs)

    def encode(self, ndc, pts, rays_dir, t=None, out=None):
        return zest_hip.encode(ndc, pts, rays_dir, t, self.vol_cl, self.imgs_cl, self.w2cs, self.intr, out=out)

    def c_in(self, has_t):
        """input width of the MLP these views feed: PE(xyz[t]) + 8 + 4V features + PE(direction)"""
        return (4 if has_t else 3) * 21 + (8 + 4 * self.imgs_cl.shape[0] if self.vol_cl is not None else 0) + 27


def _check_embedders(embedding_pts, embedding_dir, in_channels):
    for e, c, L, what in ((embedding_pts, in_channels, 10, "points"), (embedding_dir, 3, 4, "directions")):
        if e is None or getattr(e, "N_freqs", None) != L or getattr(e, "in_channels", None) != c:
            raise NotImplementedError(
                "zest renderer: the HIP encoder implements the shipped embedders only "
                "(%s: in_channels=%d, N_freqs=%d)" % (what, c, L))


def _net(network_fn, what):
    if not isinstance(network_fn, MVSNeRF):
        raise TypeError("zest renderer: %s must be a zest networks.MVSNeRF (got %s)" %
                        (what, type(network_fn).__name__))
    return network_fn


def _render_maps_fused(prec, time_codes, rays_ndc, ndc, pts, z, dirs, net_s, network_fn_dy, scene_flow, vol_s, vol_d,
                       imgs, nb_imgs, cam, nb_cam, embedding_xyzt, embedding_dir, ref_frame_idx,
                       white_bkgd, raw_noise_std):
    """Per-ray maps only, one kernel launch (zest_render_fused_fwd).  Returns the per-ray keys
    of the reference dict (plus 'acc_map'); per-sample keys are not produced."""
    if raw_noise_std > 0:
        raise NotImplementedError("zest fused renderer: density noise is a training-time option")
    vs = _Views(vol_s, imgs, cam)
    views_s = zest_hip.make_view_set(vs.vol_cl, vs.imgs_cl, vs.w2cs, vs.intr)
    desc_d = packed_d = views_d = None
    if scene_flow:
        net_d = _net(network_fn_dy, "network_fn_dy")
        _check_embedders(embedding_xyzt, embedding_dir, 4)
        vd = _Views(vol_d, nb_imgs, nb_cam)
        desc_d, packed_d = net_d.desc(), net_d.packed(prec)
        views_d = zest_hip.make_view_set(vd.vol_cl, vd.imgs_cl, vd.w2cs, vd.intr)
    out = zest_hip.render_fused(ndc, pts, z, dirs, net_s.desc(), net_s.packed(prec, time_codes),
                                views_s, desc_d, packed_d, views_d,
                                ref_frame_idx if scene_flow else 0.0, white_bkgd, precision=prec)
    ret = {'rgb_map': out[None, :, 0:3], 'depth_map': out[None, :, 3], 'acc_map': out[None, :, 4],
           'zest_packed_maps': out}
    if scene_flow:
        ret.update({'rgb_map_ref': out[None, :, 5:8], 'depth_map_ref': out[None, :, 8],
                    'rgb_map_ref_dy': out[None, :, 9:12], 'depth_map_ref_dy': out[None, :, 12],
                    'weights_map_dd': out[None, :, 13]})
    return ret


# ------------------------------------------------------------------------------- rendering
def rendering(args, rays_pts, rays_ndc, depth_candidates, rays_dir,
              volume_feature_static=None, volume_feature_dynamic=None,
              imgs=None, img_feat=None, neighbour_frames=None,
              im_cam_mat=None, nb_cam_mat=None,
              network_fn=None, network_fn_dy=None,
              embedding_pts=None, embedding_xyzt=None, embedding_dir=None,
              chain_bwd=False, chain_5frames=False, ref_frame_idx=None, num_frames=None,
              time_codes=None, white_bkgd=False, scene_flow=False, val=False,
              raw_noise_std=0):
    """Volume-render a batch of rays; same arguments and result keys as the reference
    (renderer.py:579-626).  rays_pts/rays_ndc [1,R,S,3], depth_candidates [1,R,S],
    rays_dir [1,R,3]."""
    if img_feat is not None:
        raise NotImplementedError("zest renderer: img_feat is always None in the reference callers")
    if getattr(args, "use_color_volume", False):
        raise NotImplementedError("zest renderer: use_color_volume is never set by the reference configs")
    if rays_ndc.shape[0] != 1:
        raise RuntimeError("zest renderer: image batch must be 1 (reference train.py:307)")
    net_s = _net(network_fn, "network_fn")
    _check_embedders(embedding_pts, embedding_dir, 3)
    prec = resolve_precision(args)
    N, R, S, _ = rays_ndc.shape
    ndc, pts = rays_ndc[0].float().contiguous(), rays_pts[0].float().contiguous()
    z, dirs = depth_candidates[0].float().contiguous(), rays_dir[0].float().contiguous()
    # Training path: autograd recording and something upstream wants a gradient.  Every stage
    # then runs as an autograd.Function with HIP forward and backward (zest_autograd.py), fp32.
    def wants_grad(*ts):
        return any(t is not None and torch.is_tensor(t) and t.requires_grad for t in ts)
    # Neural3D time code (reference renderer.py:269-273; static net only, :351): constant over the
    # batch, so the net folds it into the biases of its layers 0 and 5 (zest_networks.effective_parameters)
    train = torch.is_grad_enabled() and (
        wants_grad(rays_ndc, volume_feature_static, volume_feature_dynamic, time_codes)
        or any(p.requires_grad for p in net_s.parameters())
        or (network_fn_dy is not None and any(p.requires_grad for p in network_fn_dy.parameters())))
    if train:
        import zest_autograd as za

    # Fused single-launch plan: inference-shaped calls that read per-ray maps only.  fp32 mode runs
    # it on split-fp16 operand pairs (fp32-class results); a call that records a gradient takes the
    # complete plan below, whose stages have backward kernels.
    # The engine kernels are written for the shipped MLP shape (D=8, W=256, skips=[4]); a net of another
    # shape takes the complete plan with the exact-product fp32 MLP kernel.
    engine_shape = net_s.nerf.default_shape and (network_fn_dy is None or _net(network_fn_dy, "network_fn_dy").nerf.default_shape)
    # ('v2' static nets have single-net fused kernels only; next to a dynamic net they take the complete plan)
    fused_ok = engine_shape and not (scene_flow and net_s.desc().net_type == 2)
    if getattr(args, "zest_maps_only", False) and (val or not scene_flow) and not train and fused_ok:
        fused_prec = zest_hip.PREC_F16X3 if prec == zest_hip.PREC_F32 else prec     # no exact-product fused kernel
        return _render_maps_fused(fused_prec, time_codes, rays_ndc, ndc, pts, z, dirs, net_s, network_fn_dy, scene_flow,
                                  volume_feature_static, volume_feature_dynamic, imgs,
                                  neighbour_frames, im_cam_mat, nb_cam_mat, embedding_xyzt,
                                  embedding_dir, ref_frame_idx, white_bkgd, raw_noise_std)

    mlp_prec = inference_precision(prec, args)

    train16 = prec in (zest_hip.PREC_BF16, zest_hip.PREC_F16)       # --precision 16: bf16 MFMA training kernels
    if train and prec == zest_hip.PREC_F16:
        import warnings
        warnings.warn("zest: zest_dtype16='f16' selects fp16 operands for inference only; under autograd --precision 16 "
                      "trains on the bf16 MFMA kernels (a gradient wants bf16's exponent range), so this run trains in "
                      "bf16 and evaluates in fp16", stacklevel=2)

    shared = {}                 # per net: what its passes of this call share (packed weight streams, work buffer)

    def mlp(net, x, tc=None):
        if train:
            return za.mlp_apply(net, x, tc, bf16=train16, shared=shared.setdefault(id(net), {}))
        p = net.nerf.engine_precision(mlp_prec)
        return zest_hip.mlp_fwd(net.desc(), p, net.packed(p, tc), x)

    def vol_node(views, volume):
        # the volume's channels-last copy as an autograd node (one per volume and call) when the volume trains
        if getattr(views, "vol_g", None) is None and volume is not None and torch.is_tensor(volume) and volume.requires_grad:
            views.vol_g = za.VolumeCLFn.apply(volume, views)
        return getattr(views, "vol_g", None)

    def encode(views, volume, ndc3, t=None):
        if train:
            return za.EncodeFn.apply(ndc3, vol_node(views, volume), views, pts, dirs, None if t is None else float(t))
        return views.encode(ndc3, pts, dirs, t)

    def composite(raw4, nz, std, white):
        if train:
            return za.CompositeFn.apply(raw4.contiguous(), z, dirs, nz, float(std), bool(white))
        return zest_hip.composite(raw4, z, dirs, nz, float(std), bool(white))

    def noise():
        return _draw_noise((1, R, S), ndc.device)[0] if raw_noise_std > 0 else None

    # ---- static NeRF (reference render_static, renderer.py:322-373)
    vs = _Views(volume_feature_static, imgs, im_cam_mat)
    if train and rays_ndc.requires_grad:
        ndc = rays_ndc[0].float()
    x_s = encode(vs, volume_feature_static, ndc)
    raw_s = mlp(net_s, x_s, time_codes)
    if scene_flow and train:
        raw_rgba, blend = za.split_last(raw_s, (4, 1))
        blend = blend[..., 0]
    else:
        raw_rgba = raw_s[..., :4]
        blend = raw_s[..., 4] if scene_flow else None
    rgb_map, _, _, weights, depth_map, alpha = composite(raw_rgba, noise(), raw_noise_std, white_bkgd)
    F = net_s.in_ch_feat if vs.vol_cl is not None else 0
    input_feat = x_s[None, ..., 63:63 + F] if F else None
    ret = {'rgb_map': rgb_map[None], 'depth_map': depth_map[None], 'raw_rgba': raw_rgba[None],
           'input_feat': input_feat, 'weights': weights[None],
           'raw_blend_w': blend[None] if blend is not None else None, 'alpha': alpha[None]}
    if not scene_flow:
        return ret

    # ---- dynamic NeRF (reference render_dynamic, renderer.py:378-575)
    net_d = _net(network_fn_dy, "network_fn_dy")
    _check_embedders(embedding_xyzt, embedding_dir, 4)
    vd = _Views(volume_feature_dynamic, neighbour_frames, nb_cam_mat)

    def dyn_pass(ndc3, t):
        return mlp(net_d, encode(vd, volume_feature_dynamic, ndc3, float(t)))

    raw_ref = dyn_pass(ndc, ref_frame_idx)
    if train:
        # column groups through ONE autograd node per network output (their gradients come back as one concatenation)
        ref_rgba, sf_prev, sf_post, prob_prev, prob_post = za.split_last(raw_ref, (4, 3, 3, 1, 1))
        prob_prev, prob_post = prob_prev[..., 0], prob_post[..., 0]
        rgb_ref, depth_ref, rgb_fg, depth_fg, w_fg, w_dd, dd_sum = za.BlendFn.apply(
            ref_rgba, raw_rgba.contiguous(), blend.contiguous(), z, dirs, noise(),
            float(raw_noise_std))
    else:
        sf_prev, sf_post = raw_ref[..., 4:7], raw_ref[..., 7:10]
        prob_prev, prob_post = raw_ref[..., 10], raw_ref[..., 11]
        rgb_ref, depth_ref, rgb_fg, depth_fg, w_fg, w_dd, dd_sum = zest_hip.composite_blend(
            raw_ref[..., :4], raw_rgba, blend, z, dirs, noise(), float(raw_noise_std))
    ret.update({'rgb_map_ref': rgb_ref[None], 'depth_map_ref': depth_ref[None],
                'rgb_map_ref_dy': rgb_fg[None], 'depth_map_ref_dy': depth_fg[None],
                'weights_map_dd': dd_sum[None]})
    if val:
        if getattr(args, "zest_resample_weights", False):        # render_hierarchical's coarse pass
            ret['zest_weights_fg'] = w_fg[None]
        return ret
    ret.update({'raw_sf_ref2prev': sf_prev[None], 'raw_sf_ref2post': sf_post[None],
                'raw_pts_ref': ndc[None] if train else rays_ndc[..., :3], 'weights_ref_dy': w_fg[None],
                'raw_blend_w': blend[None], 'raw_prob_ref2prev': prob_prev[None],
                'raw_prob_ref2post': prob_post[None]})
    # Neighbour frames.  Reference quirk kept on purpose (renderer.py:478-479, 505-506,
    # 541-543, 570-572): raw_noise_std is passed in the white_bkgd position, so these renders
    # composite onto white whenever raw_noise_std != 0 and never receive density noise.
    white_nb = bool(raw_noise_std)
    step = 1. / num_frames * 2.

    def nb_render(raw4):
        rgb, _, _, w, _, _ = composite(raw4, None, 0.0, white_nb)
        return rgb, w

    def parts(raw):             # (rgba, columns 4:7, columns 7:10) of a dynamic-net output
        if train:
            a, b, c, _ = za.split_last(raw, (4, 3, 3, 2))
            return a, b, c
        return raw[..., :4], raw[..., 4:7], raw[..., 7:10]

    def prob2d(w, p):
        if train:
            return za.Prob2dFn.apply(w.detach(), p.contiguous())
        return zest_hip.weighted_complement_sum(w, p)

    ndc_prev, ndc_post = ndc + sf_prev, ndc + sf_post
    if train and train16:
        # the two neighbour frames as ONE batch of 2R rays through the dynamic net (the reference runs them one after
        # the other, renderer.py:460-505; rays are independent, so the rows are the same): one forward, one backward,
        # one set of parameter gradients for autograd to add instead of two.  bf16 kernels only: the fp32 path's
        # rocBLAS GEMMs run 20 % slower per sample on the doubled batch (49 against 41 ms per step)
        x_nb = za.EncodePairFn.apply(ndc_prev, ndc_post, vol_node(vd, volume_feature_dynamic), vd, pts, dirs,
                                     float(ref_frame_idx - step), float(ref_frame_idx + step))
        raw_prev, raw_post = za.SplitRowsFn.apply(mlp(net_d, x_nb), 2)
    else:
        raw_prev, raw_post = dyn_pass(ndc_prev, ref_frame_idx - step), dyn_pass(ndc_post, ref_frame_idx + step)
    prev4, prev_4_7, prev_7_10 = parts(raw_prev)
    rgb_prev, w_prev = nb_render(prev4)
    post4, post_4_7, post_7_10 = parts(raw_post)
    rgb_post, w_post = nb_render(post4)
    ret.update({'raw_pts_prev': ndc_prev[None], 'raw_sf_prev2ref': prev_7_10[None],
                'rgb_map_prev_dy': rgb_prev[None], 'raw_pts_post': ndc_post[None],
                'raw_sf_post2ref': post_4_7[None], 'rgb_map_post_dy': rgb_post[None],
                'prob_map_prev': prob2d(w_prev, prob_prev)[None],
                'prob_map_post': prob2d(w_post, prob_post)[None]})
    if chain_bwd:
        ndc_pp, t_pp = ndc_prev + prev_4_7, ref_frame_idx - 2. / num_frames * 2.
    else:
        ndc_pp, t_pp = ndc_post + post_7_10, ref_frame_idx + 2. / num_frames * 2.
    ret['raw_pts_pp'] = ndc_pp[None]
    if chain_5frames:
        ret['rgb_map_pp_dy'] = nb_render(dyn_pass(ndc_pp, t_pp)[..., :4])[0][None]
    return ret


# ------------------------------------------------------------------- coarse -> fine (build extension)
def merge_samples(rays_pts, depth_candidates, rays_dir, z_new):
    """The depth samples of a batch and `z_new` [1,R,N] more along the same rays -> (z_all [1,R,S+N] ascending,
    points [1,R,S+N,3]).  A ray is o + z d with the caller's un-normalised d (reference utils.py:378-379), so its
    origin is recovered from the first sample."""
    z, d = depth_candidates[0].float(), rays_dir[0].float()
    o = rays_pts[0, :, 0].float() - z[:, :1] * d
    z_all, _ = torch.sort(torch.cat([z, z_new[0].float()], -1), -1)
    return z_all[None], (o[:, None, :] + z_all[..., None] * d[:, None, :])[None]


def render_hierarchical(args, rays_pts, rays_ndc, depth_candidates, rays_dir, N_importance, ndc_of, det=True,
                        network_fn_fine=None, **kw):
    """Coarse -> fine rendering: BASELINE.json configs[3]'s "192 samples (coarse+fine)".

    BUILD EXTENSION, PARITY UNPINNED: the reference parses --N_importance (opt.py:161) and builds a fine net
    (train.py:143-148) but no renderer call ever receives either, and it has no sample_pdf ("Read this
    first" 1).  Canonical NeRF semantics: a coarse pass over the caller's samples yields per-sample compositing
    weights (static net's; with scene_flow and val=True plus the dynamic net's own weights), `utils.sample_pdf`
    draws N_importance more depths from them (inverse CDF over the mid-point bins, interior weights), the merged,
    sorted samples are rendered once more (`network_fn_fine` if given, else the same net) - that second render is
    an ordinary rendering() call and takes the fused single-launch kernel when args.zest_maps_only asks for it.
      ndc_of: points [1,R,S',3] (world) -> volume coordinates [1,R,S',3], the map the ray sampler applied to the
              caller's samples (e.g. zest_utils.get_ndc_coordinate on the reference view);
      **kw:   the keyword arguments of rendering().
    -> the fine pass's result dict plus 'rgb_map_coarse', 'depth_map_coarse' and 'z_vals' (the merged depths).
    The coarse pass runs without a graph (its weights are data for the sampler, as in NeRF)."""
    import copy
    from zest_utils import sample_pdf
    if N_importance < 1:
        raise ValueError("render_hierarchical: N_importance must be positive")
    coarse_args = copy.copy(args)
    coarse_args.zest_maps_only, coarse_args.zest_resample_weights = False, True     # per-sample weights wanted
    with torch.no_grad():
        coarse = rendering(coarse_args, rays_pts, rays_ndc, depth_candidates, rays_dir, **kw)
        w = coarse['weights'][0]
        if coarse.get('zest_weights_fg') is not None:
            w = w + coarse['zest_weights_fg'][0]
        z = depth_candidates[0].float()
        z_new = sample_pdf(0.5 * (z[:, 1:] + z[:, :-1]), w[:, 1:-1].contiguous(), N_importance, det=det)
        z_all, pts_all = merge_samples(rays_pts, depth_candidates, rays_dir, z_new[None])
        ndc_all = ndc_of(pts_all)
    kw_fine = dict(kw)
    if network_fn_fine is not None:
        kw_fine['network_fn'] = network_fn_fine
    fine = rendering(args, pts_all, ndc_all, z_all, rays_dir, **kw_fine)
    key = 'rgb_map_ref' if kw.get('scene_flow') else 'rgb_map'
    fine.update(rgb_map_coarse=coarse[key], depth_map_coarse=coarse[key.replace('rgb', 'depth')], z_vals=z_all)
    return fine
