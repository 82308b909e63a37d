"""ctypes binding of libzest_hip.so (include/zest_render.h) for torch tensors.

The library is the product: if it is missing or a call fails this module raises — there
is no PyTorch or CPU fallback anywhere in the package.  Tensors cross the boundary as raw
device pointers (`Tensor.data_ptr()`), sizes, and the current HIP stream handle; outputs
are allocated here through torch's caching allocator (the reference allocates every
output fresh as well, renderer.py:64, utils.py:480).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZEST_HIP_LIB selects an experiment build (build_hip.py --tag); default is the product library
LIB_PATH = os.environ.get("ZEST_HIP_LIB") or os.path.join(_HERE, "libzest_hip.so")

# arithmetic of the MLP contraction (include/zest_render.h ZEST_PREC_*): exact-product fp32 MFMA;
# bf16 / fp16 operands; split-fp16 pairs (fp32-class results on the fp16 matrix pipe)
PREC_F32, PREC_BF16, PREC_F16, PREC_F16X3 = 0, 1, 2, 3
ENGINE_PRECISIONS = (PREC_BF16, PREC_F16, PREC_F16X3)
PREC_NAMES = {PREC_F32: "f32", PREC_BF16: "bf16", PREC_F16: "f16", PREC_F16X3: "f16x3"}
HEAD_NONE, HEAD_BLEND, HEAD_DYNAMIC = 0, 1, 2
P_COUNT = 15

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_fp = C.POINTER(C.c_float)


class MlpDesc(C.Structure):
    _fields_ = [("in_ch_pts", C.c_int32), ("in_ch_feat", C.c_int32), ("in_ch_views", C.c_int32),
                ("use_feat", C.c_int32), ("net_type", C.c_int32), ("head", C.c_int32),
                ("depth", C.c_int32), ("width", C.c_int32), ("skip_mask", C.c_int32)]

    @property
    def D(self):
        return self.depth or 8

    @property
    def W(self):
        return self.width or 256

    @property
    def skips(self):
        """Reference `skips` list (networks.py:93-100): layer i+1 takes [pts | h] for i in skips."""
        mask = self.skip_mask if (self.depth or self.width or self.skip_mask) else 1 << 4
        return [i for i in range(8) if mask >> i & 1]

    @property
    def is_default_shape(self):
        return self.D == 8 and self.W == 256 and self.skips == [4]

    @property
    def in_ch(self):
        return self.in_ch_pts + (self.in_ch_feat if self.use_feat else 0) + self.in_ch_views

    @property
    def out_ch(self):
        return {HEAD_NONE: 4, HEAD_BLEND: 5, HEAD_DYNAMIC: 12}[self.head]


class ViewSet(C.Structure):
    _fields_ = [("vol_cl", _vp), ("D", C.c_int32), ("Hv", C.c_int32), ("Wv", C.c_int32),
                ("imgs_cl", _vp), ("V", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("w2cs", _vp), ("intrinsics", _vp)]


_SIGS = {
    "zest_abi_version": (_i, []),
    "zest_last_error": (C.c_char_p, []),
    "zest_device_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.c_char_p, _sz]),
    "zest_composite_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "zest_composite_blend_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "zest_weighted_complement_sum": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "zest_embed_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "zest_volume_to_cl": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "zest_images_to_cl": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "zest_nchw_to_nhwc": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "zest_distortion_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "zest_project_rays_fwd": (_i, [_vp, _vp, _vp, _i, _i, _f, _i, _i, _vp, _vp]),
    "zest_project_rays_bwd": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _i, _i, _vp, _vp, _vp]),
    "zest_volume_cost_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "zest_homo_warp_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "zest_volume_cost_cl_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "zest_costreg_stat_rows": (_i, []),
    "zest_conv2d_packed_bytes": (_sz, [_i, _i, _i, _i]),
    "zest_conv2d_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "zest_costreg_packed_bytes": (_sz, [_i, _i, _i]),
    "zest_costreg_conv_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "zest_costreg_deconv_packed_bytes": (_sz, [_i, _i, _i]),
    "zest_costreg_deconv_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "zest_costreg_bn": (_i, [_vp, _i, C.c_longlong, _vp, _vp, _f, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "zest_costreg_out": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "zest_costreg_bn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, C.c_longlong, _vp, _vp, _vp, _vp]),
    "zest_volume_cost_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "zest_homo_warp_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "zest_volume_lookup_fwd": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "zest_color_lookup_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "zest_encode_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp, _i, _i, _i, _vp, _i, _i, _i,
                             _vp, _vp, _vp, _vp]),
    "zest_gather_encode_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp, _i, _i, _i, _vp, _i, _i, _i,
                                    _vp, _vp, _vp, _vp]),            # = zest_encode_fwd (SURVEY 8(b)'s name)
    "zest_build_rays_fwd": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                                 _vp, _vp, _vp, _vp, _vp]),
    "zest_sample_pdf_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "zest_ndc_fwd": (_i, [_vp, _i, _vp, _vp, _f, _f, _f, _f, _i, _i, _vp, _vp]),
    "zest_composite_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "zest_composite_blend_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _vp]),
    "zest_encode_bwd": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "zest_volume_from_cl": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "zest_mlp_train_saved_floats": (_sz, [C.POINTER(MlpDesc), _i]),
    "zest_mlp_train_workspace_floats": (_sz, [C.POINTER(MlpDesc), _i]),
    "zest_mlp_train_fwd": (_i, [C.POINTER(MlpDesc), C.POINTER(_vp), _vp, _i, _vp, _vp, _vp, _vp]),
    "zest_mlp_train_bwd": (_i, [C.POINTER(MlpDesc), C.POINTER(_vp), _vp, _i, _vp, _vp, _vp, _vp, _vp,
                                C.POINTER(_vp), _vp]),
    "zest_mlp_train16_stash_bytes": (_sz, [C.POINTER(MlpDesc), _i]),
    "zest_mlp_train16_work_bytes": (_sz, [C.POINTER(MlpDesc), _i]),
    "zest_mlp_train16_packed_bytes": (_sz, [C.POINTER(MlpDesc)]),
    "zest_mlp_train16_pack": (_i, [C.POINTER(MlpDesc), C.POINTER(_vp), _vp, _vp]),
    "zest_mlp_train16_fwd": (_i, [C.POINTER(MlpDesc), _vp, _vp, _i, _vp, _vp, _vp]),
    "zest_mlp_train16_bwd": (_i, [C.POINTER(MlpDesc), _vp, C.POINTER(_vp), _vp, _i, _vp, _vp, _vp, _vp, _vp,
                                  C.POINTER(_vp), _i, _vp]),
    "zest_mlp_packed_bytes": (_sz, [C.POINTER(MlpDesc), _i]),
    "zest_mlp_pack": (_i, [C.POINTER(MlpDesc), _i, C.POINTER(_vp), _vp, _vp]),
    "zest_pack_weights": (_i, [C.POINTER(MlpDesc), _i, C.POINTER(_vp), _vp, _vp]),       # = zest_mlp_pack
    "zest_mlp_fwd": (_i, [C.POINTER(MlpDesc), _i, _vp, _vp, _i, _vp, _vp]),
    "zest_render_fused_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, C.POINTER(MlpDesc), _vp,
                                   C.POINTER(ViewSet), C.POINTER(MlpDesc), _vp, C.POINTER(ViewSet),
                                   _f, _i, _i, _vp, _vp, _vp]),
    "zest_render_fused_workspace": (_sz, [_i, _i]),
    "zest_render_fused_set_passes": (_i, [_i]),
    "zest_render_fused_pass_shape": (_i, [_i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
}

_lib = None


def exported_symbols():
    """Names include/zest_render.h declares (used by the CPU symbol test)."""
    return list(_SIGS)


def lib():
    """Load the shared library once; raise if it is absent (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not found: build it with `python %s/build_hip.py` "
                               "(zest-nerf_amd has no non-HIP execution path)" % (LIB_PATH, _HERE))
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)           # AttributeError if the header and library disagree
            fn.restype, fn.argtypes = res, args
        if L.zest_abi_version() != 3:
            raise RuntimeError("libzest_hip.so ABI %d != 3: rebuild it (python zest-nerf_amd/build_hip.py)"
                               % L.zest_abi_version())
        _lib = L
    return _lib


def _check(code, what):
    if code != 0:
        msg = lib().zest_last_error().decode(errors="replace")
        raise RuntimeError("%s failed (hipError %d): %s" % (what, code, msg))


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(t, name, shape=None):
    """Normalise to a contiguous fp32 device tensor; reject CPU tensors loudly.  shape: the extents the
    kernel will index with (None entries: any) - the C ABI sees pointers and sizes only, so a tensor of
    another shape would be read out of bounds: refuse it here, with both shapes in the message."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("zest_hip: %s is on %s; this path runs only on a HIP device" % (name, t.device))
    if shape is not None and (t.dim() != len(shape) or any(w is not None and int(w) != g for w, g in zip(shape, t.shape))):
        raise RuntimeError("zest_hip: %s has shape %s, expected %s" % (
            name, tuple(t.shape), tuple("*" if w is None else int(w) for w in shape)))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ------------------------------------------------------------------------ compositing
def _spacing(z, rays_dir, dists):
    """The sample spacing is either rebuilt in the kernel from z and |rays_dir| (depth2dist) or taken
    from the caller's `dists` [R,S]."""
    if z.dim() != 2:
        raise RuntimeError("zest_hip: z has shape %s, expected (R, S)" % (tuple(z.shape),))
    rays_dir, dists = _dev(rays_dir, "rays_dir", (z.shape[0], 3)), _dev(dists, "dists", z.shape)
    if rays_dir is None and dists is None:
        raise RuntimeError("zest_hip: compositing needs rays_dir or dists")
    return rays_dir, dists


def composite(raw, z, rays_dir, noise=None, noise_std=0.0, white_bkgd=False, want_disp=True, dists=None):
    """raw [R,S,4], z [R,S], rays_dir [R,3] (or dists [R,S]) -> rgb_map, disp, acc, weights, depth, alpha."""
    z = _dev(z, "z")
    rays_dir, dists = _spacing(z, rays_dir, dists)
    R, S = z.shape
    raw, noise = _dev(raw, "raw", (R, S, 4)), _dev(noise, "noise", (R, S))
    o = lambda *s: torch.empty(*s, device=z.device, dtype=torch.float32)
    rgb, depth, acc, disp, w, a = o(R, 3), o(R), o(R), o(R), o(R, S), o(R, S)
    _check(lib().zest_composite_fwd(_ptr(raw), _ptr(z), _ptr(rays_dir), _ptr(dists), _ptr(noise), float(noise_std),
                                    int(bool(white_bkgd)), R, S, _ptr(rgb), _ptr(depth), _ptr(acc),
                                    _ptr(disp), _ptr(w), _ptr(a), _stream(z)), "zest_composite_fwd")
    return rgb, disp, acc, w, depth, a


def composite_blend(raw_dy, raw_st, blend, z, rays_dir, noise=None, noise_std=0.0, dists=None):
    z = _dev(z, "z")
    rays_dir, dists = _spacing(z, rays_dir, dists)
    R, S = z.shape
    raw_dy, raw_st = _dev(raw_dy, "raw_dy", (R, S, 4)), _dev(raw_st, "raw_st", (R, S, 4))
    blend, noise = _dev(blend, "blend", (R, S)), _dev(noise, "noise", (R, S))
    o = lambda *s: torch.empty(*s, device=z.device, dtype=torch.float32)
    rgb, depth, rgb_fg, depth_fg, w_fg, w_dy, dd = o(R, 3), o(R), o(R, 3), o(R), o(R, S), o(R, S), o(R)
    _check(lib().zest_composite_blend_fwd(_ptr(raw_dy), _ptr(raw_st), _ptr(blend), _ptr(z),
                                          _ptr(rays_dir), _ptr(dists), _ptr(noise), float(noise_std), R, S,
                                          _ptr(rgb), _ptr(depth), _ptr(rgb_fg), _ptr(depth_fg),
                                          _ptr(w_fg), _ptr(w_dy), _ptr(dd), _stream(z)),
           "zest_composite_blend_fwd")
    return rgb, depth, rgb_fg, depth_fg, w_fg, w_dy, dd


def weighted_complement_sum(w, p):
    w = _dev(w, "w", (None, None))
    R, S = w.shape
    p = _dev(p, "p", (R, S))
    out = torch.empty(R, device=w.device, dtype=torch.float32)
    _check(lib().zest_weighted_complement_sum(_ptr(w), _ptr(p), R, S, _ptr(out), _stream(w)),
           "zest_weighted_complement_sum")
    return out


# --------------------------------------------------------------------- per-sample operators
def embed(x, n_freqs):
    x = _dev(x, "x")
    Cn = x.shape[-1]
    M = x.numel() // Cn
    y = torch.empty(*x.shape[:-1], Cn * (2 * n_freqs + 1), device=x.device, dtype=torch.float32)
    _check(lib().zest_embed_fwd(_ptr(x), M, Cn, n_freqs, _ptr(y), _stream(x)), "zest_embed_fwd")
    return y


def volume_to_cl(vol):
    """[1,8,D,H,W] or [8,D,H,W] -> the kernels' copy: channels-last, depth innermost [H,W,D,8]."""
    vol = _dev(vol, "volume")
    if vol.dim() == 5:
        if vol.shape[0] != 1:
            raise RuntimeError("zest_hip: volume batch must be 1, got %d" % vol.shape[0])
        vol = vol[0]
    if vol.shape[0] != 8:
        raise RuntimeError("zest_hip: encoding volume must have 8 channels, got %d" % vol.shape[0])
    _, D, H, W = vol.shape
    out = torch.empty(H, W, D, 8, device=vol.device, dtype=torch.float32)
    _check(lib().zest_volume_to_cl(_ptr(vol), D, H, W, _ptr(out), _stream(vol)), "zest_volume_to_cl")
    return out


def distortion(weights, t_vals, want_grad=True):
    """weights [R,S], t_vals [1,S] or [R,S] -> (loss_ray [R], d loss_ray / d weights [R,S] or None)."""
    weights = _dev(weights, "ray_weights", (None, None))
    R, S = weights.shape
    t_vals = _dev(t_vals, "t_vals", (None, S))
    if t_vals.shape[0] not in (1, R):
        raise RuntimeError("zest_hip: t_vals has %d rows, expected 1 or %d" % (t_vals.shape[0], R))
    loss = torch.empty(R, device=weights.device, dtype=torch.float32)
    grad = torch.empty(R, S, device=weights.device, dtype=torch.float32) if want_grad else None
    _check(lib().zest_distortion_fwd(_ptr(weights), _ptr(t_vals), t_vals.shape[0], R, S, _ptr(loss), _ptr(grad),
                                     _stream(weights)), "zest_distortion_fwd")
    return loss, grad


def project_rays(weights, pts, w2c, H, W, focal):
    """weights [R,S], pts [R,S,3], w2c [4,4] (or [3,4]) -> [R,2]."""
    weights = _dev(weights, "weights_ref", (None, None))
    R, S = weights.shape
    pts, w2c = _dev(pts, "raw_pts", (R, S, 3)), _dev(w2c, "w2c", (None, 4))
    if w2c.shape[0] < 3:
        raise RuntimeError("zest_hip: w2c has shape %s, expected (3 or 4, 4)" % (tuple(w2c.shape),))
    out = torch.empty(R, 2, device=weights.device, dtype=torch.float32)
    _check(lib().zest_project_rays_fwd(_ptr(weights), _ptr(pts), _ptr(w2c), int(H), int(W), float(focal), R, S,
                                       _ptr(out), _stream(weights)), "zest_project_rays_fwd")
    return out


def project_rays_bwd(weights, pts, w2c, H, W, focal, grad_out, want_w=True, want_pts=True):
    weights = _dev(weights, "weights_ref", (None, None))
    R, S = weights.shape
    pts, w2c, grad_out = _dev(pts, "raw_pts", (R, S, 3)), _dev(w2c, "w2c", (None, 4)), _dev(grad_out, "grad", (R, 2))
    if w2c.shape[0] < 3:
        raise RuntimeError("zest_hip: w2c has shape %s, expected (3 or 4, 4)" % (tuple(w2c.shape),))
    dw = torch.empty(R, S, device=weights.device, dtype=torch.float32) if want_w else None
    dp = torch.empty(R, S, 3, device=weights.device, dtype=torch.float32) if want_pts else None
    _check(lib().zest_project_rays_bwd(_ptr(weights), _ptr(pts), _ptr(w2c), int(H), int(W), float(focal),
                                       _ptr(grad_out), R, S, _ptr(dw), _ptr(dp), _stream(weights)),
           "zest_project_rays_bwd")
    return dw, dp


def nchw_to_nhwc(x):
    """[N,C,H,W] -> [N,H,W,C]."""
    x = _dev(x, "x")
    N, Cc, H, W = x.shape
    out = torch.empty(N, H, W, Cc, device=x.device, dtype=torch.float32)
    _check(lib().zest_nchw_to_nhwc(_ptr(x), N, Cc, H, W, _ptr(out), _stream(x)), "zest_nchw_to_nhwc")
    return out


def volume_cost(feats, imgs_lr, proj, depth, pad=0, return_feats_cl=False):
    """Plane-sweep cost volume.  feats [V,32,H,W]; imgs_lr [V,3,H,W] (at feature resolution);
    proj [V-1,3,4]; depth [D] -> img_feat [3V+32, D, H+2pad, W+2pad], in_masks [V, D, Hp, Wp]
    (and, on request, the channels-last feature maps the backward pass gathers from again)."""
    feats, imgs_lr = _dev(feats, "feats"), _dev(imgs_lr, "imgs")
    proj, depth = _dev(proj, "proj_mats"), _dev(depth, "depth_values")
    V, Cc, H, W = feats.shape
    if tuple(imgs_lr.shape) != (V, 3, H, W) or tuple(proj.shape) != (V - 1, 3, 4) or depth.dim() != 1:
        raise RuntimeError("zest_hip.volume_cost: feats %s imgs %s proj %s depth %s"
                           % (tuple(feats.shape), tuple(imgs_lr.shape), tuple(proj.shape), tuple(depth.shape)))
    D, Hp, Wp = depth.shape[0], H + 2 * pad, W + 2 * pad
    fcl, icl = nchw_to_nhwc(feats), images_to_cl(imgs_lr)
    img_feat = torch.empty(3 * V + Cc, D, Hp, Wp, device=feats.device, dtype=torch.float32)
    masks = torch.empty(V, D, Hp, Wp, device=feats.device, dtype=torch.float32)
    _check(lib().zest_volume_cost_fwd(_ptr(fcl), _ptr(icl), _ptr(proj), _ptr(depth), V, Cc, D, H, W, pad,
                                      _ptr(img_feat), _ptr(masks), _stream(feats)), "zest_volume_cost_fwd")
    return (img_feat, masks, fcl) if return_feats_cl else (img_feat, masks)


COST_CL_CHANNELS = 48            # channels of the channels-last cost volume (41 used for V = 3)


def volume_cost_cl(feats, imgs_lr, proj, depth, pad=0, feats_cl=None):
    """The plane-sweep cost volume for the HIP regularisation net: [D, H+2pad, W+2pad, 48] channels-last
    (channels 3V+32 .. 47 are zero), no masks.  Arguments as volume_cost; feats_cl [V,H,W,32]: the features already
    channels-last (FeatureNet.forward_hip) instead of `feats`."""
    imgs_lr = _dev(imgs_lr, "imgs")
    proj, depth = _dev(proj, "proj_mats"), _dev(depth, "depth_values")
    if feats_cl is not None:
        feats_cl = _dev(feats_cl, "feats_cl")
        feats = feats_cl.permute(0, 3, 1, 2)                 # shape bookkeeping only
    else:
        feats = _dev(feats, "feats")
    V, Cc, H, W = feats.shape
    if tuple(imgs_lr.shape) != (V, 3, H, W) or tuple(proj.shape) != (V - 1, 3, 4) or depth.dim() != 1:
        raise RuntimeError("zest_hip.volume_cost_cl: feats %s imgs %s proj %s depth %s"
                           % (tuple(feats.shape), tuple(imgs_lr.shape), tuple(proj.shape), tuple(depth.shape)))
    D, Hp, Wp = depth.shape[0], H + 2 * pad, W + 2 * pad
    fcl, icl = (feats_cl if feats_cl is not None else nchw_to_nhwc(feats)), images_to_cl(imgs_lr)
    out = torch.empty(D, Hp, Wp, COST_CL_CHANNELS, device=feats.device, dtype=torch.float32)
    _check(lib().zest_volume_cost_cl_fwd(_ptr(fcl), _ptr(icl), _ptr(proj), _ptr(depth), V, Cc, D, H, W, pad,
                                         _ptr(out), _stream(feats)), "zest_volume_cost_cl_fwd")
    return out


def costreg_stat_rows():
    return int(lib().zest_costreg_stat_rows())


def costreg_stats(cout, device):
    """A table of batch statistics for one layer: [rows, 2, cout] float64 (a row per workgroup of the kernel that
    fills it, and a last one with the number of rows in use)."""
    return torch.empty(costreg_stat_rows(), 2, cout, device=device, dtype=torch.float64)


def _check_stats(stats, cout, who):
    if tuple(stats.shape) != (costreg_stat_rows(), 2, cout) or stats.dtype != torch.float64 or not stats.is_contiguous():
        raise RuntimeError("zest_hip.%s: stats %s %s for %d channels" % (who, tuple(stats.shape), stats.dtype, cout))


def costreg_conv(x, pre, w_packed, cout, stride, passes, stats):
    """x [D,H,W,cin] raw channels-last; pre [2,cin] or None; -> raw [Do,Ho,Wo,cout]; stats: costreg_stats(cout)."""
    x = _dev(x, "x")
    D, H, W, cin = x.shape
    need = int(lib().zest_costreg_packed_bytes(cin, cout, passes))
    if w_packed.numel() * w_packed.element_size() != need or not w_packed.is_cuda:
        raise RuntimeError("zest_hip.costreg_conv: packed weights of %d bytes, %d -> %d channels in %d passes take %d"
                           % (w_packed.numel() * w_packed.element_size(), cin, cout, passes, need))
    if pre is not None and (tuple(pre.shape) != (2, cin) or pre.dtype != torch.float32 or not pre.is_contiguous()):
        raise RuntimeError("zest_hip.costreg_conv: pre %s for %d channels" % (tuple(pre.shape), cin))
    _check_stats(stats, cout, "costreg_conv")
    o = lambda n: (n - 1) // stride + 1
    out = torch.empty(o(D), o(H), o(W), cout, device=x.device, dtype=torch.float32)
    _check(lib().zest_costreg_conv_fwd(_ptr(x), _ptr(pre), _ptr(w_packed), cin, cout, stride, passes, D, H, W,
                                       _ptr(out), _ptr(stats), _stream(x)), "zest_costreg_conv_fwd")
    return out


def conv2d_cl(x, pre, w_packed, cout, k, stride, passes, stats):
    """x [N,H,W,cin] raw channels-last; pre [2,cin] or None; -> raw [N,Ho,Wo,cout] (Conv2d k x k, padding k//2, no bias,
    on act(norm(x))); stats: costreg_stats(cout)."""
    x = _dev(x, "x")
    N, H, W, cin = x.shape
    need = int(lib().zest_conv2d_packed_bytes(cin, cout, k, passes))
    if w_packed.numel() * w_packed.element_size() != need or not w_packed.is_cuda:
        raise RuntimeError("zest_hip.conv2d_cl: packed weights of %d bytes, %d -> %d channels, k %d, %d passes take %d"
                           % (w_packed.numel() * w_packed.element_size(), cin, cout, k, passes, need))
    if pre is not None and (tuple(pre.shape) != (2, cin) or pre.dtype != torch.float32 or not pre.is_contiguous()):
        raise RuntimeError("zest_hip.conv2d_cl: pre %s for %d channels" % (tuple(pre.shape), cin))
    _check_stats(stats, cout, "conv2d_cl")
    o = lambda n: (n - 1) // stride + 1
    out = torch.empty(N, o(H), o(W), cout, device=x.device, dtype=torch.float32)
    _check(lib().zest_conv2d_fwd(_ptr(x), _ptr(pre), _ptr(w_packed), cin, cout, k, stride, passes, N, H, W,
                                 _ptr(out), _ptr(stats), _stream(x)), "zest_conv2d_fwd")
    return out


def costreg_deconv(x0, pre0, x1, pre1, w_packed, cout, passes, stats):
    """act(norm(x0)) [+ act(norm(x1))] [D,H,W,cin] -> raw [2D,2H,2W,cout]; w_packed: CostRegNet._pack_deconv."""
    x0 = _dev(x0, "x0")
    D, H, W, cin = x0.shape
    x1 = _dev(x1, "x1", (D, H, W, cin))
    for nm, t in (("pre0", pre0), ("pre1", pre1)):
        if t is not None and (tuple(t.shape) != (2, cin) or t.dtype != torch.float32 or not t.is_contiguous()):
            raise RuntimeError("zest_hip.costreg_deconv: %s %s for %d channels" % (nm, tuple(t.shape), cin))
    need = int(lib().zest_costreg_deconv_packed_bytes(cin, cout, passes))
    if w_packed.numel() * w_packed.element_size() != need or not w_packed.is_cuda:
        raise RuntimeError("zest_hip.costreg_deconv: packed weights of %d bytes, %d -> %d channels in %d passes take %d"
                           % (w_packed.numel() * w_packed.element_size(), cin, cout, passes, need))
    _check_stats(stats, cout, "costreg_deconv")
    out = torch.empty(2 * D, 2 * H, 2 * W, cout, device=x0.device, dtype=torch.float32)
    _check(lib().zest_costreg_deconv_fwd(_ptr(x0), _ptr(pre0), _ptr(x1), _ptr(pre1), _ptr(w_packed), cin, cout, passes,
                                         D, H, W, _ptr(out), _ptr(stats), _stream(x0)), "zest_costreg_deconv_fwd")
    return out


def costreg_bn(stats, count, bn, batch_stats, pre, moments=None):
    """pre [2,C] <- scale / shift of the batch norm module `bn` (weight, bias, running_*, eps, momentum) from the
    batch statistics `stats` [2,C] of `count` voxels (and the running estimates updated) or from the running ones."""
    Cn = pre.shape[1]
    if batch_stats:
        _check_stats(stats, Cn, "costreg_bn")
    track = bn.running_mean is not None
    if not batch_stats and not track:
        raise RuntimeError("zest_hip.costreg_bn: a norm without running statistics needs batch statistics")
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    _check(lib().zest_costreg_bn(_ptr(stats), Cn, int(count), _ptr(bn.weight), _ptr(bn.bias), float(bn.eps),
                                 1 if batch_stats else 0, _ptr(bn.running_mean) if track else None,
                                 _ptr(bn.running_var) if track else None, mom,
                                 _ptr(bn.num_batches_tracked) if (track and batch_stats) else None, _ptr(pre),
                                 _ptr(moments), _stream(pre)), "zest_costreg_bn")
    return pre


def costreg_bn_bwd(raw, g_act, pre, moments, gamma):
    """Backward of act(norm(raw)) with batch statistics: raw, g_act [..., C] channels-last -> (g_raw like raw,
    g_gamma [C], g_beta [C])."""
    raw, g_act = _dev(raw, "raw"), _dev(g_act, "g_act", tuple(raw.shape))
    Cn = raw.shape[-1]
    M = raw.numel() // Cn
    stats = costreg_stats(Cn, raw.device)
    totals = torch.empty(2, Cn, device=raw.device, dtype=torch.float32)
    g_raw = torch.empty_like(raw)
    _check(lib().zest_costreg_bn_bwd(_ptr(raw), _ptr(g_act), _ptr(pre), _ptr(moments), _ptr(gamma), Cn, M, _ptr(stats),
                                     _ptr(totals), _ptr(g_raw), _stream(raw)), "zest_costreg_bn_bwd")
    return g_raw, totals[1], totals[0]


def costreg_out(raw_a, pre_a, raw_b, pre_b):
    """act(norm(raw_a)) + act(norm(raw_b)), [D,H,W,8] each -> the encoding volume [1,8,D,H,W]."""
    raw_a = _dev(raw_a, "raw_a")
    D, H, W, Cn = raw_a.shape
    raw_b = _dev(raw_b, "raw_b", (D, H, W, 8))
    if Cn != 8 or tuple(pre_a.shape) != (2, 8) or tuple(pre_b.shape) != (2, 8):
        raise RuntimeError("zest_hip.costreg_out: %s, %s, %s" % (tuple(raw_a.shape), tuple(pre_a.shape), tuple(pre_b.shape)))
    out = torch.empty(1, 8, D, H, W, device=raw_a.device, dtype=torch.float32)
    _check(lib().zest_costreg_out(_ptr(raw_a), _ptr(pre_a), _ptr(raw_b), _ptr(pre_b), D, H, W, _ptr(out),
                                  _stream(raw_a)), "zest_costreg_out")
    return out


def volume_cost_bwd(feats_cl, proj, depth, pad, g_img_feat):
    """g_img_feat [3V+32, D, Hp, Wp] -> gradient of the feature maps [V,32,H,W] (a channels-first view of
    the channels-last buffer the kernel scatters into)."""
    g_img_feat, proj, depth = _dev(g_img_feat, "g_img_feat"), _dev(proj, "proj_mats"), _dev(depth, "depth_values")
    V, H, W, Cc = feats_cl.shape
    D = depth.shape[0]
    if tuple(g_img_feat.shape) != (3 * V + Cc, D, H + 2 * pad, W + 2 * pad):
        raise RuntimeError("zest_hip.volume_cost_bwd: gradient %s for V=%d D=%d H=%d W=%d pad=%d"
                           % (tuple(g_img_feat.shape), V, D, H, W, pad))
    g = torch.zeros_like(feats_cl)
    _check(lib().zest_volume_cost_bwd(_ptr(feats_cl), _ptr(proj), _ptr(depth), V, Cc, D, H, W, pad, _ptr(g_img_feat),
                                      _ptr(g), _stream(feats_cl)), "zest_volume_cost_bwd")
    return g.permute(0, 3, 1, 2)


def homo_warp_bwd(g_warped, src_shape, proj=None, depth=None, grid=None, pad=0):
    """g_warped [C,D,Hp,Wp] -> g_src [C,H,W]; grid [D,Hp,Wp,2] or proj [3,4] + depth [D] as in homo_warp."""
    g_warped = _dev(g_warped, "g_warped")
    Cc, H, W = src_shape
    D, Hp, Wp = g_warped.shape[1:]
    g = torch.zeros(Cc, H, W, device=g_warped.device, dtype=torch.float32)
    gin = _dev(grid, "src_grid") if grid is not None else None
    proj, depth = (None, None) if gin is not None else (_dev(proj, "proj_mat"), _dev(depth, "depth_values"))
    _check(lib().zest_homo_warp_bwd(_ptr(proj), _ptr(depth), _ptr(gin), Cc, D, H, W, Hp, Wp, pad, _ptr(g_warped),
                                    _ptr(g), _stream(g_warped)), "zest_homo_warp_bwd")
    return g


def homo_warp(src, proj=None, depth=None, grid=None, pad=0):
    """src [C,H,W]; proj [3,4] + depth [D], or grid [D,Hp,Wp,2] -> warped [C,D,Hp,Wp], grid."""
    src = _dev(src, "src_feat")
    Cc, H, W = src.shape
    if grid is None:
        proj, depth = _dev(proj, "proj_mat"), _dev(depth, "depth_values")
        D, Hp, Wp = depth.shape[0], H + 2 * pad, W + 2 * pad
        grid_out = torch.empty(D, Hp, Wp, 2, device=src.device, dtype=torch.float32)
        gin = None
    else:
        gin = _dev(grid, "src_grid")
        D, Hp, Wp = gin.shape[:3]
        grid_out = gin
    warped = torch.empty(Cc, D, Hp, Wp, device=src.device, dtype=torch.float32)
    _check(lib().zest_homo_warp_fwd(_ptr(src), _ptr(proj) if gin is None else None,
                                    _ptr(depth) if gin is None else None, _ptr(gin) if gin is not None else None,
                                    Cc, D, H, W, Hp, Wp, pad, _ptr(warped),
                                    _ptr(grid_out) if gin is None else None, _stream(src)), "zest_homo_warp_fwd")
    return warped, grid_out


def images_to_cl(imgs):
    """[1,V,3,H,W] or [V,3,H,W] -> [V,H,W,4]."""
    imgs = _dev(imgs, "imgs")
    if imgs.dim() == 5:
        if imgs.shape[0] != 1:
            raise RuntimeError("zest_hip: image batch must be 1, got %d" % imgs.shape[0])
        imgs = imgs[0]
    V, c, H, W = imgs.shape
    if c != 3:
        raise RuntimeError("zest_hip: images must have 3 channels, got %d" % c)
    out = torch.empty(V, H, W, 4, device=imgs.device, dtype=torch.float32)
    _check(lib().zest_images_to_cl(_ptr(imgs), V, H, W, _ptr(out), _stream(imgs)), "zest_images_to_cl")
    return out


def volume_lookup(vol_cl, ndc):
    ndc = _dev(ndc, "ndc", (None,) * (ndc.dim() - 1) + (3,))
    vol_cl = _dev(vol_cl, "vol_cl", (None, None, None, 8))
    H, W, D, _ = vol_cl.shape
    M = ndc.numel() // 3
    out = torch.empty(*ndc.shape[:-1], 8, device=ndc.device, dtype=torch.float32)
    _check(lib().zest_volume_lookup_fwd(_ptr(vol_cl), D, H, W, _ptr(ndc), M, _ptr(out), _stream(ndc)),
           "zest_volume_lookup_fwd")
    return out


def color_lookup(imgs_cl, w2cs, intrinsics, pts):
    pts = _dev(pts, "pts", (None,) * (pts.dim() - 1) + (3,))
    imgs_cl = _dev(imgs_cl, "imgs_cl", (None, None, None, 4))
    w2cs = _dev(w2cs, "w2cs", (None,) * (w2cs.dim() - 2) + (4, 4))
    intrinsics = _dev(intrinsics, "intrinsics", (None,) * (intrinsics.dim() - 2) + (3, 3))
    V, H, W, _ = imgs_cl.shape
    if w2cs.numel() < 16 * V or intrinsics.numel() < 9 * V:
        raise RuntimeError("zest_hip: %d views but only %d poses" % (V, w2cs.shape[-3]))
    M = pts.numel() // 3
    out = torch.empty(*pts.shape[:-1], 4 * V, device=pts.device, dtype=torch.float32)
    _check(lib().zest_color_lookup_fwd(_ptr(imgs_cl), V, H, W, _ptr(w2cs), _ptr(intrinsics), _ptr(pts),
                                       M, _ptr(out), _stream(pts)), "zest_color_lookup_fwd")
    return out


def encode(ndc, pts, rays_dir, t=None, vol_cl=None, imgs_cl=None, w2cs=None, intrinsics=None, out=None):
    """ndc, pts [R,S,3]; rays_dir [R,3] -> x [R,S,C_in] (reference prepare_pts layout); out: a contiguous
    [R,S,C_in] fp32 tensor to write into (e.g. half of a larger batch) instead of a new one."""
    ndc = _dev(ndc, "ndc", (None, None, 3))
    R, S, _ = ndc.shape
    pts, rays_dir = _dev(pts, "pts", (R, S, 3)), _dev(rays_dir, "rays_dir", (R, 3))
    w2cs, intrinsics = _dev(w2cs, "w2cs"), _dev(intrinsics, "intrinsics")
    has_t = t is not None
    D = Hv = Wv = V = H = W = 0
    if vol_cl is not None:
        vol_cl, imgs_cl = _dev(vol_cl, "vol_cl", (None, None, None, 8)), _dev(imgs_cl, "imgs_cl", (None, None, None, 4))
        Hv, Wv, D, _ = vol_cl.shape
        V, H, W, _ = imgs_cl.shape
        if pts is None or w2cs is None or intrinsics is None or w2cs.numel() < 16 * V or intrinsics.numel() < 9 * V:
            raise RuntimeError("zest_hip: encode with features needs pts and one pose (4x4) and intrinsic (3x3) per source view")
    if w2cs is not None and w2cs.shape[-2:] != (4, 4):
        raise RuntimeError("zest_hip: w2cs has shape %s, expected (..., 4, 4)" % (tuple(w2cs.shape),))
    c_in = (4 if has_t else 3) * 21 + (8 + 4 * V if vol_cl is not None else 0) + 27
    x = torch.empty(R, S, c_in, device=ndc.device, dtype=torch.float32) if out is None else _dev(out, "out", (R, S, c_in))
    _check(lib().zest_encode_fwd(_ptr(ndc), _ptr(pts), _ptr(rays_dir), R, S, int(has_t),
                                 float(t) if has_t else 0.0, _ptr(vol_cl), D, Hv, Wv, _ptr(imgs_cl),
                                 V, H, W, _ptr(w2cs), _ptr(intrinsics), _ptr(x), _stream(ndc)),
           "zest_encode_fwd")
    return x


# ----------------------------------------------------------------------------- ray sampling
def build_rays(xs, ys, t_rand, S, k_tgt, c2w_tgt, w2c_ref, k_ref, nf_tgt, nf_ref, pad, W, H):
    """xs, ys [R] pixel coordinates; camera matrices and (near, far) pairs as device tensors
    (views into the batch dict) -> rays_dir [R,3], depth [R,S], pts, ndc [R,S,3]."""
    xs = _dev(xs, "xs")
    R = xs.numel()
    ys, t_rand = _dev(ys, "ys", xs.shape), _dev(t_rand, "t_rand")
    if t_rand is not None and (t_rand.dim() != 2 or t_rand.shape[0] < R or t_rand.shape[1] != S):
        raise RuntimeError("zest_hip: t_rand has shape %s, expected (>= %d, %d)" % (tuple(t_rand.shape), R, S))
    mats = [_dev(m, n, sh) for m, n, sh in ((k_tgt, "k_tgt", (3, 3)), (c2w_tgt, "c2w_tgt", (4, 4)), (w2c_ref, "w2c_ref", (4, 4)),
                                           (k_ref, "k_ref", (3, 3)), (nf_tgt, "near_far_tgt", (2,)), (nf_ref, "near_far_ref", (2,)))]
    o = lambda *s: torch.empty(*s, device=xs.device, dtype=torch.float32)
    d, z, pts, ndc = o(R, 3), o(R, S), o(R, S, 3), o(R, S, 3)
    _check(lib().zest_build_rays_fwd(_ptr(xs), _ptr(ys), _ptr(t_rand), R, int(S), *[_ptr(m) for m in mats],
                                     int(pad), int(W), int(H), _ptr(d), _ptr(z), _ptr(pts), _ptr(ndc),
                                     _stream(xs)), "zest_build_rays_fwd")
    return d, z, pts, ndc


def sample_pdf(bins, weights, n_samples=None, u=None):
    """bins [R, Nb+1], weights [R, Nb], u [R, Ns] (or None + n_samples: deterministic) -> samples [R, Ns]."""
    weights = _dev(weights, "weights", (None, None))
    R, Nb = weights.shape
    bins, u = _dev(bins, "bins", (R, Nb + 1)), _dev(u, "u", (R, None))
    Ns = u.shape[1] if u is not None else int(n_samples)
    out = torch.empty(R, Ns, device=bins.device, dtype=torch.float32)
    _check(lib().zest_sample_pdf_fwd(_ptr(bins), _ptr(weights), _ptr(u), R, Nb, Ns, _ptr(out), _stream(bins)),
           "zest_sample_pdf_fwd")
    return out


def ndc_coordinate(pts, w2c, k, inv_w, inv_h, near, far, pad=0, lindisp=False):
    pts = _dev(pts, "pts", (None,) * (pts.dim() - 1) + (3,))
    w2c, k = _dev(w2c, "w2c", (4, 4)), _dev(k, "k", (3, 3))
    M = pts.numel() // 3
    out = torch.empty_like(pts)
    _check(lib().zest_ndc_fwd(_ptr(pts), M, _ptr(w2c), _ptr(k), float(inv_w), float(inv_h), float(near),
                              float(far), int(pad), int(bool(lindisp)), _ptr(out), _stream(pts)),
           "zest_ndc_fwd")
    return out


def composite_bwd(raw, z, rays_dir, noise, noise_std, white_bkgd, g_rgb, g_depth, g_acc, g_weights, dists=None):
    z = _dev(z, "z")
    rays_dir, dists = _spacing(z, rays_dir, dists)
    R, S = z.shape
    raw, noise = _dev(raw, "raw", (R, S, 4)), _dev(noise, "noise", (R, S))
    gs = [_dev(g, n, sh) for g, n, sh in ((g_rgb, "g_rgb", (R, 3)), (g_depth, "g_depth", (R,)), (g_acc, "g_acc", (R,)),
                                          (g_weights, "g_weights", (R, S)))]
    g_raw = torch.empty(R, S, 4, device=z.device, dtype=torch.float32)
    _check(lib().zest_composite_bwd(_ptr(raw), _ptr(z), _ptr(rays_dir), _ptr(dists), _ptr(noise), float(noise_std),
                                    int(bool(white_bkgd)), R, S, *[_ptr(g) for g in gs], _ptr(g_raw),
                                    _stream(z)), "zest_composite_bwd")
    return g_raw


def composite_blend_bwd(raw_dy, raw_st, blend, z, rays_dir, noise, noise_std, g_rgb, g_depth, g_rgb_fg,
                        g_depth_fg, g_wfg, g_wd, dists=None):
    z = _dev(z, "z")
    rays_dir, dists = _spacing(z, rays_dir, dists)
    R, S = z.shape
    raw_dy, raw_st = _dev(raw_dy, "raw_dy", (R, S, 4)), _dev(raw_st, "raw_st", (R, S, 4))
    blend, noise = _dev(blend, "blend", (R, S)), _dev(noise, "noise", (R, S))
    gs = [_dev(g, n, sh) for g, n, sh in ((g_rgb, "g_rgb", (R, 3)), (g_depth, "g_depth", (R,)), (g_rgb_fg, "g_rgb_fg", (R, 3)),
                                          (g_depth_fg, "g_depth_fg", (R,)), (g_wfg, "g_weights_fg", (R, S)),
                                          (g_wd, "g_weights_dy", (R, S)))]
    o = lambda *s: torch.empty(*s, device=z.device, dtype=torch.float32)
    g_dy, g_st, g_b = o(R, S, 4), o(R, S, 4), o(R, S)
    _check(lib().zest_composite_blend_bwd(_ptr(raw_dy), _ptr(raw_st), _ptr(blend), _ptr(z), _ptr(rays_dir),
                                          _ptr(dists), _ptr(noise), float(noise_std), R, S, *[_ptr(g) for g in gs],
                                          _ptr(g_dy), _ptr(g_st), _ptr(g_b), _stream(z)),
           "zest_composite_blend_bwd")
    return g_dy, g_st, g_b


# ------------------------------------------------------------------- training path (backward)
def encode_bwd(g_x, ndc, t, vol_cl, V, want_vol_grad, g_vol=None):
    """-> g_ndc [R,S,3], g_vol_cl [H,W,D,8] or None.  g_vol: an existing volume gradient to add into (the kernel
    scatter-adds) instead of a fresh zero-filled one."""
    ndc = _dev(ndc, "ndc", (None, None, 3))
    R, S, _ = ndc.shape
    c_in = (4 if t is not None else 3) * 21 + (8 + 4 * int(V) if vol_cl is not None else 0) + 27
    g_x = _dev(g_x, "g_x", (R, S, c_in))
    D = Hv = Wv = 0
    add_to, g_vol = g_vol, None
    if vol_cl is not None:
        vol_cl = _dev(vol_cl, "vol_cl", (None, None, None, 8))
        Hv, Wv, D, _ = vol_cl.shape
        if want_vol_grad:
            g_vol = torch.zeros_like(vol_cl) if add_to is None else _dev(add_to, "g_vol", tuple(vol_cl.shape))
    g_ndc = torch.empty_like(ndc)
    _check(lib().zest_encode_bwd(_ptr(g_x), _ptr(ndc), R, S, int(t is not None), float(t) if t is not None else 0.0,
                                 _ptr(vol_cl), D, Hv, Wv, int(V), _ptr(g_ndc), _ptr(g_vol), _stream(ndc)),
           "zest_encode_bwd")
    return g_ndc, g_vol


def volume_from_cl(vol_cl):
    vol_cl = _dev(vol_cl, "vol_cl", (None, None, None, 8))
    H, W, D, _ = vol_cl.shape
    out = torch.empty(1, 8, D, H, W, device=vol_cl.device, dtype=torch.float32)
    _check(lib().zest_volume_from_cl(_ptr(vol_cl), D, H, W, _ptr(out), _stream(vol_cl)), "zest_volume_from_cl")
    return out


def _ptr_table(tensors):
    return (_vp * (2 * P_COUNT))(*[_ptr(t) for t in tensors])


def param_shapes(desc):
    """{ZEST_P_* slot: weight shape} of the Linears a net of this descriptor has (bias: (rows,))."""
    W, P, skips = desc.W, desc.in_ch_pts, desc.skips
    sh = {l: (W, P if l == 0 else W + (P if (l - 1) in skips else 0)) for l in range(desc.D)}
    if desc.use_feat:
        sh[8] = (W, desc.in_ch_feat)
    sh.update({9: (W // 2, W + desc.in_ch_views), 10: (W, W), 11: (1, W), 12: (3, W // 2)})
    if desc.head == HEAD_BLEND:
        sh[13] = (1, W)
    elif desc.head == HEAD_DYNAMIC:
        sh[13], sh[14] = (6, W), (2, W)
    return sh


def _params(desc, params):
    """The 2*P_COUNT parameter table as contiguous fp32 device tensors, every present tensor checked against the
    shape the packer / the training GEMMs will index it with (a state dict of another architecture must not be
    read out of bounds)."""
    if len(params) != 2 * P_COUNT:
        raise RuntimeError("zest_hip: parameter table has %d entries, expected %d" % (len(params), 2 * P_COUNT))
    want = param_shapes(desc)
    keep = []
    for i, p in enumerate(params):
        slot, is_bias = i // 2, i % 2
        if p is None:
            if slot in want:
                raise RuntimeError("zest_hip: parameter slot %d (%s) is required by this net" % (slot, "bias" if is_bias else "weight"))
            keep.append(None)
            continue
        if slot not in want:
            keep.append(None)            # a tensor the descriptor has no use for (e.g. pts_bias of a net without features)
            continue
        shape = (want[slot][0],) if is_bias else want[slot]
        keep.append(_dev(p, "parameter slot %d %s" % (slot, "bias" if is_bias else "weight"), shape))
    return keep


def _mlp_rows(desc, x, name="x"):
    x = _dev(x, name)
    if x.dim() < 1 or x.shape[-1] != desc.in_ch:
        raise RuntimeError("zest_hip: MLP expects %d input channels, got shape %s" % (desc.in_ch, tuple(x.shape)))
    return x


def mlp_train_fwd(desc, params, x):
    """params: 2*P_COUNT tensors-or-None.  -> out [M,C_out], saved (opaque activation stash)."""
    x = _mlp_rows(desc, x)
    M = x.numel() // x.shape[-1]
    L = lib()
    saved = torch.empty(int(L.zest_mlp_train_saved_floats(C.byref(desc), M)), device=x.device)
    work = torch.empty(int(L.zest_mlp_train_workspace_floats(C.byref(desc), M)), device=x.device)
    out = torch.empty(*x.shape[:-1], desc.out_ch, device=x.device, dtype=torch.float32)
    keep = _params(desc, params)
    _check(L.zest_mlp_train_fwd(C.byref(desc), _ptr_table(keep), _ptr(x), M, _ptr(saved), _ptr(work), _ptr(out),
                                _stream(x)), "zest_mlp_train_fwd")
    return out, saved


def mlp_train_bwd(desc, params, x, saved, out, g_out, want_gx=True):
    """-> g_x [.., C_in] or None, list of 2*P_COUNT parameter gradients (None where absent)."""
    x = _mlp_rows(desc, x)
    M = x.numel() // x.shape[-1]
    g_out, out = _dev(g_out, "g_out", x.shape[:-1] + (desc.out_ch,)), _dev(out, "out", x.shape[:-1] + (desc.out_ch,))
    L = lib()
    if saved.numel() < int(L.zest_mlp_train_saved_floats(C.byref(desc), M)):
        raise RuntimeError("zest_hip: `saved` is not the stash of this forward call (too small)")
    work = torch.empty(int(L.zest_mlp_train_workspace_floats(C.byref(desc), M)), device=x.device)
    keep = _params(desc, params)
    grads = [(torch.empty_like(p) if p is not None else None) for p in keep]
    g_x = torch.empty_like(x) if want_gx else None
    _check(L.zest_mlp_train_bwd(C.byref(desc), _ptr_table(keep), _ptr(x), M, _ptr(saved), _ptr(out), _ptr(g_out),
                                _ptr(work), _ptr(g_x), _ptr_table(grads), _stream(x)), "zest_mlp_train_bwd")
    return g_x, grads


# ------------------------------------------------ bf16 training path on the MFMA engine
def mlp_train16_pack_bwd(desc, params):
    """Transposed weight stream of the backward data kernel (params as for mlp_pack)."""
    dev = next(p for p in params if p is not None).device
    keep = _params(desc, params)
    nbytes = int(lib().zest_mlp_train16_packed_bytes(C.byref(desc)))
    if nbytes == 0:
        raise RuntimeError("zest_mlp_train16_packed_bytes: %s" % (lib().zest_last_error() or b"").decode())
    packed = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    _check(lib().zest_mlp_train16_pack(C.byref(desc), _ptr_table(keep), _ptr(packed),
                                       torch.cuda.current_stream(dev).cuda_stream), "zest_mlp_train16_pack")
    return packed


def mlp_train16_fwd(desc, packed_fwd, x):
    """x [M, C_in] -> out [M, C_out], stash (opaque: layer outputs as bf16 operand tiles + ReLU masks)."""
    x = _mlp_rows(desc, x)
    M = x.numel() // x.shape[-1]
    stash = torch.empty(int(lib().zest_mlp_train16_stash_bytes(C.byref(desc), M)), device=x.device, dtype=torch.uint8)
    out = torch.empty(*x.shape[:-1], desc.out_ch, device=x.device, dtype=torch.float32)
    _check(lib().zest_mlp_train16_fwd(C.byref(desc), _ptr(packed_fwd), _ptr(x), M, _ptr(stash), _ptr(out), _stream(x)),
           "zest_mlp_train16_fwd")
    return out, stash


def mlp_train16_bwd(desc, packed_bwd, params, x, stash, out, g_out, stages=7, work=None):
    """-> g_x [M, C_in] (direction columns zero), list of 2*P_COUNT fp32 parameter gradients (None where absent)."""
    x = _mlp_rows(desc, x)
    M = x.numel() // x.shape[-1]
    g_out, out = _dev(g_out, "g_out", x.shape[:-1] + (desc.out_ch,)), _dev(out, "out", x.shape[:-1] + (desc.out_ch,))
    if stash.numel() < int(lib().zest_mlp_train16_stash_bytes(C.byref(desc), M)):
        raise RuntimeError("zest_hip: `stash` is not the stash of this forward call (too small)")
    need = int(lib().zest_mlp_train16_work_bytes(C.byref(desc), M))
    if work is None or work.numel() < need:
        work = torch.empty(need, device=x.device, dtype=torch.uint8)
    keep = _params(desc, params)
    # the gradients are accumulated with atomics: one zero-filled buffer (one fill launch), views into it
    sizes = [(p.numel() + 3) // 4 * 4 if p is not None else 0 for p in keep]
    flat = torch.zeros(sum(sizes), device=x.device, dtype=torch.float32)
    grads, off = [], 0
    for p, n in zip(keep, sizes):
        grads.append(flat[off:off + p.numel()].view(p.shape) if p is not None else None)
        off += n
    # the finishing kernel writes every point and feature column of every row; the direction columns receive no
    # gradient (data) and are the only ones that need the zero
    g_x = torch.empty_like(x)
    if stages & 2:
        g_x[..., desc.in_ch_pts + (desc.in_ch_feat if desc.use_feat else 0):].zero_()
    else:
        g_x.zero_()
    _check(lib().zest_mlp_train16_bwd(C.byref(desc), _ptr(packed_bwd), _ptr_table(keep), _ptr(x), M, _ptr(stash), _ptr(out),
                                      _ptr(g_out), _ptr(work), _ptr(g_x), _ptr_table(grads), int(stages), _stream(x)),
           "zest_mlp_train16_bwd")
    return g_x, grads, work


# ----------------------------------------------------------------------------------- MLP
def mlp_packed_bytes(desc, precision):
    return int(lib().zest_mlp_packed_bytes(C.byref(desc), int(precision)))


def mlp_pack(desc, precision, params):
    """params: list of 2*P_COUNT tensors-or-None (weight, bias per ZEST_P_* slot)."""
    dev = next(p for p in params if p is not None).device
    nbytes = mlp_packed_bytes(desc, precision)
    if nbytes == 0:                      # shape / precision the kernels refuse: the library says which
        raise RuntimeError("zest_mlp_packed_bytes: %s" % (lib().zest_last_error() or b"").decode())
    keep = _params(desc, params)
    arr = (_vp * (2 * P_COUNT))(*[_ptr(p) for p in keep])
    packed = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    _check(lib().zest_mlp_pack(C.byref(desc), int(precision), arr, _ptr(packed),
                               torch.cuda.current_stream(dev).cuda_stream), "zest_mlp_pack")
    return packed


def mlp_fwd(desc, precision, packed, x):
    x = _mlp_rows(desc, x)
    M = x.numel() // x.shape[-1]
    if packed is None or packed.numel() * packed.element_size() < mlp_packed_bytes(desc, precision):
        raise RuntimeError("zest_hip: `packed` is not the packed weights of this net and precision (too small)")
    out = torch.empty(*x.shape[:-1], desc.out_ch, device=x.device, dtype=torch.float32)
    _check(lib().zest_mlp_fwd(C.byref(desc), int(precision), _ptr(packed), _ptr(x), M, _ptr(out),
                              _stream(x)), "zest_mlp_fwd")
    return out


def device_info():
    cu, khz = _i(0), _i(0)
    name = C.create_string_buffer(64)
    _check(lib().zest_device_info(C.byref(cu), C.byref(khz), name, 64), "zest_device_info")
    return dict(cu_count=cu.value, clock_khz=khz.value, arch=name.value.decode())


# --------------------------------------------------------------- state dict -> parameter table
_PARAM_SLOTS = [("pts_linears.%d" % i, i) for i in range(8)] + [
    ("pts_bias", 8), ("views_linears.0", 9), ("feature_linear", 10), ("alpha_linear", 11),
    ("rgb_linear", 12)]


def param_slots(desc):
    """(module name, ZEST_P_* slot) of the Linears every net of this shape has (heads apart)."""
    return [(n, s) for n, s in _PARAM_SLOTS
            if not (n == "pts_bias" and not desc.use_feat) and not (s < 8 and s >= desc.D)]


def param_table(state, desc, prefix="nerf."):
    """Order the nn.Linear tensors of a reference-layout state dict as zest_mlp_pack expects."""
    tab = [None] * (2 * P_COUNT)

    def put(slot, name):
        tab[2 * slot] = state[prefix + name + ".weight"]
        tab[2 * slot + 1] = state[prefix + name + ".bias"]
    for name, slot in param_slots(desc):
        put(slot, name)
    if desc.head == HEAD_BLEND:
        put(13, "w_linear")
    elif desc.head == HEAD_DYNAMIC:
        put(13, "sf_linear")
        put(14, "prob_linear")
    return tab


# ----------------------------------------------------------------------------- fused renderer
def make_view_set(vol_cl=None, imgs_cl=None, w2cs=None, intrinsics=None):
    """Pack the feature sources of one net into the C struct; keeps the tensors alive."""
    vs = ViewSet()
    keep = []
    if vol_cl is not None:
        vs.vol_cl, (vs.Hv, vs.Wv, vs.D) = _ptr(vol_cl), vol_cl.shape[:3]
        vs.imgs_cl, (vs.V, vs.H, vs.W) = _ptr(imgs_cl), imgs_cl.shape[:3]
        keep += [vol_cl, imgs_cl]
    if w2cs is not None:
        w2cs = _dev(w2cs, "w2cs")
        vs.w2cs = _ptr(w2cs)
        keep.append(w2cs)
    if intrinsics is not None:
        intrinsics = _dev(intrinsics, "intrinsics")
        vs.intrinsics = _ptr(intrinsics)
        keep.append(intrinsics)
    vs._keep = keep
    return vs


def set_fused_passes(shape):
    """None / "auto": pass shape chosen per launch; "dense" | "ranges": forced (tests, measurements)."""
    _check(lib().zest_render_fused_set_passes({None: 0, "auto": 0, "dense": 1, "ranges": 2}[shape]),
           "zest_render_fused_set_passes")


def fused_pass_shape(R, S, precision=PREC_BF16, cus=256):
    """-> (ray_ranges, n_wg, rounds) the fused renderer takes for this batch shape: ray_ranges 1 = a range of
    whole rays per workgroup (rays finished in the kernel), 0 = dense block shares + combine launch; n_wg =
    workgroups launched, rounds = passes of the busiest workgroup."""
    rr, n, rounds = _i(0), _i(0), _i(0)
    _check(lib().zest_render_fused_pass_shape(int(R), int(S), int(precision), int(cus), C.byref(rr), C.byref(n),
                                              C.byref(rounds)), "zest_render_fused_pass_shape")
    return rr.value, n.value, rounds.value


def render_fused(ndc, pts, z, rays_dir, desc_s, packed_s, views_s, desc_d=None, packed_d=None,
                 views_d=None, frame_idx=0.0, white_bkgd=False, out=None, workspace=None,
                 precision=PREC_BF16):
    """One launch for the inference path.  ndc, pts [R,S,3]; z [R,S]; rays_dir [R,3]; the packed
    weights must have been packed for `precision` (one of ENGINE_PRECISIONS).
    Returns out [R,16] (column layout: include/zest_render.h)."""
    if precision not in ENGINE_PRECISIONS:
        raise RuntimeError("zest_hip.render_fused: precision %r is not an engine operand type" % (precision,))
    z = _dev(z, "z", (None, None))
    R, S = z.shape
    ndc, pts, rays_dir = _dev(ndc, "ndc", (R, S, 3)), _dev(pts, "pts", (R, S, 3)), _dev(rays_dir, "rays_dir", (R, 3))
    for d_, pk, what in ((desc_s, packed_s, "static"), (desc_d, packed_d, "dynamic")):
        if d_ is not None and (pk is None or pk.numel() * pk.element_size() < mlp_packed_bytes(d_, precision)):
            raise RuntimeError("zest_hip.render_fused: the %s net's packed weights do not belong to its descriptor "
                               "and this precision (too small)" % what)
    if out is None:
        out = torch.empty(R, 16, device=z.device, dtype=torch.float32)
    elif tuple(out.shape) != (R, 16) or out.dtype != torch.float32 or not out.is_contiguous():
        raise RuntimeError("zest_hip.render_fused: out must be a contiguous fp32 [%d, 16] tensor" % R)
    need = int(lib().zest_render_fused_workspace(R, S))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=z.device, dtype=torch.uint8)
    _check(lib().zest_render_fused_fwd(
        _ptr(ndc), _ptr(pts), _ptr(z), _ptr(rays_dir), R, S, C.byref(desc_s), _ptr(packed_s),
        C.byref(views_s) if views_s is not None else None,
        C.byref(desc_d) if desc_d is not None else None, _ptr(packed_d),
        C.byref(views_d) if views_d is not None else None, float(frame_idx), int(precision),
        int(bool(white_bkgd)), _ptr(workspace), _ptr(out), _stream(z)), "zest_render_fused_fwd")
    return out
