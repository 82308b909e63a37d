"""Autograd wiring of the training path: each stage of `rendering` is a torch.autograd.Function
whose forward AND backward are HIP kernels behind the C ABI (plus rocBLAS sgemm inside the
MLP).  Gradients exist exactly where the reference's training graph has them:

  EncodeFn     d/d(volume coordinates)  - through the positional encoding and the trilinear
               lookup (scene-flow displaced points, renderer.py:461,488) - and d/d(encoding
               volume) (MVSNet trains through it; in the kernels' layout, VolumeCLFn converts once per step);
               world points, images, cameras, directions are data.  EncodePairFn: the two neighbour frames in one batch.
  MlpFn        d/d(input point-encoding and feature columns) and d/d(every parameter).
  CompositeFn, BlendFn   d/d(raw predictions[, blend weight]); depth samples are data.
  Prob2dFn     compute_2d_prob: the weights are detached in the reference (renderer.py:31).

  VolumeCostFn, HomoWarpFn   d/d(feature maps) of the MVS plane sweep (scatter-add through the bilinear taps).

Two training modes of the MLP: fp32 (MlpFn: rocBLAS sgemm + fused elementwise kernels, the 1e-3 parity
mode; `--precision 32`) and bf16 on the hand-written MFMA kernels (MlpFn16; `--precision 16`).  The other
stages compute in fp32 in both.
"""
import torch
from torch.autograd import Function

import zest_hip


class VolumeCLFn(Function):
    """The caller's encoding volume [1,8,D,H,W] -> the kernels' copy [H,W,D,8] (zest_utils' cached conversion) as ONE
    autograd node per volume and step: every lookup of the step (the dynamic volume is read by three to five
    passes) scatters into a channels-last gradient, autograd sums those, and the way back to the caller's layout is
    paid once instead of per pass."""

    @staticmethod
    def forward(ctx, volume, views):
        """views: renderer._Views (not a tensor: its cached channels-last copy is handed on as a new tensor object
        on the same memory, which nobody writes)."""
        ctx.shape = tuple(volume.shape)
        return views.vol_cl.detach()

    @staticmethod
    def backward(ctx, g_cl):
        return zest_hip.volume_from_cl(g_cl.contiguous()).view(ctx.shape), None


class EncodeFn(Function):
    @staticmethod
    def forward(ctx, ndc, vol_cl, views, pts, dirs, t):
        """ndc [R,S,3]; vol_cl: VolumeCLFn's output when the volume wants a gradient, else None (the lookup reads
        views.vol_cl either way); views: renderer._Views."""
        x = views.encode(ndc, pts, dirs, t)
        ctx.views, ctx.t = views, t
        ctx.save_for_backward(ndc)
        return x

    @staticmethod
    def backward(ctx, g_x):
        (ndc,) = ctx.saved_tensors
        v = ctx.views
        want_vol = v.vol_cl is not None and ctx.needs_input_grad[1]
        V = v.imgs_cl.shape[0] if v.imgs_cl is not None else 0
        g_ndc, g_vol_cl = zest_hip.encode_bwd(g_x.contiguous(), ndc, ctx.t, v.vol_cl, V, want_vol)
        return g_ndc, g_vol_cl, None, None, None, None


class EncodePairFn(Function):
    """Two encodes of the same rays at two frame indices (the neighbour frames t -+ 1 of the scene-flow chain,
    reference renderer.py:460-497) into the halves of ONE [2R,S,C] batch, so the dynamic MLP runs forward and
    backward once for both - one set of parameter gradients instead of two for autograd to add - and both halves
    scatter into one volume gradient."""

    @staticmethod
    def forward(ctx, ndc_a, ndc_b, vol_cl, views, pts, dirs, t_a, t_b):
        R, S = ndc_a.shape[:2]
        x = ndc_a.new_empty(2 * R, S, views.c_in(True))
        views.encode(ndc_a, pts, dirs, t_a, out=x[:R])
        views.encode(ndc_b, pts, dirs, t_b, out=x[R:])
        ctx.views, ctx.t = views, (t_a, t_b)
        ctx.save_for_backward(ndc_a, ndc_b)
        return x

    @staticmethod
    def backward(ctx, g_x):
        ndc_a, ndc_b = ctx.saved_tensors
        v, R = ctx.views, ndc_a.shape[0]
        want_vol = v.vol_cl is not None and ctx.needs_input_grad[2]
        V = v.imgs_cl.shape[0] if v.imgs_cl is not None else 0
        g_x = g_x.contiguous()
        g_a, g_vol_cl = zest_hip.encode_bwd(g_x[:R], ndc_a, ctx.t[0], v.vol_cl, V, want_vol)
        g_b, g_vol_cl = zest_hip.encode_bwd(g_x[R:], ndc_b, ctx.t[1], v.vol_cl, V, want_vol, g_vol=g_vol_cl)
        return g_a, g_b, g_vol_cl, None, None, None, None, None


class MlpFn(Function):
    @staticmethod
    def forward(ctx, x, desc, slots, *params):
        """params: the (weight, bias) tensors of the slots in `slots` (ZEST_P_* order)."""
        table = [None] * (2 * zest_hip.P_COUNT)
        for i, s in enumerate(slots):
            table[2 * s], table[2 * s + 1] = params[2 * i], params[2 * i + 1]
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        out, saved = zest_hip.mlp_train_fwd(desc, table, x2)
        ctx.desc, ctx.slots, ctx.lead = desc, slots, lead
        ctx.save_for_backward(x2, saved, out, *params)
        return out.view(*lead, desc.out_ch)

    @staticmethod
    def backward(ctx, g_out):
        x2, saved, out, *params = ctx.saved_tensors
        table = [None] * (2 * zest_hip.P_COUNT)
        for i, s in enumerate(ctx.slots):
            table[2 * s], table[2 * s + 1] = params[2 * i], params[2 * i + 1]
        g_x, grads = zest_hip.mlp_train_bwd(ctx.desc, table, x2, saved, out,
                                            g_out.reshape(-1, ctx.desc.out_ch).contiguous(),
                                            want_gx=ctx.needs_input_grad[0])
        gp = []
        for s in ctx.slots:
            gp += [grads[2 * s], grads[2 * s + 1]]
        return (g_x.view(*ctx.lead, -1) if g_x is not None else None, None, None, *gp)


class MlpFn16(Function):
    """bf16 training path: forward and backward entirely on the hand-written MFMA kernels
    (zest_mlp_train16_*): engine forward with activation stash; data / modulation / weight-gradient
    kernels.  fp32 parameters in, fp32 gradients out; 'v0' nets.  `shared`: a dict the calls of one net within one
    rendering() share - the packed forward and transposed backward weight streams are built once per step, not
    once per pass (the dynamic net runs two to three passes)."""

    @staticmethod
    def forward(ctx, x, desc, slots, shared, *params):
        table = [None] * (2 * zest_hip.P_COUNT)
        for i, s in enumerate(slots):
            table[2 * s], table[2 * s + 1] = params[2 * i].detach(), params[2 * i + 1].detach()
        lead = x.shape[:-1]
        x2 = x.detach().reshape(-1, x.shape[-1]).contiguous()
        if shared is None:
            shared = {}
        if "fwd" not in shared:
            shared["fwd"] = zest_hip.mlp_pack(desc, zest_hip.PREC_BF16, table)
        out, stash = zest_hip.mlp_train16_fwd(desc, shared["fwd"], x2)
        ctx.desc, ctx.slots, ctx.lead, ctx.shared = desc, slots, lead, shared
        ctx.save_for_backward(x2, stash, out, *params)
        return out.view(*lead, desc.out_ch)

    @staticmethod
    def backward(ctx, g_out):
        x2, stash, out, *params = ctx.saved_tensors
        table = [None] * (2 * zest_hip.P_COUNT)
        for i, s in enumerate(ctx.slots):
            table[2 * s], table[2 * s + 1] = params[2 * i].detach(), params[2 * i + 1].detach()
        if "bwd" not in ctx.shared:
            ctx.shared["bwd"] = zest_hip.mlp_train16_pack_bwd(ctx.desc, table)
        g_x, grads, ctx.shared["work"] = zest_hip.mlp_train16_bwd(
            ctx.desc, ctx.shared["bwd"], table, x2, stash, out, g_out.reshape(-1, ctx.desc.out_ch).contiguous(),
            work=ctx.shared.get("work"))
        gp = []
        for s in ctx.slots:
            gp += [grads[2 * s], grads[2 * s + 1]]
        return (g_x.view(*ctx.lead, -1) if ctx.needs_input_grad[0] else None, None, None, None, *gp)


class CompositeFn(Function):
    @staticmethod
    def forward(ctx, raw, z, dirs, noise, noise_std, white_bkgd, is_dists=False):
        sp = dict(rays_dir=None, dists=dirs) if is_dists else dict(rays_dir=dirs)
        rgb, disp, acc, w, depth, alpha = zest_hip.composite(raw, z, noise=noise, noise_std=noise_std,
                                                             white_bkgd=white_bkgd, **sp)
        ctx.cfg = (noise_std, white_bkgd, is_dists)
        ctx.save_for_backward(raw, z, dirs, noise if noise is not None else raw.new_empty(0))
        ctx.mark_non_differentiable(disp, alpha)
        return rgb, disp, acc, w, depth, alpha

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth, g_alpha):
        raw, z, dirs, noise = ctx.saved_tensors
        sp = dict(rays_dir=None, dists=dirs) if ctx.cfg[2] else dict(rays_dir=dirs)
        g_raw = zest_hip.composite_bwd(raw, z, sp["rays_dir"], noise if noise.numel() else None, ctx.cfg[0],
                                       ctx.cfg[1], g_rgb, g_depth, g_acc, g_w, dists=sp.get("dists"))
        return g_raw, None, None, None, None, None, None


class BlendFn(Function):
    @staticmethod
    def forward(ctx, raw_dy, raw_st, blend, z, dirs, noise, noise_std, is_dists=False):
        """dirs: rays_dir [R,3], or with is_dists the caller's own sample spacings [R,S]."""
        sp = dict(rays_dir=None, dists=dirs) if is_dists else dict(rays_dir=dirs)
        rgb, depth, rgb_fg, depth_fg, w_fg, w_dy, dd = zest_hip.composite_blend(raw_dy, raw_st, blend, z,
                                                                                noise=noise, noise_std=noise_std, **sp)
        ctx.noise_std, ctx.is_dists = noise_std, is_dists
        ctx.save_for_backward(raw_dy, raw_st, blend, z, dirs, noise if noise is not None else z.new_empty(0))
        ctx.mark_non_differentiable(dd)
        return rgb, depth, rgb_fg, depth_fg, w_fg, w_dy, dd

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_rgb_fg, g_depth_fg, g_wfg, g_wd, g_dd):
        raw_dy, raw_st, blend, z, dirs, noise = ctx.saved_tensors
        g_dy, g_st, g_b = zest_hip.composite_blend_bwd(raw_dy, raw_st, blend, z, None if ctx.is_dists else dirs,
                                                       noise if noise.numel() else None, ctx.noise_std, g_rgb,
                                                       g_depth, g_rgb_fg, g_depth_fg, g_wfg, g_wd,
                                                       dists=dirs if ctx.is_dists else None)
        return g_dy, g_st, g_b, None, None, None, None, None


class Prob2dFn(Function):
    @staticmethod
    def forward(ctx, weights, prob):
        ctx.save_for_backward(weights)
        return zest_hip.weighted_complement_sum(weights, prob)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        return None, -(g[:, None] * w)


class SplitLastFn(Function):
    """raw [..., C] -> its column groups (contiguous).  Backward is ONE concatenation of the groups' gradients, where
    autograd's own slice backward launches a zero fill, a strided copy and an accumulation per group (a dozen small
    kernels per network output in the ZeST training step)."""

    @staticmethod
    def forward(ctx, raw, *sizes):
        ctx.sizes = sizes
        ctx.set_materialize_grads(False)
        return tuple(p.contiguous() for p in raw.split(sizes, -1))

    @staticmethod
    def backward(ctx, *grads):
        ref = next(g for g in grads if g is not None)
        parts = [g if g is not None else ref.new_zeros(*ref.shape[:-1], n) for g, n in zip(grads, ctx.sizes)]
        return (torch.cat(parts, -1),) + (None,) * len(ctx.sizes)


class SplitRowsFn(Function):
    """x [n R, ...] -> n blocks of R rows; backward is one concatenation (autograd's own slice backward fills and
    copies a full-size tensor per block)."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.chunk(n, 0))

    @staticmethod
    def backward(ctx, *grads):
        return torch.cat(grads, 0), None


def split_last(raw, sizes):
    """Column groups of raw's last dimension: under autograd through SplitLastFn, otherwise plain views."""
    if torch.is_grad_enabled() and raw.requires_grad:
        return SplitLastFn.apply(raw, *sizes)
    return raw.split(sizes, -1)


def mlp_apply(net, x, time_codes=None, bf16=False, shared=None):
    """Training forward of a zest networks.MVSNeRF on x [..., C_in] with autograd.  time_codes: the
    frame's latent code for a net with time-code channels (folded into layer 0 / 5 biases with
    differentiable torch ops: gradients reach the code and the full-width weights).  bf16: the fast
    training mode (MlpFn16, 'v0' nets); otherwise the fp32 parity path (MlpFn).  shared: see MlpFn16."""
    mod = net.nerf
    desc = mod._desc()
    named = mod.effective_parameters(time_codes)
    slots, params = [], []
    for name, slot in zest_hip.param_slots(desc):
        slots.append(slot)
        params += [named[name + ".weight"], named[name + ".bias"]]
    extra = {zest_hip.HEAD_BLEND: [("w_linear", 13)],
             zest_hip.HEAD_DYNAMIC: [("sf_linear", 13), ("prob_linear", 14)]}.get(desc.head, [])
    for name, slot in extra:
        slots.append(slot)
        params += [named[name + ".weight"], named[name + ".bias"]]
    if bf16 and desc.net_type == 0 and desc.is_default_shape:      # the MFMA training kernels' shape
        return MlpFn16.apply(x, desc, tuple(slots), shared, *params)
    if bf16:
        import warnings
        warnings.warn("zest: --precision 16 training of a %s net takes the fp32 training path (rocBLAS sgemm): the "
                      "bf16 MFMA training kernels cover 'v0' nets of depth 8 / width 256 / skips [4]"
                      % ("'v2'" if desc.net_type else "D=%d W=%d" % (desc.D, desc.W)), stacklevel=2)
    return MlpFn.apply(x, desc, tuple(slots), *params)


class VolumeCostFn(Function):
    """MVSNet.build_volume_cost: plane sweep forward and backward in HIP.  The gradient goes to the
    feature maps (FeatureNet); images, homographies and depths are data (reference networks.py:1077-1140
    under autograd gives the same: the masks are comparisons, the grid depends on cameras only)."""

    @staticmethod
    def forward(ctx, feats, imgs_lr, proj, depth, pad):
        img_feat, masks, fcl = zest_hip.volume_cost(feats, imgs_lr, proj, depth, pad, return_feats_cl=True)
        ctx.pad = pad
        ctx.save_for_backward(fcl, proj, depth)
        ctx.mark_non_differentiable(masks)
        return img_feat, masks

    @staticmethod
    def backward(ctx, g_img_feat, g_masks):
        fcl, proj, depth = ctx.saved_tensors
        return zest_hip.volume_cost_bwd(fcl, proj, depth, ctx.pad, g_img_feat.contiguous()), None, None, None, None


class HomoWarpFn(Function):
    """utils.homo_warp with respect to the source map (the sampling positions are data)."""

    @staticmethod
    def forward(ctx, src, proj, depth, grid, pad):
        warped, grid_out = zest_hip.homo_warp(src, proj, depth, grid, pad)
        ctx.pad, ctx.shape = pad, tuple(src.shape)
        ctx.save_for_backward(grid_out)
        ctx.mark_non_differentiable(grid_out)
        return warped, grid_out

    @staticmethod
    def backward(ctx, g_warped, g_grid):
        (grid,) = ctx.saved_tensors
        return zest_hip.homo_warp_bwd(g_warped.contiguous(), ctx.shape, grid=grid, pad=ctx.pad), None, None, None, None


class DistortionFn(Function):
    """distortion_loss: per-ray O(S^2) pair sum and its weight gradient from one HIP launch."""

    @staticmethod
    def forward(ctx, weights, t_vals):
        need = weights.requires_grad
        loss_ray, grad = zest_hip.distortion(weights, t_vals, want_grad=need)
        if need:
            ctx.save_for_backward(grad)
        return loss_ray.sum()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


class ProjectRaysFn(Function):
    """projection_from_ndc: expected point -> Euclidean -> camera -> pixels, fused per ray."""

    @staticmethod
    def forward(ctx, weights, pts, w2c, H, W, focal):
        ctx.cfg = (H, W, focal)
        ctx.save_for_backward(weights, pts, w2c)
        return zest_hip.project_rays(weights, pts, w2c, H, W, focal)

    @staticmethod
    def backward(ctx, g):
        weights, pts, w2c = ctx.saved_tensors
        H, W, focal = ctx.cfg
        dw, dp = zest_hip.project_rays_bwd(weights, pts, w2c, H, W, focal, g.contiguous(),
                                           ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dw, dp, None, None, None, None


# ------------------------------------------------------------------------------ volume builder: regularisation net
class CostRegFn(Function):
    """CostRegNet under autograd with its FORWARD on the HIP kernels (csrc/costreg.hip, 1 ms instead of the library's
    8-11 ms): the raw output of every layer, its norm constants and batch moments are kept, and the backward pass
    walks the U-Net in reverse with the library's own backward operators on them (aten.convolution_backward,
    aten.native_batch_norm_backward; the leaky ReLU and the skip additions are a few elementwise operations).
    Inputs after `passes`: the ten convolution weights, then (weight, bias) of the ten norms, in layer order."""

    @staticmethod
    def forward(ctx, cost, net, passes, *params):
        cl = torch.nn.functional.pad(cost[0].permute(1, 2, 3, 0), (0, zest_hip.COST_CL_CHANNELS - cost.shape[1])).contiguous()
        vol, raw, pr, mo = net.forward_hip(cl, passes, keep=True)
        ctx.net, ctx.passes, ctx.n = net, passes, len(raw)
        ctx.save_for_backward(cost, *raw, *pr, *mo, *params)
        return vol

    @staticmethod
    def backward(ctx, g_out):
        t = ctx.saved_tensors
        n = ctx.n
        cost, raw, pr, mo, params = t[0], t[1:1 + n], t[1 + n:1 + 2 * n], t[1 + 2 * n:1 + 3 * n], t[1 + 3 * n:]
        W, gam = params[:n], params[n::2]
        net = ctx.net
        bns = [getattr(net, nm).bn for nm, _ in net._HIP_CONVS] + [getattr(net, nm)[1] for nm in net._HIP_UPS]
        low = torch.bfloat16 if ctx.passes == 1 else None          # the --precision 16 path: library backward in bf16, as under AMP
        cf = lambda x: x.permute(3, 0, 1, 2)[None]                  # [D,H,W,C] -> [1,C,D,H,W] (channels-last memory)
        vec = lambda v: v.view(1, -1, 1, 1, 1)

        def act(i):                                                 # the activation a layer's consumers read
            return torch.nn.functional.leaky_relu(cf(raw[i]) * vec(pr[i][0]) + vec(pr[i][1]), 0.01)

        def norm_bwd(i, g_a):
            """Through leaky ReLU and the training-mode batch norm of layer i: one HIP call on the channels-last
            tensors (zest_costreg_bn_bwd; the library's norm backward has no channels-last kernel - its generic
            fallback took 7 ms of a 40 ms step, the same arithmetic in torch operators as much)."""
            g_cl = g_a[0].permute(1, 2, 3, 0).contiguous()          # a view when the gradient is channels-last already
            g_r, g_w, g_b = zest_hip.costreg_bn_bwd(raw[i], g_cl, pr[i], mo[i], gam[i])
            return cf(g_r), g_w, g_b

        def conv_bwd(g_r, x, w, stride, transposed, want_x=True):
            if low is not None:
                g_r, x, w = g_r.to(low), x.to(low), w.to(low)
            g_x, g_w, _ = torch.ops.aten.convolution_backward(g_r, x, w, None, [stride] * 3, [1] * 3, [1] * 3, transposed,
                                                              [1 if transposed else 0] * 3, 1, [want_x, True, False])
            return (g_x.float() if want_x else None), g_w.float()
        gW, gG, gB = [None] * n, [None] * n, [None] * n
        g_out = g_out.contiguous()
        # up path (layers 9, 8, 7 = conv11, conv9, conv7), skip additions on the way
        g_r, gG[9], gB[9] = norm_bwd(9, g_out)
        g_x2, gW[9] = conv_bwd(g_r, act(2) + act(8), W[9], 2, True)
        g_r, gG[8], gB[8] = norm_bwd(8, g_x2)
        g_x1, gW[8] = conv_bwd(g_r, act(4) + act(7), W[8], 2, True)
        g_r, gG[7], gB[7] = norm_bwd(7, g_x1)
        g_a, gW[7] = conv_bwd(g_r, act(6), W[7], 2, True)
        # down path in reverse (layers 6 .. 0)
        skip = {4: g_x1, 2: g_x2, 0: g_out}
        strides = [s for _, s in net._HIP_CONVS]
        for i in range(6, -1, -1):
            if i in skip:
                g_a = g_a + skip[i]
            g_r, gG[i], gB[i] = norm_bwd(i, g_a)
            x = act(i - 1) if i > 0 else cost
            g_a, gW[i] = conv_bwd(g_r, x, W[i], strides[i], False, want_x=(i > 0 or ctx.needs_input_grad[0]))
        grads = list(gW)
        for i in range(n):
            grads += [gG[i], gB[i]]
        return (g_a, None, None) + tuple(grads)


def costreg_apply(net, cost_vol, passes):
    """CostRegNet(cost_vol [1,41,D,H,W]) -> [1,8,D,H,W] through CostRegFn."""
    convs = [getattr(net, nm).conv for nm, _ in net._HIP_CONVS] + [getattr(net, nm)[0] for nm in net._HIP_UPS]
    bns = [getattr(net, nm).bn for nm, _ in net._HIP_CONVS] + [getattr(net, nm)[1] for nm in net._HIP_UPS]
    params = [c.weight for c in convs]
    for b in bns:
        params += [b.weight, b.bias]
    return CostRegFn.apply(cost_vol.float(), net, passes, *params)


class FeatureFn(Function):
    """FeatureNet under autograd with its forward on the HIP kernels (csrc/costreg.hip, zest_conv2d_fwd), as CostRegFn:
    norm + activation backward in HIP (zest_costreg_bn_bwd), convolution backward through aten.convolution_backward on
    the kept raw outputs, the 1x1 top layer as matrix products.  Inputs after `passes`: the eight convolution weights,
    the top layer's weight and bias, then (weight, bias) of the eight norms."""

    @staticmethod
    def forward(ctx, imgs, net, passes, *params):
        feats, raw, pr, mo = net.forward_hip(imgs, passes, keep=True)
        ctx.net, ctx.passes, ctx.n = net, passes, len(raw)
        ctx.save_for_backward(imgs, *raw, *pr, *mo, *params)
        return feats.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g_feats):
        t = ctx.saved_tensors
        n = ctx.n
        imgs, raw, pr, mo, params = t[0], t[1:1 + n], t[1 + n:1 + 2 * n], t[1 + 2 * n:1 + 3 * n], t[1 + 3 * n:]
        W, top_w, gam = params[:n], params[n], params[n + 2::2]
        convs = [getattr(ctx.net, s_)[i] for s_, i in ctx.net._HIP_LAYERS]
        low = torch.bfloat16 if ctx.passes == 1 else None
        cf = lambda x: x.permute(0, 3, 1, 2)                       # [N,H,W,C] -> [N,C,H,W] (channels-last memory)
        act = lambda i: torch.nn.functional.leaky_relu(raw[i] * pr[i][0] + pr[i][1], 0.01)
        g2 = g_feats.permute(0, 2, 3, 1).reshape(-1, 32).float()
        a7 = act(n - 1).view(-1, 32)
        g_top_w, g_top_b = (g2.t() @ a7).view_as(top_w), g2.sum(0)
        g_a = (g2 @ top_w.view(32, 32).float()).view(raw[n - 1].shape)      # channels-last gradient of the last activation
        gW, gG, gB = [None] * n, [None] * n, [None] * n
        for i in range(n - 1, -1, -1):
            g_r, gG[i], gB[i] = zest_hip.costreg_bn_bwd(raw[i], g_a.contiguous(), pr[i], mo[i], gam[i])
            x = cf(act(i - 1)) if i > 0 else imgs
            k, s_ = convs[i].conv.kernel_size[0], convs[i].conv.stride[0]
            go, xi, w = cf(g_r), x, W[i]
            if low is not None:
                go, xi, w = go.to(low), xi.to(low), w.to(low)
            g_x, g_w, _ = torch.ops.aten.convolution_backward(go, xi, w, None, [s_] * 2, [k // 2] * 2, [1, 1], False, [0, 0], 1,
                                                              [i > 0, True, False])
            gW[i] = g_w.float()
            if i > 0:
                g_a = g_x.float().permute(0, 2, 3, 1)
        grads = list(gW) + [g_top_w, g_top_b]
        for i in range(n):
            grads += [gG[i], gB[i]]
        return (None, None, None) + tuple(grads)


def feature_apply(net, imgs, passes):
    """FeatureNet(imgs [N,3,H,W]) -> [N,32,H/4,W/4] through FeatureFn."""
    convs = [getattr(net, s_)[i] for s_, i in net._HIP_LAYERS]
    params = [m.conv.weight for m in convs] + [net.toplayer.weight, net.toplayer.bias]
    for m in convs:
        params += [m.bn.weight, m.bn.bias]
    return FeatureFn.apply(imgs.float(), net, passes, *params)
