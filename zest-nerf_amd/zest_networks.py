"""Drop-in `networks` module for the rendering hot path (MI355X / HIP).

Exports the classes the reference's train.py and renderer bind by name
(/root/reference/train.py:36-37, networks.py:25-26): `Embedding`, `Renderer`,
`Renderer_linear`, `MVSNeRF`.  Constructors, attribute names and state-dict keys/shapes
match the reference (networks.py:29-353) so its checkpoints load with
`load_state_dict`; `forward` runs the hand-written gfx950 kernels through the C ABI
(include/zest_render.h).  There is no PyTorch implementation of the forward pass here:
without libzest_hip.so, or on CPU tensors, calls raise.

Precision (resolve_precision): the reference's `--precision 32` (opt.py:69, the default) runs the
MLP on exact-product fp32 MFMA per operator and, in the fused single-launch renderer, on
split-fp16 operand pairs (fp32-class results at the fp16 matrix rate); `--precision 16` selects
16-bit operands with fp32 accumulation - bf16 by default, fp16 with `args.zest_dtype16 = "f16"`.
ZEST_PRECISION = fp32 | f16x3 | bf16 | f16 overrides both.
"""
import os

import weakref

import torch
import torch.nn as nn

import zest_hip

__all__ = ["Embedding", "Renderer", "Renderer_linear", "MVSNeRF", "resolve_precision", "inference_precision",
           "ActivatedBatchNorm", "ConvBnReLU", "ConvBnReLU3D", "FeatureNet", "CostRegNet", "MVSNet",
           "MVSNeRF_G", "DyMVSNeRF_G"]


_PREC_BY_NAME = {"bf16": zest_hip.PREC_BF16, "16": zest_hip.PREC_BF16,
                 "f16": zest_hip.PREC_F16, "fp16": zest_hip.PREC_F16, "half": zest_hip.PREC_F16,
                 "f16x3": zest_hip.PREC_F16X3, "fp16x3": zest_hip.PREC_F16X3, "x3": zest_hip.PREC_F16X3,
                 "fp32": zest_hip.PREC_F32, "f32": zest_hip.PREC_F32, "32": zest_hip.PREC_F32}


def resolve_precision(args=None):
    """-> zest_hip.PREC_* from ZEST_PRECISION, else args.precision (16 | 32) and, for 16,
    args.zest_dtype16 ("bf16" default | "f16")."""
    env = os.environ.get("ZEST_PRECISION", "").lower()
    if env:
        if env not in _PREC_BY_NAME:
            raise ValueError("ZEST_PRECISION=%r: expected one of %s" % (env, sorted(_PREC_BY_NAME)))
        return _PREC_BY_NAME[env]
    if args is not None and int(getattr(args, "precision", 32) or 32) == 16:
        name = str(getattr(args, "zest_dtype16", "bf16")).lower()
        if name not in ("bf16", "f16", "fp16"):
            raise ValueError("args.zest_dtype16=%r: expected 'bf16' or 'f16'" % (name,))
        return _PREC_BY_NAME[name]
    return zest_hip.PREC_F32


def inference_precision(prec, args=None):
    """Operand type of an inference (no-grad) MLP launch in mode `prec`: fp32 mode runs on split-fp16
    pairs (ZEST_PREC_F16X3: same 1e-4 / 1e-3 agreement with the fp32 reference as the exact-product
    kernel at ~7x its speed) unless exact fp32 products are asked for with `args.zest_fp32_exact`
    or ZEST_FP32_EXACT=1 (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, no fp16 range limit)."""
    if prec != zest_hip.PREC_F32:
        return prec
    if os.environ.get("ZEST_FP32_EXACT", "") not in ("", "0") or getattr(args, "zest_fp32_exact", False):
        return zest_hip.PREC_F32
    return zest_hip.PREC_F16X3


class Embedding(nn.Module):
    """Positional encoder x -> (x, sin(2^k x), cos(2^k x))_k  (reference networks.py:29-65)."""

    def __init__(self, in_channels, N_freqs, logscale=True):
        super().__init__()
        if not logscale:
            raise NotImplementedError("zest Embedding: only log-scale bands (the reference default) "
                                      "are implemented in the HIP encoder")
        self.N_freqs = N_freqs
        self.in_channels = in_channels
        self.funcs = [torch.sin, torch.cos]
        self.out_channels = in_channels * (len(self.funcs) * N_freqs + 1)
        self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs)

    def forward(self, x):
        if x.shape[-1] != self.in_channels:
            raise RuntimeError("Embedding expects %d channels, got %d" % (self.in_channels, x.shape[-1]))
        return zest_hip.embed(x, self.N_freqs)


class _MlpBase(nn.Module):
    """Parameters of the width-W MLP + the packed-weight cache for the MFMA kernels."""

    def _build(self, D, W, input_ch, input_ch_views, input_ch_feat, skips, use_viewdirs):
        # Reference networks.py:93-100 builds any D / W / skips.  The shipped shape (8 / 256 / [4]: every config
        # and opt.py:54-57's defaults) runs on the MFMA engine in every precision; other shapes run on the
        # exact-product fp32 kernel and the fp32 training path (engine_precision below).
        skips = [int(i) for i in skips]
        if not use_viewdirs:
            raise NotImplementedError("zest MLP: use_viewdirs=False (never built by the reference: networks.py:336-345 "
                                      "passes use_viewdirs=True)")
        if not 2 <= D <= 8 or W not in (64, 128, 192, 256) or any(i < 0 or i > D - 2 for i in skips):
            raise NotImplementedError(
                "zest MLP kernels cover 2 <= D <= 8, W in {64,128,192,256}, skips within 0 .. D-2 "
                "(got D=%s W=%s skips=%s)" % (D, W, skips))
        self.D, self.W, self.skips, self.use_viewdirs = D, W, skips, use_viewdirs
        self.default_shape = D == 8 and W == 256 and sorted(set(skips)) == [4]
        self.in_ch_pts, self.in_ch_views, self.in_ch_feat = input_ch, input_ch_views, input_ch_feat
        # Encoded point = 63 (xyz) or 84 (xyzt) channels; anything beyond 63 that is not 84 is the
        # Neural3D time code (reference train.py:112-113: input_ch += time_code_dim, static net).  The
        # code is one vector per rendering call, the same for every sample, so its columns of layers 0
        # and 5 fold into those layers' biases (effective_parameters) and the kernels see a 63-channel net.
        self.in_ch_pe = input_ch if input_ch in (63, 84) else 63
        self.in_ch_time = input_ch - self.in_ch_pe
        if self.in_ch_time < 0:
            raise NotImplementedError("zest MLP: input_ch=%d is smaller than the 63-channel point encoding" % input_ch)
        self.pts_linears = nn.ModuleList()
        for i in range(D - 1):
            if i == 0:
                self.pts_linears.append(nn.Linear(input_ch, W))
            self.pts_linears.append(nn.Linear(W + input_ch if i in skips else W, W))
        self.pts_bias = nn.Linear(input_ch_feat, W)
        self.views_linears = nn.ModuleList([nn.Linear(W + input_ch_views, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        self._packed = {}

    def _desc(self):
        raise NotImplementedError

    def _mk_desc(self, in_ch_pts, use_feat, net_type, head):
        if self.default_shape:          # zeros = the default shape: one plan / packed-weight cache entry
            return zest_hip.MlpDesc(in_ch_pts, self.in_ch_feat, self.in_ch_views, use_feat, net_type, head)
        return zest_hip.MlpDesc(in_ch_pts, self.in_ch_feat, self.in_ch_views, use_feat, net_type, head,
                                self.D, self.W, sum(1 << i for i in set(self.skips)))

    def engine_precision(self, prec):
        """Operand type this net can run `prec` in: the MFMA engine (bf16 / fp16 / split fp16) is
        written for the shipped shape; other depths / widths / skips run exact fp32 products."""
        return prec if self.default_shape else zest_hip.PREC_F32

    def effective_parameters(self, time_codes=None):
        """{name: tensor} of the 63/84-channel net the kernels run.  Without time-code channels these
        are the parameters themselves.  With them (in_ch_time = T > 0) `time_codes` ([T] or [1,T], the
        frame's latent code BEFORE the sigmoid, reference renderer.py:269-273) is folded in:
            layer 0:  W0 [256, P+T]      -> W0[:, :P],               b0 + W0[:, P:] sigmoid(tc)
            layer 5:  W5 [256, P+T+256]  -> W5[:, :P] | W5[:, P+T:],  b5 + W5[:, P:P+T] sigmoid(tc)
        (layer i+1 for each i in skips; the skip layer's input is [point | time code | h], networks.py:182).  Plain torch ops: under
        autograd the gradients reach the original parameters and the code."""
        params = self.__dict__.get("_zest_params")
        if params is None:
            params = self.__dict__["_zest_params"] = tuple(self.named_parameters())
        named = dict(params)
        T, P = self.in_ch_time, self.in_ch_pe
        if T == 0:
            if time_codes is not None:
                raise RuntimeError("zest MLP: time_codes given to a net without time-code channels (input_ch=%d)" % P)
            return named
        if time_codes is None:
            raise RuntimeError("zest MLP: this net has %d time-code input channels; pass time_codes to rendering()" % T)
        tc = torch.sigmoid(time_codes.float()).reshape(-1)
        if tc.numel() != T:
            raise RuntimeError("zest MLP: time code has %d values, the net expects %d" % (tc.numel(), T))
        w0 = named["pts_linears.0.weight"]
        named["pts_linears.0.weight"] = w0[:, :P].contiguous()
        named["pts_linears.0.bias"] = named["pts_linears.0.bias"] + w0[:, P:] @ tc
        for i in set(self.skips):
            k = "pts_linears.%d" % (i + 1)
            w = named[k + ".weight"]
            named[k + ".weight"] = torch.cat([w[:, :P], w[:, P + T:]], 1)
            named[k + ".bias"] = named[k + ".bias"] + w[:, P:P + T] @ tc
        return named

    def packed(self, precision, desc=None, time_codes=None):
        """Weights in MFMA stream order; re-packed when any parameter (or the time code) changes
        (storage or version counter: optimizer steps, load_state_dict and .to() all show up).  The
        (name, Parameter) list is cached - walking the module tree costs more host time than a
        1024-ray launch.  desc: a variant descriptor of the same parameters (forward_alpha), packed and
        cached apart."""
        params = self.__dict__.get("_zest_params")
        if params is None:
            params = self.__dict__["_zest_params"] = tuple(self.named_parameters())
        stamp = tuple((p.data_ptr(), p._version) for _, p in params)
        key = precision if desc is None else (precision, desc.use_feat, desc.net_type, desc.head)
        hit = self._packed.get(key)
        fresh = hit is not None and hit[0] == stamp
        if fresh and time_codes is not None:
            # The frame's code is folded into the packed biases, so it is part of what the entry is valid for.
            # Its address and version counter do not identify it: the reference's callers build
            # `self.time_codes[keyframe_id].to(device)` anew for every batch (train.py:849, 1058) and the caching
            # allocator hands the block just freed to the next frame's code at version 0.  Same live tensor object
            # at the same version = unchanged; any other tensor is compared by value with the copy the entry
            # keeps (T floats: one small device compare per call for such callers).
            ref, ver, copy = hit[2]
            if not (ref() is time_codes and ver == time_codes._version):
                fresh = (copy.shape == time_codes.shape and copy.device == time_codes.device
                         and bool(torch.equal(copy, time_codes.detach())))
                if fresh:
                    hit = (hit[0], hit[1], (weakref.ref(time_codes), time_codes._version, copy))
                    self._packed[key] = hit
        if not fresh:
            desc = self._desc() if desc is None else desc
            with torch.no_grad():
                eff = self.effective_parameters(time_codes)
            state = {"nerf." + k: v.detach() for k, v in eff.items()}
            tc = None if time_codes is None else (weakref.ref(time_codes), time_codes._version,
                                                  time_codes.detach().clone())
            hit = (stamp, zest_hip.mlp_pack(desc, precision, zest_hip.param_table(state, desc)), tc)
            self._packed[key] = hit
        return hit[1]

    def _forward_alpha(self, x, net_type, relu):
        """Density-only pass of the reference (networks.py:134-147, 266-280): x = [points | features]
        (no direction columns), the trunk ALWAYS modulated by pts_bias(features).  Runs the MLP kernel
        with zero direction columns and the alpha head, and keeps the density column."""
        if self.in_ch_time:
            raise NotImplementedError("forward_alpha: not available for nets with time-code channels")
        if x.shape[-1] != self.in_ch_pts + self.in_ch_feat:
            raise RuntimeError("forward_alpha expects %d + %d input channels, got %d"
                               % (self.in_ch_pts, self.in_ch_feat, x.shape[-1]))
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("forward_alpha: inference only (the reference never calls it: "
                                      "renderer.py:295 sets alpha_only only when no direction is given)")
        desc = self._mk_desc(self.in_ch_pts, 1, net_type, zest_hip.HEAD_NONE)
        prec = self.engine_precision(inference_precision(resolve_precision()))
        xin = torch.cat([x, x.new_zeros(*x.shape[:-1], self.in_ch_views)], -1)
        alpha = zest_hip.mlp_fwd(desc, prec, self.packed(prec, desc), xin)[..., 3:4]
        return torch.relu(alpha) if relu else alpha

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Parameter, nn.Module)):
            self.__dict__.pop("_zest_params", None)          # a replaced parameter / sub-module: rebuild the list
        super().__setattr__(name, value)

    def zest_forward(self, x, precision=None):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("zest MLP: use MVSNeRF.forward (it owns the `nerf.` parameter names the "
                               "training path needs)")
        if self.in_ch_time:
            raise NotImplementedError("zest MLP: a net with time-code channels runs through rendering(..., time_codes=...), "
                                      "which folds the frame's code into the biases of layers 0 and 5")
        desc = self._desc()
        prec = self.engine_precision(inference_precision(resolve_precision()) if precision is None else precision)
        return zest_hip.mlp_fwd(desc, prec, self.packed(prec), x)

    def forward(self, x):
        return self.zest_forward(x)


class Renderer(_MlpBase):
    """'v0' net: trunk layers relu(Linear(h) * pts_bias(feats)) (reference networks.py:73-221)."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, input_ch_feat=8,
                 skips=[4], use_viewdirs=False, sceneflow=False, static=True, use_mvs=False):
        super().__init__()
        self._build(D, W, input_ch, input_ch_views, input_ch_feat, skips, use_viewdirs)
        self.predict_sceneflow, self.static, self.use_mvs = sceneflow, static, use_mvs
        if sceneflow:
            if static:
                self.w_linear = nn.Linear(W, 1)
            else:
                self.sf_linear = nn.Linear(W, 6)
                self.prob_linear = nn.Linear(W, 2)

    def _desc(self):
        head = zest_hip.HEAD_NONE
        if self.predict_sceneflow:
            head = zest_hip.HEAD_BLEND if self.static else zest_hip.HEAD_DYNAMIC
        return self._mk_desc(self.in_ch_pe, int(bool(self.use_mvs)), 0, head)

    def forward_alpha(self, x):
        """relu(alpha_linear(trunk(x))) with multiplicative modulation (reference networks.py:134-147)."""
        return self._forward_alpha(x, 0, True)


class Renderer_linear(_MlpBase):
    """'v2' net: additive modulation, relu(alpha), sigmoid(rgb) (reference networks.py:223-319)."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, input_ch_feat=8,
                 skips=[4], use_viewdirs=False):
        super().__init__()
        self._build(D, W, input_ch, input_ch_views, input_ch_feat, skips, use_viewdirs)

    def _desc(self):
        return self._mk_desc(self.in_ch_pe, 1, 2, zest_hip.HEAD_NONE)

    def forward_alpha(self, x):
        """alpha_linear(trunk(x)), additive modulation, no activation (reference networks.py:266-280)."""
        return self._forward_alpha(x, 3, False)


class MVSNeRF(nn.Module):
    """Selects the net type; same constructor and `nerf.*` state-dict keys as the reference
    (networks.py:321-353)."""

    def __init__(self, D=8, W=256, input_ch_pts=3, output_ch=4, input_ch_views=3, input_ch_feat=8,
                 skips=[4], net_type='v2', sceneflow=False, static=True, use_mvs=False):
        super().__init__()
        self.in_ch_pts, self.out_ch_pts = input_ch_pts, output_ch
        self.in_ch_views, self.in_ch_feat = input_ch_views, input_ch_feat
        self.net_type = net_type
        if net_type == 'v0':
            self.nerf = Renderer(D=D, W=W, input_ch_feat=input_ch_feat, input_ch=input_ch_pts,
                                 output_ch=output_ch, skips=skips, input_ch_views=input_ch_views,
                                 use_viewdirs=True, sceneflow=sceneflow, static=static, use_mvs=use_mvs)
        elif net_type == 'v2':
            self.nerf = Renderer_linear(D=D, W=W, input_ch_feat=input_ch_feat, input_ch=input_ch_pts,
                                        output_ch=output_ch, skips=skips,
                                        input_ch_views=input_ch_views, use_viewdirs=True)
        else:
            raise ValueError("net_type must be 'v0' or 'v2', got %r" % (net_type,))

    def desc(self):
        return self.nerf._desc()

    def packed(self, precision, time_codes=None):
        return self.nerf.packed(precision, time_codes=time_codes)

    def forward_alpha(self, x):
        return self.nerf.forward_alpha(x)

    def zest_forward(self, x, precision=None):
        return self.nerf.zest_forward(x, precision)

    def forward(self, x):
        if self.nerf.in_ch_time == 0 and torch.is_grad_enabled() and (
                x.requires_grad or any(p.requires_grad for p in self.parameters())):
            import zest_autograd
            return zest_autograd.mlp_apply(self, x)          # fp32 training path (HIP fwd + bwd)
        return self.nerf(x)


# ------------------------------------------------------------------------------------------
# MVS volume builder (reference networks.py:935-1238).  The convolutional
# stacks are plain library convolutions (MIOpen through torch.nn: plumbing, not the product);
# the plane sweep between them - homo_warp + build_volume_cost, the part that streams
# hundreds of MB - is the HIP kernel zest_volume_cost_fwd.  The reference normalises with
# InPlaceABN (inplace_abn, a CUDA-only extension that is not installable here): its forward is
# batch norm followed by leaky ReLU(0.01), and its parameters are weight / bias / running_mean /
# running_var, so ActivatedBatchNorm keeps the reference's state-dict keys and checkpoints load.
# Parity: the sampling half of homo_warp is pinned by a reference-generated fixture
# (tests/golden/homo_warp.npz); the grid construction and build_volume_cost are checked against the
# oracle's restatement of the source text, and the convolutional stacks cannot be run in the
# reference without inplace_abn: "parity unpinned" for those.
class ActivatedBatchNorm(nn.BatchNorm2d):
    """InPlaceABN substitute: BatchNorm (any spatial rank) + leaky ReLU(0.01)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu",
                 activation_param=0.01):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        if activation not in ("leaky_relu", "identity"):
            raise NotImplementedError("ActivatedBatchNorm: activation %r" % (activation,))
        self.activation, self.activation_param = activation, activation_param

    def _check_input_dim(self, input):
        if input.dim() < 3:
            raise ValueError("expected at least 3D input (got %dD)" % input.dim())

    def forward(self, x):
        y = super().forward(x)
        return torch.nn.functional.leaky_relu(y, self.activation_param) if self.activation == "leaky_relu" else y


def pack_conv_weights(w, passes):
    """Conv3d weight [cout,cin,KD,K,K] (or Conv2d [cout,cin,K,K]: KD = 1) -> MFMA A operands of csrc/costreg.hip,
    [dz][dy][chunk][16-row tile][hi, lo][lane] x 8 bf16: lane (m = l & 15, g = l >> 4) of chunk c holds output channel
    16 nt + m against octet o = 4 c + g of the x-window (pixel o // (cin/8), channels 8 (o % (cin/8)) ..+7), zero beyond
    the window / the channels (cin is padded to a multiple of 8)."""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    cout, cin, KD, K, _ = w.shape
    opt = (cin + 7) // 8
    cpr, nt, dev = (K * opt + 3) // 4, (cout + 15) // 16, w.device
    lane, e = torch.arange(64, device=dev), torch.arange(8, device=dev)
    o = 4 * torch.arange(cpr, device=dev).view(cpr, 1, 1, 1) + (lane >> 4).view(1, 1, 64, 1)
    px, ci = o // opt, 8 * (o % opt) + e.view(1, 1, 1, 8)
    co = 16 * torch.arange(nt, device=dev).view(1, nt, 1, 1) + (lane & 15).view(1, 1, 64, 1)
    valid = ((o < K * opt) & (ci < cin) & (co < cout)).expand(cpr, nt, 64, 8)
    full = (cpr, nt, 64, 8)
    vals = w.detach().float()[co.clamp(max=cout - 1).expand(full), ci.clamp(max=cin - 1).expand(full), :, :,
                              px.clamp(max=K - 1).expand(full)]                     # [cpr,nt,64,8,dz,dy]
    vals = (vals * valid[..., None, None]).permute(4, 5, 0, 1, 2, 3)
    hi = vals.to(torch.bfloat16)
    parts = [hi] if passes == 1 else [hi, (vals - hi.float()).to(torch.bfloat16)]
    return torch.stack(parts, 4).contiguous().view(torch.int16).view(-1)           # [KD,K,cpr,nt,parts,64,8]


def _cached_packs(module, passes, weights, build):
    """Packed weights of `module`, rebuilt when a weight tensor is replaced or modified in place."""
    key = (passes,) + tuple((w._version, str(w.device)) for w in weights)
    cache = module.__dict__.get("_zest_packs")
    # the tensors themselves are compared through weak references (an id() or an address can be recycled)
    if cache is None or cache[0] != key or len(cache[2]) != len(weights) or any(r() is not w for r, w in zip(cache[2], weights)):
        module.__dict__["_zest_packs"] = cache = (key, build(), [weakref.ref(w) for w in weights])
    return cache[1]


def _hip_norm(bn):
    return isinstance(bn, ActivatedBatchNorm) and bn.activation == "leaky_relu" and abs(bn.activation_param - 0.01) < 1e-12


class ConvBnReLU(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1,
                 norm_act=ActivatedBatchNorm):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = norm_act(out_channels)

    def forward(self, x):
        return self.bn(self.conv(x))


class ConvBnReLU3D(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1,
                 norm_act=ActivatedBatchNorm):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = norm_act(out_channels)

    def forward(self, x):
        return self.bn(self.conv(x))


class FeatureNet(nn.Module):
    """Three-level feature pyramid; the top level (32 channels at 1/4 resolution) feeds the
    plane sweep (reference networks.py:962-1001)."""

    def __init__(self, norm_act=ActivatedBatchNorm):
        super().__init__()
        self.conv0 = nn.Sequential(ConvBnReLU(3, 8, 3, 1, 1, norm_act=norm_act),
                                   ConvBnReLU(8, 8, 3, 1, 1, norm_act=norm_act))
        self.conv1 = nn.Sequential(ConvBnReLU(8, 16, 5, 2, 2, norm_act=norm_act),
                                   ConvBnReLU(16, 16, 3, 1, 1, norm_act=norm_act),
                                   ConvBnReLU(16, 16, 3, 1, 1, norm_act=norm_act))
        self.conv2 = nn.Sequential(ConvBnReLU(16, 32, 5, 2, 2, norm_act=norm_act),
                                   ConvBnReLU(32, 32, 3, 1, 1, norm_act=norm_act),
                                   ConvBnReLU(32, 32, 3, 1, 1, norm_act=norm_act))
        self.toplayer = nn.Conv2d(32, 32, 1)

    # ---- HIP path, no autograd: the eight Conv + norm + activation layers as 2-D calls of the regularisation net's
    # convolution kernel (csrc/costreg.hip, one-slice-deep windows), the 1x1 top layer as one matrix product
    _HIP_LAYERS = (("conv0", 0), ("conv0", 1), ("conv1", 0), ("conv1", 1), ("conv1", 2), ("conv2", 0), ("conv2", 1), ("conv2", 2))
    _HIP_SHAPES = ((3, 8, 3, 1), (8, 8, 3, 1), (8, 16, 5, 2), (16, 16, 3, 1), (16, 16, 3, 1), (16, 32, 5, 2), (32, 32, 3, 1),
                   (32, 32, 3, 1))

    def hip_supported(self):
        convs = [getattr(self, s)[i] for s, i in self._HIP_LAYERS]
        shapes = tuple((m.conv.in_channels, m.conv.out_channels, m.conv.kernel_size[0], m.conv.stride[0]) for m in convs)
        top = self.toplayer
        return (shapes == self._HIP_SHAPES and all(_hip_norm(m.bn) for m in convs) and top.kernel_size == (1, 1)
                and top.in_channels == top.out_channels == 32)

    def forward_hip(self, imgs, passes=3, keep=False):
        """imgs [N,3,H,W] -> top-level features, channels-last [N,H/4,W/4,32] (what the plane sweep reads), no graph.
        keep: also the raw output, norm constants and batch moments of every layer (zest_autograd.FeatureFn)."""
        convs = [getattr(self, s)[i] for s, i in self._HIP_LAYERS]
        packs = _cached_packs(self, passes, [m.conv.weight for m in convs],
                              lambda: [pack_conv_weights(m.conv.weight, passes) for m in convs])
        chans = [m.bn.num_features for m in convs]
        offs = [0]
        for c in chans:
            offs.append(offs[-1] + 2 * c)
        rows, dev = zest_hip.costreg_stat_rows(), imgs.device
        stats_all = torch.empty(rows * offs[-1], device=dev, dtype=torch.float64)
        pre_all = torch.empty(offs[-1], device=dev, dtype=torch.float32)
        mom_all = torch.empty(offs[-1], device=dev, dtype=torch.float32) if keep else None
        x = torch.nn.functional.pad(imgs.float().permute(0, 2, 3, 1), (0, 5)).contiguous()        # [N,H,W,8]
        pre, raw, pr, mo = None, [], [], []
        for i, m in enumerate(convs):
            st = stats_all[rows * offs[i]:rows * offs[i + 1]].view(rows, 2, chans[i])
            x = zest_hip.conv2d_cl(x, pre, packs[i], chans[i], m.conv.kernel_size[0], m.conv.stride[0], passes, st)
            mom = mom_all[offs[i]:offs[i + 1]].view(2, chans[i]) if keep else None
            pre = zest_hip.costreg_bn(st, x.numel() // chans[i], m.bn, m.bn.training, pre_all[offs[i]:offs[i + 1]].view(2, chans[i]), mom)
            raw.append(x), pr.append(pre), mo.append(mom)
        act = torch.nn.functional.leaky_relu(x * pre[0] + pre[1], 0.01)
        top = self.toplayer
        bias = top.bias if top.bias is not None else act.new_zeros(32)
        feats = torch.addmm(bias.to(act.dtype), act.view(-1, 32), top.weight.view(32, 32).t().to(act.dtype)).float().view(x.shape)
        return (feats, raw, pr, mo) if keep else feats

    def _upsample_add(self, x, y):
        return torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True) + y

    def forward(self, x):
        """-> (top-level features, the output of every stage).  Only the first element is on the path; the
        list keeps the reference's call shape (its vis_test dumps read it, networks.py:1162-1181)."""
        stages = []
        for stage in (self.conv0, self.conv1, self.conv2, self.toplayer):
            x = stage(x)
            stages.append(x)
        return x, stages


class CostRegNet(nn.Module):
    """3-D U-Net: cost volume -> 8-channel neural encoding volume (reference networks.py:1003-1059)."""

    def __init__(self, in_channels, norm_act=ActivatedBatchNorm):
        super().__init__()
        self.conv0 = ConvBnReLU3D(in_channels, 8, norm_act=norm_act)
        self.conv1 = ConvBnReLU3D(8, 16, stride=2, norm_act=norm_act)
        self.conv2 = ConvBnReLU3D(16, 16, norm_act=norm_act)
        self.conv3 = ConvBnReLU3D(16, 32, stride=2, norm_act=norm_act)
        self.conv4 = ConvBnReLU3D(32, 32, norm_act=norm_act)
        self.conv5 = ConvBnReLU3D(32, 64, stride=2, norm_act=norm_act)
        self.conv6 = ConvBnReLU3D(64, 64, norm_act=norm_act)
        up = lambda i, o: nn.Sequential(
            nn.ConvTranspose3d(i, o, 3, padding=1, output_padding=1, stride=2, bias=False), norm_act(o))
        self.conv7, self.conv9, self.conv11 = up(64, 32), up(32, 16), up(16, 8)

    # ---- HIP path, no autograd (csrc/costreg.hip): channels-last tensors, a layer's norm + activation applied by
    # the layer that reads it, batch statistics from the convolution kernels themselves
    _HIP_CONVS = (("conv0", 1), ("conv1", 2), ("conv2", 1), ("conv3", 2), ("conv4", 1), ("conv5", 2), ("conv6", 1))
    _HIP_UPS = ("conv7", "conv9", "conv11")

    def hip_supported(self):
        """The HIP kernels cover the reference's layer shapes with leaky-ReLU(0.01) norms."""
        bns = [getattr(self, n).bn for n, _ in self._HIP_CONVS] + [getattr(self, n)[1] for n in self._HIP_UPS]
        return (self.conv0.conv.in_channels <= zest_hip.COST_CL_CHANNELS and self.conv0.conv.out_channels == 8
                and all(_hip_norm(b) for b in bns))

    @staticmethod
    def _pack_conv(w, passes):
        return pack_conv_weights(w, passes)

    @staticmethod
    def _pack_deconv(w, passes):
        """ConvTranspose3d weight [cin,cout,3,3,3] -> MFMA A operands [class][oz][oy][chunk][16-row tile][hi, lo][lane]
        x 8 bf16.  Class (pz, py, px) = parity of the output voxel; (oz, oy) = offset of the input row; per dimension
        an even output takes tap k = 1 of input i, an odd one k = 2 of input i and k = 0 of input i + 1.  Lane
        (m, g) of chunk c: output channel 16 nt + m against octet o = 4 c + g of the window of 1 + px pixels."""
        cin, cout = w.shape[:2]
        opt, nt, dev = cin // 8, (cout + 15) // 16, w.device
        ch2 = (2 * opt + 3) // 4
        lane, e = torch.arange(64, device=dev), torch.arange(8, device=dev)
        o = 4 * torch.arange(ch2, device=dev).view(ch2, 1, 1, 1) + (lane >> 4).view(1, 1, 64, 1)
        p, ci = o // opt, 8 * (o % opt) + e.view(1, 1, 1, 8)
        co = 16 * torch.arange(nt, device=dev).view(1, nt, 1, 1) + (lane & 15).view(1, 1, 64, 1)
        full = (ch2, nt, 64, 8)
        wf = w.detach().float()
        out = torch.zeros((8, 2, 2) + full, device=dev)
        tap = lambda par, off: (2 if off == 0 else 0) if par else 1
        for cls in range(8):
            pz, py, px = cls >> 2, (cls >> 1) & 1, cls & 1
            for oz in range(1 + pz):
                for oy in range(1 + py):
                    kx = torch.where(p == 0, tap(px, 0), tap(px, 1)).expand(full)
                    valid = ((p <= px) & (co < cout)).expand(full)
                    vals = wf[ci.expand(full), co.clamp(max=cout - 1).expand(full), tap(pz, oz), tap(py, oy), kx]
                    out[cls, oz, oy] = vals * valid
        hi = out.to(torch.bfloat16)
        parts = [hi] if passes == 1 else [hi, (out - hi.float()).to(torch.bfloat16)]
        return torch.stack(parts, 5).contiguous().view(torch.int16).view(-1)       # [8,2,2,ch2,nt,parts,64,8]

    def _hip_packs(self, passes):
        ws = [getattr(self, n).conv.weight for n, _ in self._HIP_CONVS] + [getattr(self, n)[0].weight for n in self._HIP_UPS]

        def build():
            packs = {n: pack_conv_weights(getattr(self, n).conv.weight, passes) for n, _ in self._HIP_CONVS}
            for n in self._HIP_UPS:
                packs[n] = self._pack_deconv(getattr(self, n)[0].weight, passes)
            return packs
        return _cached_packs(self, passes, ws, build)

    def forward_hip(self, cost_cl, passes=3, keep=False):
        """cost_cl [D,H,W,48] channels-last (zest_hip.volume_cost_cl) -> encoding volume [1,8,D,H,W], no graph.
        passes 3: split-bf16 operands (16 significant bits); 1: bf16 operands (the --precision 16 path).  Norms in
        training mode use (and record) batch statistics, as the library path does.
        keep: also return what a backward pass needs (zest_autograd.CostRegFn): the raw output of every layer, its norm
        constants [2,C] and its batch moments [2,C] (mean, 1/std)."""
        D, H, W, _ = cost_cl.shape
        if D % 8 or H % 8 or W % 8:
            raise RuntimeError("CostRegNet.forward_hip: volume %dx%dx%d is not a multiple of 8 per axis" % (D, H, W))
        packs = self._hip_packs(passes)
        bns = [getattr(self, n).bn for n, _ in self._HIP_CONVS] + [getattr(self, n)[1] for n in self._HIP_UPS]
        chans = [b.num_features for b in bns]
        offs = [0]
        for c in chans:
            offs.append(offs[-1] + 2 * c)
        rows = zest_hip.costreg_stat_rows()
        stats_all = torch.empty(rows * offs[-1], device=cost_cl.device, dtype=torch.float64)
        pre_all = torch.empty(offs[-1], device=cost_cl.device, dtype=torch.float32)
        st = [stats_all[rows * offs[i]:rows * offs[i + 1]].view(rows, 2, c) for i, c in enumerate(chans)]
        pr = [pre_all[offs[i]:offs[i + 1]].view(2, c) for i, c in enumerate(chans)]
        mom_all = torch.empty(offs[-1], device=cost_cl.device, dtype=torch.float32) if keep else None
        mo = [mom_all[offs[i]:offs[i + 1]].view(2, c) if keep else None for i, c in enumerate(chans)]
        norm = lambda i, t: zest_hip.costreg_bn(st[i], t.numel() // chans[i], bns[i], bns[i].training, pr[i], mo[i])
        raw, x, pre = [], cost_cl, None
        for i, (name, stride) in enumerate(self._HIP_CONVS):
            x = zest_hip.costreg_conv(x, pre, packs[name], chans[i], stride, passes, st[i])
            pre = norm(i, x)
            raw.append(x)
        up = zest_hip.costreg_deconv(raw[6], pr[6], None, None, packs["conv7"], chans[7], passes, st[7])
        norm(7, up)
        raw.append(up)
        up = zest_hip.costreg_deconv(raw[4], pr[4], up, pr[7], packs["conv9"], chans[8], passes, st[8])
        norm(8, up)
        raw.append(up)
        up = zest_hip.costreg_deconv(raw[2], pr[2], up, pr[8], packs["conv11"], chans[9], passes, st[9])
        norm(9, up)
        raw.append(up)
        vol = zest_hip.costreg_out(raw[0], pr[0], up, pr[9])
        return (vol, raw, pr, mo) if keep else vol

    def forward(self, x):
        """-> (encoding volume, the output of every level: three on the way down, the bottom, three on the way up).
        Only the first element is on the path (the list: reference call shape, networks.py:1213-1230)."""
        skips = []
        for level in ((self.conv0,), (self.conv1, self.conv2), (self.conv3, self.conv4)):
            for conv in level:
                x = conv(x)
            skips.append(x)
        x = self.conv6(self.conv5(x))
        levels = skips + [x]
        for up, skip in zip((self.conv7, self.conv9, self.conv11), reversed(skips)):
            x = skip + up(x)
            levels.append(x)
        return x, levels


class MVSNet(nn.Module):
    """Encoding-volume builder: 2-D features -> plane-sweep variance cost volume (HIP) -> 3-D
    regularisation.  Constructor, attribute names and state-dict keys follow the reference
    (networks.py:1061-1238)."""

    def __init__(self, num_groups=1, norm_act=ActivatedBatchNorm, levels=1):
        super().__init__()
        self.levels = levels
        self.n_depths = [128, 32, 8]
        self.G = num_groups
        self.feature = FeatureNet()
        self.chunk = 1024
        self.cost_reg_2 = CostRegNet(32 + 9, norm_act)

    def build_volume_cost(self, imgs, feats, proj_mats, depth_values, pad=0):
        """imgs [1,V,3,Hi,Wi]; feats [1,V,32,H,W]; proj_mats [1,V,3,4]; depth_values [1,D]
        -> (img_feat [1,3V+32,D,H+2pad,W+2pad], in_masks [1,V,D,H+2pad,W+2pad]).  One HIP launch
        (zest_volume_cost_fwd); under autograd the gradient reaches `feats` through
        zest_volume_cost_bwd (images, projections and depths are data)."""
        B, V, C, H, W = feats.shape
        if B != 1:
            raise RuntimeError("build_volume_cost: batch must be 1 (the reference assumes it too)")
        imgs_lr = torch.nn.functional.interpolate(imgs.reshape(B * V, *imgs.shape[2:]), (H, W), mode="bilinear",
                                                  align_corners=False)
        depth = depth_values.reshape(depth_values.shape[0], -1)[0]
        if torch.is_grad_enabled() and feats.requires_grad:
            import zest_autograd
            img_feat, masks = zest_autograd.VolumeCostFn.apply(feats[0].float(), imgs_lr.detach(), proj_mats[0, 1:].detach(),
                                                               depth.detach(), pad)
        else:
            img_feat, masks = zest_hip.volume_cost(feats[0], imgs_lr, proj_mats[0, 1:], depth, pad)
        return img_feat[None], masks[None]

    def hip_path(self, imgs, pad=0):
        """True when forward(imgs, ..., pad) runs on the HIP kernels alone: no autograd, the reference's layer shapes,
        a volume of whole octets per axis."""
        B, V, _, H, W = imgs.shape
        half = lambda n: (n - 1) // 2 + 1
        Hf, Wf = half(half(H)), half(half(W))
        return (not torch.is_grad_enabled() and imgs.is_cuda and B == 1 and 3 * V + 32 <= zest_hip.COST_CL_CHANNELS
                and not ((Hf + 2 * pad) % 8 or (Wf + 2 * pad) % 8) and getattr(self, "zest_hip_costreg", True)
                and self.cost_reg_2.hip_supported())

    def forward(self, imgs, proj_mats, near_far, pad=0, return_color=False, lindisp=False,
                vis_test=False, test_dir=None):
        """-> (volume_feat [1,8,128,H/4+2pad,W/4+2pad], feats, depth_values) (reference networks.py:1142-1238).
        Differentiable with respect to the parameters of FeatureNet and CostRegNet."""
        if vis_test:
            raise NotImplementedError("MVSNet.forward: vis_test dumps are a debugging aid of the reference")
        B, V, _, H, W = imgs.shape
        D = 128
        half = lambda n: (n - 1) // 2 + 1
        Hf, Wf = half(half(H)), half(half(W))
        # whole-image evaluation (no autograd): feature pyramid, plane sweep and regularisation net as HIP kernels
        # (csrc/costreg.hip); bf16 operands under autocast (--precision 16), split-bf16 pairs otherwise
        hip = not return_color and self.hip_path(imgs, pad)
        passes = 1 if torch.is_autocast_enabled() else 3
        feats_cl = None
        hip_train = getattr(self, "zest_hip_costreg_train", None)
        hip_train = (torch.is_autocast_enabled() if hip_train is None else bool(hip_train)) and torch.is_grad_enabled() and imgs.is_cuda
        if hip and self.feature.hip_supported():
            feats_cl = self.feature.forward_hip(imgs[0], passes)
            feats = feats_cl.permute(0, 3, 1, 2)[None]
        elif (hip_train and B == 1 and self.feature.hip_supported() and self.feature.toplayer.bias is not None
              and all(m.training for m in self.feature.modules() if isinstance(m, ActivatedBatchNorm))):
            import zest_autograd                      # the pyramid's forward on the HIP kernels under autograd (FeatureFn)
            feats = zest_autograd.feature_apply(self.feature, imgs[0], passes)[None]
        else:
            feats, _ = self.feature(imgs.reshape(B * V, 3, H, W))
            feats = feats.view(B, V, *feats.shape[1:])
        t_vals = torch.linspace(0., 1., steps=D, device=imgs.device, dtype=imgs.dtype)
        near, far = near_far
        if not lindisp:
            depth_values = near * (1. - t_vals) + far * t_vals
        else:
            depth_values = 1. / (1. / near * (1. - t_vals) + 1. / far * t_vals)
        depth_values = depth_values.unsqueeze(0)
        if hip:
            imgs_lr = torch.nn.functional.interpolate(imgs.reshape(B * V, *imgs.shape[2:]), (Hf, Wf), mode="bilinear",
                                                      align_corners=False)
            cost_cl = zest_hip.volume_cost_cl(None if feats_cl is not None else feats[0].float(), imgs_lr.float(),
                                              proj_mats[0, 1:], depth_values[0], pad, feats_cl=feats_cl)
            return self.cost_reg_2.forward_hip(cost_cl, passes=passes), feats, depth_values
        cost_vol, in_masks = self.build_volume_cost(imgs, feats, proj_mats, depth_values, pad=pad)
        Dp_ = cost_vol.shape[2]
        if return_color:
            feats = torch.cat((cost_vol[:, :V * 3].view(B, V, 3, *cost_vol.shape[2:]), in_masks.unsqueeze(2)), dim=2)
        # under autograd: the regularisation net's FORWARD on the HIP kernels (zest_autograd.CostRegFn) - by default in
        # bf16 autocast (--precision 16: a whole-generator step 47 -> 31 ms); in fp32 mode the library's fp32 backward
        # dominates either way and the library modules are kept unless zest_hip_costreg_train is set
        if (hip_train and B == 1
                and not (Dp_ % 8 or cost_vol.shape[-2] % 8 or cost_vol.shape[-1] % 8) and self.cost_reg_2.hip_supported()
                and all(m.training for m in self.cost_reg_2.modules() if isinstance(m, ActivatedBatchNorm))):
            # backward: HIP norm kernels + the library's convolution backward on the kept raw outputs
            import zest_autograd
            volume_feat = zest_autograd.costreg_apply(self.cost_reg_2, cost_vol, passes)
        else:
            volume_feat, _ = self.cost_reg_2(cost_vol)
        volume_feat = volume_feat.reshape(1, -1, *volume_feat.shape[2:])
        return volume_feat, feats, depth_values


# ------------------------------------------------------------------------------------------
# Generators: the callers of the path (reference networks.py:355-720).  Host orchestration only:
# encoding volume(s) -> ray sampling -> rendering, with the reference's constructor arguments,
# batch-dict keys and result keys.  The whole-image loop of forward_val additionally shards its
# ray chunks over the ranks of an initialised torch.distributed group (each rank
# renders its own chunks, ONE all-gather per image) and takes the fused single-launch renderer.
_IMAGENET = ((-0.485 / 0.229, -0.456 / 0.224, -0.406 / 0.225), (1 / 0.229, 1 / 0.224, 1 / 0.225))


class _Generator(nn.Module):
    def unpreprocess(self, data, shape=(1, 1, 3, 1, 1)):
        """Undo the ImageNet normalisation of the loader (N V C H W)."""
        mean = torch.tensor(_IMAGENET[0], device=data.device).view(*shape)
        std = torch.tensor(_IMAGENET[1], device=data.device).view(*shape)
        return (data - mean) / std

    def _volume(self, net, imgs, proj_mats, near_far, bn_batch_stats=False):
        """Encoding volume of `net` (None -> no volume).  With autograd recording and a builder whose
        parameters want gradients (the reference optimises generator.parameters() including the MVSNets,
        train.py:270) the builder runs under autograd: library convolutions around the HIP plane sweep,
        whose backward is zest_volume_cost_bwd; otherwise without a graph."""
        if net is None:
            return None
        if bn_batch_stats:
            net.train()         # the reference validates with batch statistics (networks.py:629, 644)
        # --precision 16 (the reference runs the whole model under AMP then): library convolutions
        # in bf16; the plane sweep and the batch norms stay fp32
        amp = int(getattr(self.args, "precision", 32) or 32) == 16 and imgs.is_cuda
        train = torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters())
        with torch.set_grad_enabled(train), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            if (not train and getattr(self.args, "zest_graph_builders", True) and isinstance(net, MVSNet)
                    and net.feature.hip_supported() and net.hip_path(imgs, self.args.pad)):
                return self._volume_replayed(net, imgs, proj_mats, near_far)
            if train and isinstance(net, MVSNet) and getattr(self.args, "zest_hip_costreg_train", None) is not None:
                net.zest_hip_costreg_train = bool(self.args.zest_hip_costreg_train)       # the caller's choice, else MVSNet's default
            return net(imgs, proj_mats, near_far, pad=self.args.pad)[0].float()

    def _volume_replayed(self, net, imgs, proj_mats, near_far):
        """The all-HIP builder is ~110 kernel launches of a few microseconds each and the host is slower at issuing
        them than the device at running them (3.4 ms of wall time for 1.8 ms of device time per image): the first
        call with a given net, shapes and weights runs it directly and then RECORDS it as a HIP graph (recording
        executes nothing: the norms' running estimates advance once per call, as without the graph); later calls
        copy their inputs into the graph's buffers and replay it.  The volume is returned as a copy."""
        params = list(net.parameters())
        norms = tuple((m.training, m.eps, m.momentum) for m in net.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm))
        key = (tuple(imgs.shape), tuple(proj_mats.shape), imgs.dtype, self.args.pad, torch.is_autocast_enabled(), norms,
               str(imgs.device), getattr(net, "zest_hip_costreg", True), tuple(p._version for p in params))
        graphs = self.__dict__.setdefault("_zest_builder_graphs", {})
        for k in [k for k, e in graphs.items() if e["net"]() is None]:
            del graphs[k]                                   # a builder that no longer exists: free its graph and buffers
        ent = graphs.get(id(net))
        if (ent is not None and ent["net"]() is net and ent["key"] == key and len(ent["params"]) == len(params)
                and all(r() is p for r, p in zip(ent["params"], params))):
            ent["imgs"].copy_(imgs), ent["proj"].copy_(proj_mats), ent["near_far"].copy_(near_far)
            ent["graph"].replay()
            return ent["out"].clone()
        out = net(imgs, proj_mats, near_far, pad=self.args.pad)[0].float()
        graphs.pop(id(net), None)
        ent = dict(key=key, net=weakref.ref(net), params=[weakref.ref(p) for p in params], imgs=imgs.clone(), proj=proj_mats.clone(),
                   near_far=near_far.clone().to(imgs.device), graph=torch.cuda.CUDAGraph())
        cur = torch.cuda.current_stream(imgs.device)
        with torch.cuda.graph(ent["graph"]):
            ent["out"] = net(ent["imgs"], ent["proj"], ent["near_far"], pad=self.args.pad)[0].float()
        torch.cuda.current_stream(imgs.device).wait_stream(cur)
        graphs[id(net)] = ent
        return out


class MVSNeRF_G(_Generator):
    """Static generator (reference networks.py:355-446)."""

    def __init__(self, args, nerf, encoding, embedding_pts, embedding_dir):
        super().__init__()
        self.nerf, self.encoding_net = nerf, encoding
        self.embedding_pts, self.embedding_dir = embedding_pts, embedding_dir
        self.N_rays, self.N_samples, self.args = args.batch_size, args.N_samples, args

    def forward(self, x, step=0, time_codes=None):
        import zest_utils as utils
        from zest_renderer import rendering
        imgs = x['images']
        depths = x['depths_h'] if 'depths_h' in x else x['depths']
        cams = {'w2cs': x['w2cs'], 'intrinsics': x['intrinsics']}
        volume = self._volume(self.encoding_net, imgs[:, :-1], x['proj_mats'][:, :-1], x['near_fars'][0, 0])
        pad = self.args.pad if self.encoding_net is not None else 0
        imgs = self.unpreprocess(imgs)
        rays_pts, rays_dir, target_s, rays_ndc, depth_candidates, rays_depth_gt, t_vals = utils.build_rays(
            imgs, depths, x['w2cs'], x['c2ws'], x['intrinsics'], x['near_fars'], self.N_samples,
            N_rays=self.N_rays, pad=pad, patch_size=self.args.patch_size, scale_anneal=self.args.scale_anneal,
            step=step, variable_patches=(self.args.gan_type == 'graf'))
        ret = rendering(self.args, rays_pts, rays_ndc, depth_candidates, rays_dir, volume_feature_static=volume,
                        imgs=imgs[:, :-1], img_feat=None, im_cam_mat=cams, network_fn=self.nerf,
                        embedding_pts=self.embedding_pts, embedding_dir=self.embedding_dir,
                        time_codes=time_codes, white_bkgd=self.args.white_bkgd)
        ret.update(target_s=target_s, depth_gt=rays_depth_gt, t_vals=t_vals)
        return ret


class DyMVSNeRF_G(_Generator):
    """Static + dynamic (scene-flow) generator (reference networks.py:448-720)."""

    VAL_KEYS = ('rgb_map_ref', 'depth_map_ref', 'rgb_map', 'depth_map', 'rgb_map_ref_dy', 'depth_map_ref_dy',
                'weights_map_dd')

    def __init__(self, args, decay_iteration, nerf_dynamic, nerf_static, encoding, encoding_dy, embedding_pts,
                 embedding_xyzt, embedding_dir):
        super().__init__()
        self.nerf_dynamic, self.nerf_static = nerf_dynamic, nerf_static
        self.encoding_net, self.encoding_net_dy = encoding, encoding_dy
        self.embedding_pts, self.embedding_xyzt, self.embedding_dir = embedding_pts, embedding_xyzt, embedding_dir
        self.N_rays, self.N_samples = args.batch_size, args.N_samples
        self.chain_bwd, self.decay_iteration, self.args = False, decay_iteration, args

    def _scene(self, x, bn_batch_stats=False):
        """Volumes, unnormalised images and cameras of one batch dict."""
        sc = dict(cams={'w2cs': x['w2cs'], 'intrinsics': x['intrinsics']}, nb_frames=None, nb_cams=None)
        imgs, near_far = x['images'], x['near_fars'][0, 0]
        # The two volume builders are independent (different images, different nets).  Without a graph (whole-image
        # evaluation) the dynamic one runs on a second HIP stream beside the static one: the small convolutions of the
        # feature pyramids and the tails of the large ones leave CUs idle that the other builder can use.
        side = None
        if (self.encoding_net_dy is not None and self.encoding_net is not None and imgs.is_cuda
                and not torch.is_grad_enabled() and getattr(self.args, 'zest_overlap_builders', True)):
            side = self.__dict__.get('_zest_side_stream')
            if side is None or side.device != imgs.device:
                side = self.__dict__['_zest_side_stream'] = torch.cuda.Stream(device=imgs.device)
        sc['vol_d'] = None
        if side is not None:
            cur = torch.cuda.current_stream(imgs.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                sc['vol_d'] = self._volume(self.encoding_net_dy, x['nb_imgs'], x['nb_proj_mats'], near_far, bn_batch_stats)
        sc['vol_s'] = self._volume(self.encoding_net, imgs[:, :-1], x['proj_mats'][:, :-1], near_far, bn_batch_stats)
        if self.encoding_net_dy is not None:
            sc['nb_cams'] = {'w2cs': x['nb_w2cs'], 'intrinsics': x['nb_intr']}
            if side is not None:
                torch.cuda.current_stream(imgs.device).wait_stream(side)
                sc['vol_d'].record_stream(torch.cuda.current_stream(imgs.device))
            else:
                sc['vol_d'] = self._volume(self.encoding_net_dy, x['nb_imgs'], x['nb_proj_mats'], near_far, bn_batch_stats)
            sc['nb_frames'] = self.unpreprocess(x['nb_imgs'])
        sc['pad'] = self.args.pad if (self.encoding_net is not None or self.encoding_net_dy is not None) else 0
        sc['imgs'] = self.unpreprocess(imgs)
        frame_t, sc['num_frames'] = x['time'].item(), x['total_frames'].item()
        sc['ref_frame_idx'] = frame_t / sc['num_frames'] * 2. - 1.0
        return sc

    def _render(self, sc, rays, time_codes, **kw):
        from zest_renderer import rendering
        rays_pts, rays_dir, rays_ndc, depth_candidates = rays
        return rendering(self.args, rays_pts, rays_ndc, depth_candidates, rays_dir,
                         volume_feature_static=sc['vol_s'], volume_feature_dynamic=sc['vol_d'],
                         imgs=sc['imgs'][:, :-1], neighbour_frames=sc['nb_frames'], im_cam_mat=sc['cams'],
                         nb_cam_mat=sc['nb_cams'], network_fn=self.nerf_static, network_fn_dy=self.nerf_dynamic,
                         embedding_pts=self.embedding_pts, embedding_xyzt=self.embedding_xyzt,
                         embedding_dir=self.embedding_dir, time_codes=time_codes, white_bkgd=self.args.white_bkgd,
                         scene_flow=True, chain_bwd=self.chain_bwd, ref_frame_idx=sc['ref_frame_idx'],
                         num_frames=sc['num_frames'], **kw)

    def forward(self, x, step=0, time_codes=None):
        import zest_utils as utils
        a = self.args
        chain_5frames = bool(a.with_chain_loss and step > self.decay_iteration * 1000 * 2)
        extra = a.num_extra_samples if (a.use_motion_mask and step < self.decay_iteration * 1000) else 0
        sc = self._scene(x)
        (rays_pts, rays_dir, target_s, rays_ndc, depth_candidates, rays_depth_gt, t_vals, flow_fwd, flow_bwd,
         mask_fwd, mask_bwd) = utils.build_rays_dy(
            sc['imgs'], x['depths'], x['w2cs'], x['c2ws'], x['intrinsics'], x['near_fars'], self.N_samples,
            N_rays=self.N_rays, pad=sc['pad'], patch_size=a.patch_size, scale_anneal=a.scale_anneal,
            num_extra_samples=extra, motion_coords=x['motion_coords'][-1], step=step,
            variable_patches=(a.gan_type == 'graf'), scene_flow=True, flow_fwd=x['flow_fwds'],
            flow_bwd=x['flow_bwds'], mask_fwd=x['mask_fwds'], mask_bwd=x['mask_bwds'])
        self.chain_bwd = not self.chain_bwd          # alternate the direction of the scene-flow chain
        ret = self._render(sc, (rays_pts, rays_dir, rays_ndc, depth_candidates), time_codes,
                           chain_5frames=chain_5frames, raw_noise_std=a.raw_noise_std)
        ret.update(target_s=target_s, depth_gt=rays_depth_gt, t_vals=t_vals, rays_flow_fwd_gt=flow_fwd,
                   rays_flow_bwd_gt=flow_bwd, rays_mask_fwd_gt=mask_fwd, rays_mask_bwd_gt=mask_bwd,
                   chain_bwd=self.chain_bwd, chain_5frames=chain_5frames)
        return ret

    def forward_val(self, x, time_codes=None):
        """Whole-image evaluation in chunks of args.chunk rays -> (imgs, rgbs_blend, depths_blend,
        rgbs_rig, depths_rig, rgbs_dy, depths_dy, weights_dd), each a list the caller concatenates
        (reference networks.py:595-720).  In a torch.distributed group the chunks are split into
        contiguous runs per rank, rendered locally and exchanged with ONE all-gather per image;
        every list then holds a single tensor covering the image."""
        import zest_utils as utils
        import zest_parallel
        a = self.args
        with torch.no_grad():
            sc = self._scene(x, bn_batch_stats=True)
            a.img_downscale = torch.rand((1,)) * 0.75 + 0.25     # as the reference; unused by the lookup
            N, V, C, H, W = sc['imgs'].shape
            n_chunks = (H * W + a.chunk - 1) // a.chunk
            world, rank = zest_parallel._world()
            lo, hi, per = zest_parallel.shard_bounds(n_chunks, world, rank)
            maps_only, a.zest_maps_only = getattr(a, 'zest_maps_only', False), True
            outs = {k: [] for k in self.VAL_KEYS}
            # args.chunk bounds the reference's per-sample tensors; the fused renderer keeps nothing per sample in
            # HBM, so neighbouring chunks go through ONE ray-sampling + ONE rendering launch (up to `group` chunks =
            # args.zest_val_rays rays, default one NSFF image: 288 x 512 x 128 samples are 0.6 GB of ray tensors; 60.7 ms
            # per image against 66.3 with chunk = 1024 launches in round 2, another 0.6 ms against 16k-ray launches).  Rays are
            # independent: the rows are the same, the lists just hold fewer, longer tensors.
            group = max(1, int(getattr(a, 'zest_val_rays', 147456)) // a.chunk)
            try:
                for _ in range(n_chunks):
                    self.chain_bwd = not self.chain_bwd          # every rank keeps the reference's alternation
                chunk_idx = lo
                while chunk_idx < hi:
                    # the largest run of chunks that starts here, is no longer than `group` and that the ray sampler
                    # can address as ONE chunk of n * args.chunk rays (its index must be a whole number)
                    n = next(k for k in range(min(group, hi - chunk_idx), 0, -1) if chunk_idx % k == 0)
                    r = utils.build_rays_dy(
                        sc['imgs'], x['depths'], x['w2cs'], x['c2ws'], x['intrinsics'], x['near_fars'],
                        self.N_samples, N_rays=self.N_rays, stratified=False, pad=sc['pad'], chunk=a.chunk * n,
                        idx=chunk_idx // n, val=True, isRandom=False, scene_flow=True, flow_fwd=x['flow_fwds'],
                        flow_bwd=x['flow_bwds'], mask_fwd=x['mask_fwds'], mask_bwd=x['mask_bwds'],
                        zest_rays_only=True)
                    ret = self._render(sc, (r[0], r[1], r[3], r[4]), time_codes, chain_5frames=False, val=True)
                    for k in self.VAL_KEYS:
                        outs[k].append(ret[k].squeeze(0))
                    chunk_idx += n
            finally:
                a.zest_maps_only = maps_only
            if zest_parallel.collectives_active():
                widths = [3, 1, 3, 1, 3, 1, 1]
                cols = [torch.cat(outs[k]).reshape(-1, w) if outs[k] else sc['imgs'].new_zeros(0, w)
                        for k, w in zip(self.VAL_KEYS, widths)]
                full = zest_parallel.gather_maps(torch.cat(cols, 1), H * W, per=per * a.chunk)
                for k, part in zip(self.VAL_KEYS, torch.split(full, widths, 1)):
                    outs[k] = [part if part.shape[1] > 1 else part[:, 0]]
        return (sc['imgs'],) + tuple(outs[k] for k in self.VAL_KEYS)
