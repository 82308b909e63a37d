"""Drop-in `networks` module for the rendering hot path (MI355X / HIP).

Exports the classes the reference's train.py and renderer bind by name
(/root/reference/train.py:36-37, networks.py:25-26): `Embedding`, `Renderer`,
`Renderer_linear`, `MVSNeRF`.  Constructors, attribute names and state-dict keys/shapes
match the reference (networks.py:29-353) so its checkpoints load with
`load_state_dict`; `forward` runs the hand-written gfx950 kernels through the C ABI
(include/zest_render.h).  There is no PyTorch implementation of the forward pass here:
without libzest_hip.so, or on CPU tensors, calls raise.

Precision: fp32 (exact-product MFMA, parity mode) unless the hyper-parameters say
`precision=16` (the reference's --precision flag, opt.py:69) or ZEST_PRECISION=bf16 is
set, which selects the bf16 MFMA engine.
"""
import os

import torch
import torch.nn as nn

import zest_hip

__all__ = ["Embedding", "Renderer", "Renderer_linear", "MVSNeRF", "resolve_precision"]


def resolve_precision(args=None):
    """-> zest_hip.PREC_F32 | PREC_BF16 from ZEST_PRECISION or args.precision (16|32)."""
    env = os.environ.get("ZEST_PRECISION", "").lower()
    if env in ("bf16", "16"):
        return zest_hip.PREC_BF16
    if env in ("fp32", "f32", "32"):
        return zest_hip.PREC_F32
    if args is not None and int(getattr(args, "precision", 32) or 32) == 16:
        return zest_hip.PREC_BF16
    return zest_hip.PREC_F32


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
        if D != 8 or W != 256 or list(skips) != [4] or not use_viewdirs:
            raise NotImplementedError(
                "zest MLP kernels cover the shipped architecture only: D=8, W=256, skips=[4], "
                "use_viewdirs=True (got D=%s W=%s skips=%s viewdirs=%s)" % (D, W, skips, use_viewdirs))
        self.D, self.W, self.skips, self.use_viewdirs = D, W, skips, use_viewdirs
        self.in_ch_pts, self.in_ch_views, self.in_ch_feat = input_ch, input_ch_views, input_ch_feat
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

    def packed(self, precision):
        """Weights in MFMA stream order; re-packed when any parameter changes."""
        params = dict(self.named_parameters())
        stamp = tuple((p.data_ptr(), p._version) for p in params.values())
        hit = self._packed.get(precision)
        if hit is None or hit[0] != stamp:
            desc = self._desc()
            state = {"nerf." + k: v.detach() for k, v in params.items()}
            hit = (stamp, zest_hip.mlp_pack(desc, precision, zest_hip.param_table(state, desc)))
            self._packed[precision] = hit
        return hit[1]

    def zest_forward(self, x, precision=None):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("zest MLP: use MVSNeRF.forward (it owns the `nerf.` parameter names the "
                               "training path needs)")
        desc = self._desc()
        prec = resolve_precision() if precision is None else precision
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
        return zest_hip.MlpDesc(self.in_ch_pts, self.in_ch_feat, self.in_ch_views,
                                int(bool(self.use_mvs)), 0, head)

    def forward_alpha(self, x):
        raise NotImplementedError("forward_alpha is dead code in the reference "
                                  "(renderer.py:295 never sets alpha_only); not implemented")


class Renderer_linear(_MlpBase):
    """'v2' net: additive modulation, relu(alpha), sigmoid(rgb) (reference networks.py:223-319)."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, input_ch_feat=8,
                 skips=[4], use_viewdirs=False):
        super().__init__()
        self._build(D, W, input_ch, input_ch_views, input_ch_feat, skips, use_viewdirs)

    def _desc(self):
        return zest_hip.MlpDesc(self.in_ch_pts, self.in_ch_feat, self.in_ch_views, 1, 2,
                                zest_hip.HEAD_NONE)

    def forward_alpha(self, x):
        raise NotImplementedError("forward_alpha is dead code in the reference; not implemented")


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

    def packed(self, precision):
        return self.nerf.packed(precision)

    def forward_alpha(self, x):
        return self.nerf.forward_alpha(x)

    def zest_forward(self, x, precision=None):
        return self.nerf.zest_forward(x, precision)

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            import zest_autograd
            return zest_autograd.mlp_apply(self, x)          # fp32 training path (HIP fwd + bwd)
        return self.nerf(x)
