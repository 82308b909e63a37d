"""GPU: the HIP regularisation net of the MVS volume builder (csrc/costreg.hip) against the library convolutions of
the same modules (networks.CostRegNet under torch: nn.Conv3d / nn.ConvTranspose3d / batch norm / leaky ReLU).

Parity with the REFERENCE is unpinned for this component (inplace_abn is not importable, nothing in the
reference tests the builder); what is checked here is that the HIP path computes what the module definition says.
Tolerances: split-bf16 operands (passes = 3) carry 16 significant bits per product - 2e-4 of the output scale per
layer; bf16 operands (passes = 1, the --precision 16 path) 8 bits - 2e-2 of the output scale.  The norm constants
and the final addition are fp32.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LAYERS = [(41, 8, 1), (8, 16, 2), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 64, 2), (64, 64, 1)]


def _cl(x):            # [1,C,D,H,W] -> [D,H,W,C]
    return x[0].permute(1, 2, 3, 0).contiguous()


def _cf(x):            # [D,H,W,C] -> [1,C,D,H,W]
    return x.permute(3, 0, 1, 2)[None]


def _rel(got, want):
    got, want = got.detach(), want.detach()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("passes,tol", [(3, 2e-4), (1, 2e-2)])
@pytest.mark.parametrize("cin,cout,stride", LAYERS)
def test_conv_layer_against_library(hip, cin, cout, stride, passes, tol):
    import zest_hip
    import zest_networks as networks
    g = torch.Generator(device=DEV).manual_seed(cin * 100 + cout + stride)
    D, H, W = (6, 10, 22) if cin <= 16 else (4, 6, 20)          # x extent: a full block of 16 and a ragged one
    cpad = (cin + 7) // 8 * 8
    x = torch.randn(1, cpad, D, H, W, device=DEV, generator=g)
    x[:, cin:] = 0
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV, generator=g) / (27 * cin) ** 0.5
    pre = None
    xin = x[:, :cin]
    if cin != 41:                                                # every layer but the first reads a normalised input
        pre = torch.stack([torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g) * 0.3])
        xin = torch.nn.functional.leaky_relu(xin * pre[0].view(1, -1, 1, 1, 1) + pre[1].view(1, -1, 1, 1, 1), 0.01)
    want = torch.nn.functional.conv3d(xin, w, stride=stride, padding=1)
    stats = zest_hip.costreg_stats(cout, DEV)
    got = zest_hip.costreg_conv(_cl(x), pre, networks.CostRegNet._pack_conv(w, passes), cout, stride, passes, stats)
    assert tuple(got.shape) == tuple(want.shape[2:]) + (cout,)
    assert _rel(_cf(got), want) < tol
    # the batch statistics are those of the kernel's own output
    flat = got.double().reshape(-1, cout)
    used = int(stats[-1, 0, 0])
    assert 1 <= used <= stats.shape[0] - 1
    assert torch.allclose(stats[:used].sum(0)[0], flat.sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(stats[:used].sum(0)[1], flat.square().sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("passes,tol", [(3, 2e-4), (1, 2e-2)])
@pytest.mark.parametrize("cin,cout,two", [(64, 32, False), (32, 16, True), (16, 8, True)])
def test_deconv_layer_against_library(hip, cin, cout, two, passes, tol):
    import zest_hip
    import zest_networks as networks
    g = torch.Generator(device=DEV).manual_seed(cin + cout)
    D, H, W = 3, 5, 21                                          # x extent: a full block of 16 and a ragged one
    x0, x1 = (torch.randn(1, cin, D, H, W, device=DEV, generator=g) for _ in range(2))
    p0, p1 = (torch.stack([torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g) * 0.3])
              for _ in range(2))
    act = lambda x, p: torch.nn.functional.leaky_relu(x * p[0].view(1, -1, 1, 1, 1) + p[1].view(1, -1, 1, 1, 1), 0.01)
    w = torch.randn(cin, cout, 3, 3, 3, device=DEV, generator=g) / (8 * cin) ** 0.5
    xin = act(x0, p0) + (act(x1, p1) if two else 0)
    want = torch.nn.functional.conv_transpose3d(xin, w, stride=2, padding=1, output_padding=1)
    stats = zest_hip.costreg_stats(cout, DEV)
    got = zest_hip.costreg_deconv(_cl(x0), p0, _cl(x1) if two else None, p1 if two else None,
                                  networks.CostRegNet._pack_deconv(w, passes), cout, passes, stats)
    assert tuple(got.shape) == (2 * D, 2 * H, 2 * W, cout)
    assert _rel(_cf(got), want) < tol
    flat = got.double().reshape(-1, cout)
    used = int(stats[-1, 0, 0])
    assert 1 <= used <= stats.shape[0] - 1
    assert torch.allclose(stats[:used].sum(0)[0], flat.sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(stats[:used].sum(0)[1], flat.square().sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("passes,tol", [(3, 2e-4), (1, 2e-2)])
@pytest.mark.parametrize("cin,cout,k,stride", [(3, 8, 3, 1), (8, 8, 3, 1), (8, 16, 5, 2), (16, 16, 3, 1), (16, 32, 5, 2), (32, 32, 3, 1)])
def test_conv2d_layer_against_library(hip, cin, cout, k, stride, passes, tol):
    """The layers of FeatureNet: the same kernel with one-slice-deep windows over a batch of images."""
    import zest_hip
    import zest_networks as networks
    g = torch.Generator(device=DEV).manual_seed(cin * 100 + cout + k)
    N, H, W = 3, 18, 37
    cpad = (cin + 7) // 8 * 8
    x = torch.randn(N, cpad, H, W, device=DEV, generator=g)
    x[:, cin:] = 0
    w = torch.randn(cout, cin, k, k, device=DEV, generator=g) / (k * k * cin) ** 0.5
    pre, xin = None, x[:, :cin]
    if cin != 3:
        pre = torch.stack([torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g) * 0.3])
        xin = torch.nn.functional.leaky_relu(xin * pre[0].view(1, -1, 1, 1) + pre[1].view(1, -1, 1, 1), 0.01)
    want = torch.nn.functional.conv2d(xin, w, stride=stride, padding=k // 2)
    stats = zest_hip.costreg_stats(cout, DEV)
    got = zest_hip.conv2d_cl(x.permute(0, 2, 3, 1).contiguous(), pre, networks.pack_conv_weights(w, passes), cout, k, stride,
                             passes, stats)
    assert tuple(got.shape) == (N,) + tuple(want.shape[2:]) + (cout,)
    assert _rel(got.permute(0, 3, 1, 2), want) < tol
    used = int(stats[-1, 0, 0])
    assert torch.allclose(stats[:used].sum(0)[0], got.double().reshape(-1, cout).sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("passes,tol", [(3, 1e-3), (1, 6e-2)])
def test_feature_pyramid_against_library(hip, training, passes, tol):
    import zest_networks as networks
    torch.manual_seed(5)
    net, ref = networks.FeatureNet().to(DEV), networks.FeatureNet().to(DEV)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, networks.ActivatedBatchNorm):
                m.weight.uniform_(0.5, 1.5), m.bias.normal_(0, 0.2), m.running_mean.normal_(0, 0.1), m.running_var.uniform_(0.5, 2)
    ref.load_state_dict(net.state_dict())
    net.train(training), ref.train(training)
    imgs = torch.randn(3, 3, 40, 72, device=DEV)
    with torch.no_grad():
        want = ref(imgs)[0]
        got = net.forward_hip(imgs, passes=passes)
    assert tuple(got.shape) == (3, 10, 18, 32) and tuple(want.shape) == (3, 32, 10, 18)
    assert _rel(got.permute(0, 3, 1, 2), want) < tol
    for a, b in zip(net.modules(), ref.modules()):
        if isinstance(a, networks.ActivatedBatchNorm):
            assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == (1 if training else 0)
            assert torch.allclose(a.running_var, b.running_var, rtol=5 * tol, atol=5 * tol)


def test_norm_constants_and_running_estimates(hip):
    import zest_hip
    import zest_networks as networks
    torch.manual_seed(3)
    C, N = 16, 5000
    x = torch.randn(N, C, device=DEV) * 2 + 0.7
    bn, ref = networks.ActivatedBatchNorm(C).to(DEV), networks.ActivatedBatchNorm(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.normal_()
        ref.load_state_dict(bn.state_dict())
    stats = zest_hip.costreg_stats(C, DEV)
    stats[:] = float("nan")                                      # rows past the count are never read
    stats[:7] = 0
    stats[5], stats[2] = torch.stack([x[:3000].double().sum(0), x[:3000].double().square().sum(0)]), \
        torch.stack([x[3000:].double().sum(0), x[3000:].double().square().sum(0)])
    stats[-1, 0, 0] = 7
    pre = torch.empty(2, C, device=DEV)
    for training in (True, False):
        bn.train(training), ref.train(training)
        with torch.no_grad():
            want = ref(x.t()[None])[0].t()                       # [N,C] through the library norm (updates ref's estimates)
        zest_hip.costreg_bn(stats, N, bn, training, pre)
        got = torch.nn.functional.leaky_relu(x * pre[0] + pre[1], 0.01)
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-5)
        assert torch.allclose(bn.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
        assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1


def test_cost_volume_channels_last_equals_planes(hip):
    import golden_cases as gc
    import zest_hip
    c = gc.cost_inputs(5, V=3, H=12, W=20)
    G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    feats, proj, depth = G(c["feats"])[0], G(c["proj_mats"])[0, 1:], G(c["depth_values"])[0]
    imgs = torch.nn.functional.interpolate(G(c["imgs"])[0], (12, 20), mode="bilinear", align_corners=False)
    planes, _ = zest_hip.volume_cost(feats, imgs, proj, depth, pad=2)
    cl = zest_hip.volume_cost_cl(feats, imgs, proj, depth, pad=2)
    assert tuple(cl.shape) == tuple(planes.shape[1:]) + (48,)
    assert torch.equal(cl[..., :41], planes.permute(1, 2, 3, 0)) and float(cl[..., 41:].abs().max()) == 0.0


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("passes,tol", [(3, 1e-3), (1, 6e-2)])
def test_regularisation_net_against_library(hip, training, passes, tol):
    import zest_networks as networks
    torch.manual_seed(11)
    net, ref = networks.CostRegNet(41).to(DEV), networks.CostRegNet(41).to(DEV)
    with torch.no_grad():
        for m in net.modules():                                 # running estimates and affine terms away from their defaults
            if isinstance(m, networks.ActivatedBatchNorm):
                m.weight.uniform_(0.5, 1.5), m.bias.normal_(0, 0.2), m.running_mean.normal_(0, 0.1), m.running_var.uniform_(0.5, 2)
    ref.load_state_dict(net.state_dict())
    net.train(training), ref.train(training)
    D, H, W = 16, 24, 40
    cost = torch.randn(1, 48, D, H, W, device=DEV)
    cost[:, 41:] = 0
    with torch.no_grad():
        want, _ = ref(cost[:, :41])
        got = net.forward_hip(_cl(cost), passes=passes)
    assert tuple(got.shape) == tuple(want.shape) == (1, 8, D, H, W)
    assert _rel(got, want) < tol
    for a, b in zip(net.modules(), ref.modules()):
        if isinstance(a, networks.ActivatedBatchNorm):
            assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == (1 if training else 0)
            assert torch.allclose(a.running_mean, b.running_mean, rtol=5 * tol, atol=5 * tol)
            assert torch.allclose(a.running_var, b.running_var, rtol=5 * tol, atol=5 * tol)
    with torch.no_grad():                                       # the batch statistics are summed in a fixed order
        assert torch.equal(net.forward_hip(_cl(cost), passes=passes), net.forward_hip(_cl(cost), passes=passes))
    # weights are repacked when they change
    with torch.no_grad():
        net.conv2.conv.weight.mul_(1.5), ref.conv2.conv.weight.mul_(1.5)
        net.conv9[0].weight.mul_(0.5), ref.conv9[0].weight.mul_(0.5)
        assert _rel(net.forward_hip(_cl(cost), passes=passes), ref(cost[:, :41])[0]) < tol
        # ... and when a parameter OBJECT is replaced (a fresh tensor at version 0)
        for m in (net, ref):
            m.conv4.conv.weight = torch.nn.Parameter(m.conv4.conv.weight.detach() * 0.5)
        assert _rel(net.forward_hip(_cl(cost), passes=passes), ref(cost[:, :41])[0]) < tol


@pytest.mark.parametrize("precision", [32, 16])
def test_builder_takes_the_hip_net_without_a_graph(hip, precision, monkeypatch):
    """MVSNet.forward under no_grad: plane sweep + HIP nets == plane sweep + library nets; what it takes with autograd
    recording."""
    import test_generators as tg
    import zest_networks as networks
    x = tg._batch(17)
    net = networks.MVSNet().to(DEV)
    calls = []
    orig = networks.CostRegNet.forward_hip
    monkeypatch.setattr(networks.CostRegNet, "forward_hip", lambda self, *a, **k: (calls.append(k), orig(self, *a, **k))[1])
    imgs, proj, nf = x["images"][:, :3], x["proj_mats"][:, :3], x["near_fars"][0, 0]
    amp = torch.autocast("cuda", dtype=torch.bfloat16, enabled=precision == 16)
    with torch.no_grad(), amp:
        got = net(imgs, proj, nf, pad=4)[0].float()
        net.zest_hip_costreg = False
        want = net(imgs, proj, nf, pad=4)[0].float()
    assert len(calls) == 1 and calls[0]["passes"] == (1 if precision == 16 else 3)
    assert tuple(got.shape) == tuple(want.shape) == (1, 8, 128, 16, 16)
    assert _rel(got, want) < (1e-3 if precision == 32 else 0.1)
    net.zest_hip_costreg = True
    net.requires_grad_(True)
    with amp:
        net(imgs, proj, nf, pad=4)[0].sum().backward()
    # under autograd: library modules in fp32 mode; in bf16 autocast the regularisation net's forward on the HIP kernels
    # (zest_autograd.CostRegFn) unless zest_hip_costreg_train says otherwise
    assert len(calls) == (2 if precision == 16 else 1) and net.cost_reg_2.conv0.conv.weight.grad is not None
    assert net.feature.conv0[0].conv.weight.grad is not None
    net.zest_hip_costreg_train = False
    with amp:
        net(imgs, proj, nf, pad=4)[0].sum().backward()
    assert len(calls) == (2 if precision == 16 else 1)


@pytest.mark.parametrize("passes,tol", [(3, 2e-2), (1, 0.35)])
def test_regularisation_net_trains_through_the_hip_forward(hip, passes, tol):
    """zest_autograd.CostRegFn: forward on the HIP kernels, backward = the library's backward operators on the kept raw
    outputs.  Gradients of the input, of every convolution weight and of every norm's weight and bias against plain
    autograd through the library modules (same weights, training-mode norms).  The two forwards differ by 5e-6 .. 2e-5
    of a layer's output scale (split-bf16 products; 1e-2 with bf16 operands), so a pre-activation within that distance
    of zero falls on the other side of the leaky ReLU's kink - 2e-5 of the first layer's 393 k elements here
    (tools/dbg_costreg_train.py) - and the gradient through it differs by the slope ratio 100: each path is the exact
    gradient of ITS forward, and the relative L2 difference of two such gradients is sqrt(flipped fraction) ~ 5e-3
    (layers without a flipped element agree to 1e-5).  Bounds: relative L2 error per gradient tensor 2e-2 with
    split-bf16 operands, 0.35 with bf16 operands (against the library's bf16 autocast path, itself that far from
    fp32)."""
    import zest_autograd
    import zest_networks as networks
    torch.manual_seed(21)
    net, ref = networks.CostRegNet(41).to(DEV).train(), networks.CostRegNet(41).to(DEV).train()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, networks.ActivatedBatchNorm):
                m.weight.uniform_(0.5, 1.5), m.bias.normal_(0, 0.2)
    ref.load_state_dict(net.state_dict())
    cost = torch.randn(1, 41, 16, 16, 24, device=DEV)
    g_out = torch.randn(1, 8, 16, 16, 24, device=DEV)
    ca, cb = cost.clone().requires_grad_(True), cost.clone().requires_grad_(True)
    amp = torch.autocast("cuda", dtype=torch.bfloat16, enabled=passes == 1)
    with amp:
        want = ref(cb)[0].float()
    want.backward(g_out)
    got = zest_autograd.costreg_apply(net, ca, passes)
    got.backward(g_out)
    assert _rel(got, want) < (1e-3 if passes == 3 else 6e-2)
    l2 = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert l2(ca.grad, cb.grad) < tol
    for (name, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and l2(p.grad, q.grad) < tol, name
    for a, b in zip(net.modules(), ref.modules()):                # the running estimates advanced once, as in the library path
        if isinstance(a, networks.ActivatedBatchNorm):
            assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 1


@pytest.mark.parametrize("passes,tol", [(3, 2e-2), (1, 0.35)])
def test_feature_pyramid_trains_through_the_hip_forward(hip, passes, tol):
    """zest_autograd.FeatureFn against plain autograd through the library modules; bounds as for the regularisation net."""
    import zest_autograd
    import zest_networks as networks
    torch.manual_seed(23)
    net, ref = networks.FeatureNet().to(DEV).train(), networks.FeatureNet().to(DEV).train()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, networks.ActivatedBatchNorm):
                m.weight.uniform_(0.5, 1.5), m.bias.normal_(0, 0.2)
    ref.load_state_dict(net.state_dict())
    imgs = torch.randn(3, 3, 40, 72, device=DEV)
    g_out = torch.randn(3, 32, 10, 18, device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=passes == 1):
        want = ref(imgs)[0].float()
    want.backward(g_out)
    got = zest_autograd.feature_apply(net, imgs, passes)
    got.backward(g_out)
    assert tuple(got.shape) == tuple(want.shape) and _rel(got, want) < (1e-3 if passes == 3 else 6e-2)
    l2 = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    for (name, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and l2(p.grad, q.grad) < tol, name
