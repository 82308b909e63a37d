"""Gradients of CostRegNet three ways: plain autograd through the library modules on the GPU (fp32), the HIP forward +
library backward operators (zest_autograd.CostRegFn), and plain autograd in float64 on the CPU (the judge of the two)."""
import sys, copy, torch
sys.path.insert(0, "/root/repo/zest-nerf_amd"); sys.path.insert(0, "/root/repo/tests")
import zest_autograd, zest_networks as networks
DEV = "cuda:0"
torch.manual_seed(21)
net, ref = networks.CostRegNet(41).to(DEV).train(), networks.CostRegNet(41).to(DEV).train()
with torch.no_grad():
    for m in net.modules():
        if isinstance(m, networks.ActivatedBatchNorm):
            m.weight.uniform_(0.5, 1.5), m.bias.normal_(0, 0.2)
ref.load_state_dict(net.state_dict())
r64 = copy.deepcopy(ref).cpu().double()
cost = torch.randn(1, 41, 16, 16, 24, device=DEV)
g_out = torch.randn(1, 8, 16, 16, 24, device=DEV)
ca, cb, cc = cost.clone().requires_grad_(True), cost.clone().requires_grad_(True), cost.detach().cpu().double().requires_grad_(True)
ref(cb)[0].backward(g_out)
zest_autograd.costreg_apply(net, ca, 3).backward(g_out)
r64(cc)[0].backward(g_out.cpu().double())
l2 = lambda a, b: float((a.detach().cpu().double() - b).norm() / b.norm().clamp_min(1e-30))
print("g_cost: hip-fwd path %.2e   library path %.2e   (relative L2 against float64)" % (l2(ca.grad, cc.grad), l2(cb.grad, cc.grad)))
for (name, p), (_, q), (_, w) in zip(net.named_parameters(), ref.named_parameters(), r64.named_parameters()):
    if name.startswith(("conv0", "conv1.")):
        print("  %-20s hip-fwd path %.2e   library path %.2e" % (name, l2(p.grad, w.grad), l2(q.grad, w.grad)))
# forward agreement per level and the fraction of pre-activations on the other side of the kink
with torch.no_grad():
    net2 = copy.deepcopy(ref)
    cl = torch.nn.functional.pad(cost[0].permute(1, 2, 3, 0), (0, 7)).contiguous()
    vol, raw, pr, mo = net2.forward_hip(cl, 3, keep=True)
    x = cost.detach().cpu().double()
    outs = {}
    h = x
    names = ["conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6"]
    for i, nm in enumerate(names):
        m = getattr(r64, nm)
        rr = m.conv(h)
        h = m.bn(rr)
        mine = raw[i].permute(3, 0, 1, 2)[None].cpu().double()
        y_mine = mine * pr[i][0].cpu().double().view(1, -1, 1, 1, 1) + pr[i][1].cpu().double().view(1, -1, 1, 1, 1)
        bnm = m.bn
        # library pre-activation in float64
        mu, var = rr.mean((0, 2, 3, 4), keepdim=True), rr.var((0, 2, 3, 4), unbiased=False, keepdim=True)
        y_ref = (rr - mu) / (var + bnm.eps).sqrt() * bnm.weight.view(1, -1, 1, 1, 1) + bnm.bias.view(1, -1, 1, 1, 1)
        flips = float(((y_mine > 0) != (y_ref > 0)).double().mean())
        print("%s raw rel err %.2e   pre-activation abs err %.2e   sign flips %.2e" % (
            nm, float((mine - rr).abs().max() / rr.abs().max()), float((y_mine - y_ref).abs().max()), flips))
