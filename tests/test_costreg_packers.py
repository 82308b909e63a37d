"""CPU: the weight packers of the HIP regularisation net (zest_networks.pack_conv_weights, CostRegNet._pack_deconv).
The kernels of csrc/costreg.hip contract, per MFMA, lane (m, g) of the packed operand - output channel 16 nt + m,
octet o = 4 c + g of a voxel's x-window - with the 8 channels of that octet; this test performs the same contraction
with index arithmetic written out from the kernels' comments (numpy, one output row) and compares it with
torch's convolutions: a packer that put a weight at the wrong (tap, channel, lane) position fails here, without a GPU."""
import numpy as np
import pytest
import torch

import zest_networks as networks


def _unpack(packed, dims, passes):
    """int16 stream -> float array [..., 64 lanes, 8] (hi + lo)."""
    parts = 1 if passes == 1 else 2
    a = packed.view(torch.bfloat16).float().reshape(*dims, parts, 64, 8)
    return a.sum(-3).numpy()


@pytest.mark.parametrize("passes", [1, 3])
@pytest.mark.parametrize("cin,cout,K,KD,stride", [(16, 8, 3, 3, 1), (8, 16, 3, 3, 2), (41, 8, 3, 3, 1), (3, 8, 3, 1, 1), (8, 16, 5, 1, 2)])
def test_conv_packer_reproduces_the_convolution(cin, cout, K, KD, stride, passes):
    g = torch.Generator().manual_seed(cin + cout + K)
    w = torch.randn(cout, cin, KD, K, K, generator=g)
    if passes == 1:
        w = w.to(torch.bfloat16).float()                     # one part: exactly representable weights
    cpad = (cin + 7) // 8 * 8
    opt, nt = cpad // 8, (cout + 15) // 16
    cpr = (K * opt + 3) // 4
    A = _unpack(networks.pack_conv_weights(w if KD > 1 else w[:, :, 0], passes), (KD, K, cpr, nt), passes)
    D, H, W = (4, 5, 9) if KD > 1 else (2, 6, 11)
    x = torch.randn(1, cin, D, H, W, generator=g)
    if KD > 1:
        want = torch.nn.functional.conv3d(x, w, stride=stride, padding=1)[0]
    else:                                                     # the slices are images of a batch
        want = torch.nn.functional.conv2d(x[0].permute(1, 0, 2, 3), w[:, :, 0], stride=stride, padding=K // 2).permute(1, 0, 2, 3)
    xp = np.zeros((D, H, W, cpad), np.float32)
    xp[..., :cin] = x[0].permute(1, 2, 3, 0).numpy()
    sz, pz, pad = (stride, 1, K // 2) if KD > 1 else (1, 0, K // 2)
    got = np.zeros(tuple(want.shape), np.float64)
    Do, Ho, Wo = want.shape[1:]
    for z in range(Do):
        for y in range(Ho):
            for xo in range(Wo):
                for dz in range(KD):
                    for dy in range(K):
                        zi, yi = sz * z + dz - pz, stride * y + dy - pad
                        if not (0 <= zi < D and 0 <= yi < H):
                            continue
                        for c in range(cpr):
                            for gq in range(4):
                                o = 4 * c + gq
                                p, q = o // opt, o % opt
                                xi = stride * xo - pad + p
                                if o >= K * opt or not 0 <= xi < W:
                                    continue
                                v = xp[zi, yi, xi, 8 * q:8 * q + 8]
                                for co in range(cout):
                                    got[co, z, y, xo] += float(A[dz, dy, c, co // 16, (co % 16) + 16 * gq] @ v)
    tol = 1e-5 if passes == 1 else 3e-4                       # hi + lo carries 16 bits of every weight
    assert np.abs(got - want.numpy()).max() < tol * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("cin,cout", [(16, 8), (32, 16)])
def test_deconv_packer_reproduces_the_transposed_convolution(cin, cout):
    g = torch.Generator().manual_seed(cin)
    w = torch.randn(cin, cout, 3, 3, 3, generator=g).to(torch.bfloat16).float()
    opt, nt = cin // 8, (cout + 15) // 16
    ch2 = (2 * opt + 3) // 4
    A = _unpack(networks.CostRegNet._pack_deconv(w, 1), (8, 2, 2, ch2, nt), 1)
    D, H, W = 2, 3, 4
    x = torch.randn(1, cin, D, H, W, generator=g)
    want = torch.nn.functional.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1)[0].numpy()
    xp = x[0].permute(1, 2, 3, 0).numpy()
    got = np.zeros_like(want, dtype=np.float64)
    for cls in range(8):
        pz, py, px = cls >> 2, (cls >> 1) & 1, cls & 1
        for zi in range(D):
            for yi in range(H):
                for xi in range(W):
                    for oz in range(1 + pz):
                        for oy in range(1 + py):
                            if zi + oz >= D or yi + oy >= H:
                                continue
                            for c in range(ch2 if px else (opt + 3) // 4):
                                for gq in range(4):
                                    o = 4 * c + gq
                                    p, q = o // opt, o % opt
                                    if p > 1 or xi + p >= W:
                                        continue
                                    v = xp[zi + oz, yi + oy, xi + p, 8 * q:8 * q + 8]
                                    for co in range(cout):
                                        got[co, 2 * zi + pz, 2 * yi + py, 2 * xi + px] += float(
                                            A[cls, oz, oy, c, co // 16, (co % 16) + 16 * gq] @ v)
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max())
