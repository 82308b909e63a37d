"""GPU: loss-side reductions over a ray's samples against the reference's
fixtures (values and the reference's own autograd gradients) and the oracle."""
import numpy as np
import pytest
import torch

import golden_cases as gc
from test_hip_ops import G, close, ATOL, RTOL

pytestmark = pytest.mark.gpu


def test_distortion_loss_matches_reference(hip):
    import zest_losses as losses
    inp, gold = gc.build("loss_side"), gc.load_golden("loss_side")
    w = G(inp["weights"]).requires_grad_(True)
    loss = losses.distortion_loss(w, G(inp["t_vals"]))
    (2.0 * loss).backward()
    close(loss.detach().reshape(1), gold["distortion"], name="distortion")
    close(w.grad[0], 2.0 * gold["distortion_dw"], name="d distortion / d weights")
    with torch.no_grad():                                       # no graph: forward only
        close(losses.distortion_loss(G(inp["weights"]), G(inp["t_vals"])).reshape(1), gold["distortion"], name="no grad")


def test_distortion_loss_per_ray_positions_and_long_rays(hip):
    """t_vals per ray (the C ABI's extension; the reference's broadcasting only admits [1,S]) and
    S beyond one wave, against the oracle."""
    import zest_hip
    from oracle import zest_oracle as zo
    for seed, R, S, jitter in ((5, 9, 40, True), (6, 3, 193, True), (7, 5, 2, False)):
        inp = gc.loss_inputs(seed, R=R, S=S, jitter=jitter)
        w = torch.from_numpy(inp["weights"])[0].requires_grad_(True)
        want = zo.distortion_loss(w, torch.from_numpy(inp["t_vals"]))
        want.backward()
        loss_ray, grad = zest_hip.distortion(G(inp["weights"])[0], G(inp["t_vals"]))
        close(loss_ray.sum().reshape(1), want.detach().numpy().reshape(1), name="loss S=%d" % S)
        close(grad, w.grad.numpy(), name="grad S=%d" % S)


def test_projection_from_ndc_matches_reference(hip):
    import zest_utils as utils
    inp, gold = gc.build("loss_side"), gc.load_golden("loss_side")
    w, pts = G(inp["weights"]).requires_grad_(True), G(inp["pts"]).requires_grad_(True)
    uv = utils.projection_from_ndc(G(inp["w2c"]), inp["H"], inp["W"], inp["f"], w, pts)
    assert tuple(uv.shape) == (1,) + gold["uv"].shape
    (uv * G(inp["gw"])).sum().backward()
    scale = np.abs(gold["uv"]).max()
    close(uv[0], gold["uv"], atol=ATOL * scale, name="uv")
    close(w.grad[0], gold["uv_dw"], atol=ATOL * np.abs(gold["uv_dw"]).max(), name="d uv / d weights")
    close(pts.grad[0], gold["uv_dpts"], atol=ATOL * np.abs(gold["uv_dpts"]).max(), name="d uv / d pts")
