"""Drop-in `losses` names for the loss-side reductions over a ray's samples.

`distortion_loss` keeps the reference's signature (/root/reference/losses.py:53-87); the O(S^2)
pair sum per ray runs in one HIP launch (csrc/losses.hip) that also yields the weight gradient,
instead of building [N,R,S,S] tensors.  The other names of the reference's losses.py are
elementwise image-space terms outside this path.
"""
import torch

import zest_autograd

__all__ = ["distortion_loss"]


def distortion_loss(ray_weights, t_vals):
    """ray_weights [N,R,S] (N = 1), t_vals [1,S] or [R,S] normalised sample positions -> scalar."""
    if ray_weights.dim() != 3 or ray_weights.shape[0] != 1:
        raise RuntimeError("distortion_loss: ray_weights must be [1, N_rays, N_samples], got %s"
                           % (tuple(ray_weights.shape),))
    t = t_vals.reshape(-1, t_vals.shape[-1])
    if t.shape[0] not in (1, ray_weights.shape[1]) or t.shape[1] != ray_weights.shape[2]:
        raise RuntimeError("distortion_loss: t_vals %s does not match weights %s"
                           % (tuple(t_vals.shape), tuple(ray_weights.shape)))
    return zest_autograd.DistortionFn.apply(ray_weights[0], t.detach())
