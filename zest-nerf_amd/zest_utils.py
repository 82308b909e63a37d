"""Drop-in `utils` functions on the rendering hot path (MI355X / HIP).

`index_point_feature` and `build_color_volume` keep the reference's signatures
(/root/reference/utils.py:433-505) and run the gather kernels behind the C ABI.  The
channels-last copies the kernels read (volume [H,W,D,8] - depth innermost -, images [V,H,W,4]) are made once per
tensor and cached by storage identity + version: the reference builds a volume once per
image and renders ~144 ray chunks from it (networks.py:660).
"""
import weakref

import torch

import zest_hip

__all__ = ["homo_warp", "index_point_feature", "build_color_volume", "volume_channels_last",
           "images_channels_last", "get_ndc_coordinate", "get_rays_mvs", "build_rays_base",
           "build_rays", "build_rays_dy", "sample_pdf", "patch_ray_sampler", "projection_from_ndc"]

_CL_CACHE = {}
_CL_CACHE_MAX = 8


def _cached(kind, t, make):
    """Channels-last copy of `t`, made once per (memory, layout, content version).  The key is the
    view's address, shape, strides and version counter; the entry holds a weak reference to the tensor
    that OWNS the memory (`t._base` for a view), so a fresh view of the same live tensor - the
    generators pass `imgs[:, :-1]` anew for each of the ~144 chunks of an image - hits, and memory that
    was freed and handed out again does not (its owner is gone)."""
    owner = t._base if t._base is not None else t
    key = (kind, t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version, str(t.device))
    hit = _CL_CACHE.get(key)
    if hit is not None and hit[0]() is owner:
        return hit[1]
    out = make(t)
    if len(_CL_CACHE) >= _CL_CACHE_MAX:
        _CL_CACHE.pop(next(iter(_CL_CACHE)))
    try:
        _CL_CACHE[key] = (weakref.ref(owner), out)
    except TypeError:
        pass
    return out


def volume_channels_last(volume_feature):
    """[1,8,D,H,W] -> cached channels-last, depth-innermost [H,W,D,8] device tensor."""
    return _cached("vol", volume_feature, zest_hip.volume_to_cl)


def images_channels_last(imgs):
    """[1,V,3,H,W] -> cached [V,H,W,4] device tensor."""
    return _cached("img", imgs, zest_hip.images_to_cl)


def index_point_feature(volume_feature, ray_coordinate_ref):
    """Trilinear lookup of the 8-channel encoding volume at [N,R,S,3] volume coordinates
    -> [N,R,S,8] (zero padding, align_corners)."""
    if ray_coordinate_ref.shape[0] != 1:
        raise RuntimeError("index_point_feature: batch must be 1 (the reference never uses more)")
    vcl = volume_channels_last(volume_feature)
    return zest_hip.volume_lookup(vcl, ray_coordinate_ref)


def build_color_volume(point_samples, poses, imgs, img_feat=None, downscale=1.0, with_mask=False):
    """Project [N,R,S,3] world points into each of the V source views and gather colours.
    -> [N,R,S,V*3] or, with_mask, [N,R,S,V*4] laid out (r,g,b,mask) per view."""
    if img_feat is not None:
        raise NotImplementedError("build_color_volume: img_feat is dead in the reference "
                                  "(every caller passes None) and is not implemented")
    if point_samples.shape[0] != 1:
        raise RuntimeError("build_color_volume: batch must be 1")
    icl = images_channels_last(imgs)
    out = zest_hip.color_lookup(icl, poses['w2cs'][0], poses['intrinsics'][0], point_samples)
    if with_mask:
        return out
    V = icl.shape[0]
    return out.view(*out.shape[:-1], V, 4)[..., :3].reshape(*out.shape[:-1], 3 * V)


# ---------------------------------------------------------------------------- loss-side
def projection_from_ndc(w2c, H, W, f, weights_ref, raw_pts):
    """Expected 3-D point of every ray (sum_s w_s p_s over NDC points) -> Euclidean -> camera of
    `w2c` -> pixel position [N,R,2] (reference utils.py:507-539).  One HIP
    launch forward, one backward (gradients for the weights and the points)."""
    import zest_autograd
    if weights_ref.shape[0] != 1 or w2c.reshape(-1, 4, 4).shape[0] != 1:
        raise RuntimeError("projection_from_ndc: batch must be 1 (the reference's broadcasting assumes it)")
    out = zest_autograd.ProjectRaysFn.apply(weights_ref[0], raw_pts[0], w2c.reshape(4, 4).contiguous(),
                                            int(H), int(W), float(f))
    return out[None]


# ---------------------------------------------------------------------------- plane sweep
def homo_warp(src_feat, proj_mat, depth_values, src_grid=None, pad=0):
    """Warp a source feature map onto the fronto-parallel planes of the reference view
    (reference utils.py:49-99).  src_feat [1,C,H,W]; proj_mat [1,3,4] =
    src_proj @ ref_proj_inv; depth_values [1,D].  -> (warped [1,C,D,H+2pad,W+2pad],
    src_grid [1,D,W+2pad,H+2pad,2]: the reference's shape label for the normalised sampling
    positions, memory order [D][y][x]).  With src_grid given it is reused, as the reference does
    for the colour images.  Differentiable with respect to src_feat (HIP backward: scatter-add
    through the bilinear taps); the sampling positions are data."""
    if src_feat.shape[0] != 1:
        raise RuntimeError("homo_warp: batch must be 1 (the reference never uses more)")
    if torch.is_grad_enabled() and proj_mat is not None and proj_mat.requires_grad:
        raise NotImplementedError("homo_warp: no gradient with respect to the projection matrices "
                                  "(cameras are data in the reference's training graph)")
    depth = depth_values.reshape(depth_values.shape[0], -1)[0] if src_grid is None else None
    gin = None
    if src_grid is not None:
        D, Wp, Hp = src_grid.shape[1:4]
        gin = src_grid.reshape(D, Hp, Wp, 2)
    if torch.is_grad_enabled() and src_feat.requires_grad:
        import zest_autograd
        warped, grid = zest_autograd.HomoWarpFn.apply(src_feat[0], None if gin is not None else proj_mat[0], depth, gin, pad)
    elif gin is None:
        warped, grid = zest_hip.homo_warp(src_feat[0], proj_mat[0], depth, None, pad)
    else:
        warped, grid = zest_hip.homo_warp(src_feat[0], grid=gin, pad=pad)
    D, Hp, Wp = grid.shape[:3]
    return warped[None], grid.view(1, D, Wp, Hp, 2)


# ---------------------------------------------------------------------------- ray sampling
# The step in front of the renderer: same functions and return tuples
# as the reference (utils.py:133-431).  Pixel selection is R integers and fixes the RNG call
# order, so it stays on the host with the reference's own torch calls (a run seeded like the
# reference picks the same pixels); everything per sample runs in zest_build_rays_fwd.
def _draw_uniform(shape, device):
    """Stratified jitter ~ U[0,1), one draw per call like the reference (utils.py:374).
    Tests replace this hook to inject known numbers."""
    return torch.rand(shape, device=device)


def get_ndc_coordinate(w2c_ref, intrinsic_ref, point_samples, inv_scale, near=2, far=6, pad=0,
                       lindisp=False):
    """World points [N,R,S,3] -> [1,R,S,3] image-plane / depth coordinates of a view, normalised
    by inv_scale and [near, far], with the padded-volume rescale (reference utils.py:232-288)."""
    if intrinsic_ref is None:
        raise NotImplementedError("get_ndc_coordinate: the bounding-box branch is unused by the "
                                  "reference's callers and not implemented")
    R, S = point_samples.shape[1], point_samples.shape[2]
    inv = [float(v) for v in inv_scale.reshape(-1).tolist()]
    out = zest_hip.ndc_coordinate(point_samples.reshape(-1, 3), None if w2c_ref is None else w2c_ref[0],
                                  intrinsic_ref[0], inv[0], inv[1], float(near), float(far), pad, lindisp)
    return out.view(1, R, S, 3)


def sample_pdf(bins, weights, N_samples, det=False):
    """Importance resampling of depth candidates from the coarse pass's compositing weights (the `sample_pdf`
    BASELINE.json's north_star names).  The reference has none (`--N_importance` is parsed, `opt.py:161`, and a
    fine net is built, `train.py:143-148`, but no renderer call receives it): canonical NeRF inverse-CDF
    semantics, HIP kernel zest_sample_pdf_fwd, parity unpinned (tests: numpy inverse CDF + properties).
    bins [..., Nb+1] bin edges (e.g. the midpoints of the coarse depths), weights [..., Nb]; det: evenly spaced
    quantiles instead of torch.rand draws.  -> [..., N_samples] new depths (detached: data for the fine pass)."""
    lead = weights.shape[:-1]
    b2, w2 = bins.detach().reshape(-1, bins.shape[-1]), weights.detach().reshape(-1, weights.shape[-1])
    u = None if det else _draw_uniform((w2.shape[0], N_samples), w2.device)
    return zest_hip.sample_pdf(b2, w2, N_samples, u).view(*lead, N_samples)


def patch_ray_sampler(patch_size, step, random_shift=True, random_scale=True, min_scale=0.25, max_scale=1.,
                      scale_anneal=-1):
    """GRAF's variable patch: a patch_size x patch_size lattice over [-1, 1]^2, randomly scaled (the smallest
    scale annealed towards the whole image with the step) and shifted so it stays inside (reference
    utils.py:102-131).  -> [patch_size, patch_size, 2] grid_sample positions (x, y).  The random draws are the
    reference's own CPU calls in the reference's order (scale; then per axis offset magnitude, offset sign), so a
    run seeded like the reference samples the same patches."""
    import math
    lin = torch.linspace(-1, 1, patch_size)
    rows, cols = lin[:, None, None].expand(-1, patch_size, 1), lin[None, :, None].expand(patch_size, -1, 1)
    if scale_anneal > 0:
        floor = max(min_scale, max_scale * math.exp(-(step // 1000 * 3) * scale_anneal))
        min_scale = min(0.9, floor)
    scale = torch.ones(1)
    if random_scale:
        scale = torch.Tensor(1).uniform_(min_scale, max_scale)
    x, y = cols * scale, rows * scale
    if random_shift:
        room = 1 - scale.item()
        x = x + torch.Tensor(1).uniform_(0, room) * (torch.randint(2, (1,)).float() - 0.5) * 2
        y = y + torch.Tensor(1).uniform_(0, room) * (torch.randint(2, (1,)).float() - 0.5) * 2
    return torch.cat([x, y], dim=2)


def get_rays_mvs(H, W, intrinsic, c2w, N_rays=1024, isRandom=True, chunk=-1, idx=-1, N_patches=None,
                 patch_size=-1, scale_anneal=-1, step=0, variable_patches=False, num_extra_samples=0,
                 motion_coords=None):
    """Ray origin, un-normalised world directions and the (row, col) pixels they pass through
    (reference utils.py:133-230).  -> rays_o [N,3], rays_d [N,R,3], pixel_coordinates [N,2,R]."""
    device = c2w.device
    if variable_patches:
        # GRAF discriminator patches (utils.py:157-170): the lattice positions are turned into pixel indices by
        # sampling the coordinate ramps bilinearly and truncating - done with the very same host ops so the
        # truncation falls where the reference's does
        grid = patch_ray_sampler(patch_size, step, scale_anneal=scale_anneal)[None]
        row_ramp = torch.linspace(0, H - 1, H)[:, None].expand(H, W)[None, None]
        col_ramp = torch.linspace(0, W - 1, W)[None, :].expand(H, W)[None, None]
        gs = torch.nn.functional.grid_sample
        xs = gs(col_ramp, grid, mode="bilinear", align_corners=True).reshape(-1).int().to(device)
        ys = gs(row_ramp, grid, mode="bilinear", align_corners=True).reshape(-1).int().to(device)
    elif N_patches:
        xb, yb = torch.randint(0, W - patch_size, (N_patches,)), torch.randint(0, H - patch_size, (N_patches,))
        ar = torch.arange(patch_size, dtype=torch.float32)
        ys = (yb.float()[:, None, None] + ar[None, :, None]).expand(-1, -1, patch_size).reshape(-1)
        xs = (xb.float()[:, None, None] + ar[None, None, :]).expand(-1, patch_size, -1).reshape(-1)
        if (ys < 0).any() or (ys >= H).any() or (xs < 0).any() or (xs >= W).any():
            raise ValueError("point coordinates out of bounds")
        ys, xs = ys.to(device), xs.to(device)
    elif isRandom:
        xs, ys = torch.randint(0, W, (N_rays,)).float().to(device), torch.randint(0, H, (N_rays,)).float().to(device)
    else:
        lo, hi = (idx * chunk, min((idx + 1) * chunk, H * W)) if chunk > 0 else (0, H * W)
        lin = torch.arange(lo, hi, device=device)
        ys, xs = (lin // W).float(), (lin % W).float()
    xs, ys = xs[None].repeat(intrinsic.shape[0], 1), ys[None].repeat(intrinsic.shape[0], 1)
    if motion_coords is not None and num_extra_samples > 0:
        hard = motion_coords[torch.randint(0, motion_coords.shape[0], (num_extra_samples,))]
        xs = torch.cat([xs, hard[:, 1].unsqueeze(0).to(xs)], dim=1)
        ys = torch.cat([ys, hard[:, 0].unsqueeze(0).to(ys)], dim=1)
    dirs = torch.stack([(xs - intrinsic[:, 0, 2].reshape(-1, 1)) / intrinsic[:, 0, 0].reshape(-1, 1),
                        (ys - intrinsic[:, 1, 2].reshape(-1, 1)) / intrinsic[:, 1, 1].reshape(-1, 1),
                        torch.ones_like(xs)], -1)
    rays_d = torch.matmul(dirs, c2w[:, :3, :3].transpose(1, 2))
    return c2w[:, :3, -1].clone(), rays_d, torch.stack((ys, xs), dim=1)


def build_rays_base(imgs, depths, w2cs, c2ws, intrinsics, near_fars, N_samples, N_rays=1024,
                    stratified=True, pad=0, chunk=-1, idx=-1, ref_idx=0, val=False, isRandom=True,
                    patch_size=-1, scale_anneal=-1, step=0, variable_patches=False, scene_flow=False,
                    flow_fwd=None, flow_bwd=None, mask_fwd=None, mask_bwd=None, num_extra_samples=0,
                    motion_coords=None, zest_rays_only=False):
    """Sample rays of the target (last) view and points along them; same 11-tuple as the
    reference (utils.py:290-394).  zest_rays_only (an addition): leave the ground-truth gathers
    (colour, depth, flow, masks) and t_vals as None - the whole-image loops never read them and
    they are most of the host time of a 1024-ray chunk."""
    device = imgs.device
    N, V, C, H, W = imgs.shape
    if N != 1:
        raise RuntimeError("build_rays_base: image batch must be 1")
    N_patches = None
    if patch_size > 0:
        N_patches = N_rays // (patch_size * patch_size)
        assert N_rays % (patch_size * patch_size) == 0, \
            "Batch size %d is not divisible by patch size of %d" % (N_rays, patch_size)
    _, _, pix = get_rays_mvs(H, W, intrinsics[:, -1], c2ws[:, -1], N_rays, isRandom=isRandom, chunk=chunk,
                             idx=idx, N_patches=N_patches, patch_size=patch_size, scale_anneal=scale_anneal,
                             step=step, variable_patches=variable_patches,
                             num_extra_samples=num_extra_samples, motion_coords=motion_coords)
    ys, xs = pix[0, 0].contiguous(), pix[0, 1].contiguous()
    R = xs.numel()
    color = rays_depth_gt = t_vals = None
    gt = [None, None, None, None]
    if not zest_rays_only:
        yi, xi = ys.long(), xs.long()
        color = imgs[:, -1, :, yi, xi].permute(0, 2, 1)
        rays_depth_gt = depths[:, -1, yi, xi]
        t_vals = torch.linspace(0., 1., steps=N_samples).view(1, N_samples).to(device)
    if scene_flow and not zest_rays_only:
        gt = [flow_fwd[:, -1, :, yi, xi].permute(0, 2, 1), flow_bwd[:, -1, :, yi, xi].permute(0, 2, 1),
              mask_fwd[:, -1, yi, xi], mask_bwd[:, -1, yi, xi]]
    t_rand = _draw_uniform((R, N_samples), device) if stratified else None
    d, z, pts, ndc = zest_hip.build_rays(
        xs, ys, t_rand, N_samples, intrinsics[0, -1], c2ws[0, -1], w2cs[0, ref_idx], intrinsics[0, ref_idx],
        near_fars[0, -1], near_fars[0, ref_idx], pad, W, H)
    return (pts[None], d[None], color, ndc[None], z[None], rays_depth_gt, t_vals, gt[0], gt[1], gt[2], gt[3])


def build_rays(imgs, depths, w2cs, c2ws, intrinsics, near_fars, N_samples, N_rays=1024, stratified=True,
               pad=0, chunk=-1, idx=-1, ref_idx=0, val=False, isRandom=True, patch_size=-1,
               scale_anneal=-1, step=0, variable_patches=False):
    """Reference utils.py:396-407."""
    return build_rays_base(imgs, depths, w2cs, c2ws, intrinsics, near_fars, N_samples, N_rays=N_rays,
                           stratified=stratified, pad=pad, chunk=chunk, idx=idx, ref_idx=ref_idx, val=val,
                           isRandom=isRandom, patch_size=patch_size, scale_anneal=scale_anneal, step=step,
                           variable_patches=variable_patches, scene_flow=False)[:7]


def build_rays_dy(imgs, depths, w2cs, c2ws, intrinsics, near_fars, N_samples, N_rays=1024, stratified=True,
                  pad=0, chunk=-1, idx=-1, ref_idx=0, val=False, isRandom=True, patch_size=-1,
                  scale_anneal=-1, step=0, variable_patches=False, scene_flow=False, flow_fwd=None,
                  flow_bwd=None, mask_fwd=None, mask_bwd=None, num_extra_samples=0, motion_coords=None,
                  zest_rays_only=False):
    """Reference utils.py:409-431."""
    return build_rays_base(imgs, depths, w2cs, c2ws, intrinsics, near_fars, N_samples, N_rays=N_rays,
                           stratified=stratified, pad=pad, chunk=chunk, idx=idx, ref_idx=ref_idx, val=val,
                           isRandom=isRandom, patch_size=patch_size, scale_anneal=scale_anneal, step=step,
                           variable_patches=variable_patches, scene_flow=scene_flow, flow_fwd=flow_fwd,
                           flow_bwd=flow_bwd, mask_fwd=mask_fwd, mask_bwd=mask_bwd,
                           num_extra_samples=num_extra_samples, motion_coords=motion_coords,
                           zest_rays_only=zest_rays_only)
