/*
 * zest_render.h — C ABI of the MI355X (gfx950) ZeST-NeRF volume-rendering library.
 *
 * The reference (violetamenendez/zest-nerf) has no FFI: its hot path is bound by
 * Python import (`from renderer import rendering`, train.py:44; `from utils import
 * ...`, renderer.py:20; networks.py:25-26).  This header is the drop-in boundary
 * underneath that Python surface: each entry point replaces the device work of one
 * reference function, takes plain device pointers + sizes + a HIP stream, allocates
 * nothing the caller sees, never synchronises, and returns a hipError_t as int
 * (0 = success).  `zest_last_error()` gives the message for the calling thread.
 *
 * All tensors are dense row-major fp32 unless stated; R = rays, S = samples per ray,
 * M = R*S flattened samples, V = source views.  Inputs are never written.
 * An empty batch (R == 0 or M == 0) is a successful no-op for every per-ray / per-sample
 * entry point, whatever the pointers (an empty tensor has no storage); every other
 * argument the kernels do not cover is refused with a message, before any launch.
 * INTEGRATION.md shows the ctypes binding (zest-nerf_amd/zest_hip.py) a maintainer of
 * the reference would add.
 *
 * Devices.  Every entry point works on the CURRENT HIP device of the calling thread (hipSetDevice /
 * torch.cuda.device): all pointers of a call, and the stream, must belong to it.  The library's own device-side
 * state - the gather tables behind zest_mlp_pack, the job and position tables of the zest_mlp_train16_* path, the
 * rocBLAS handle of the fp32 training path - is kept per device (keyed on the device that is current at the call),
 * so one process may render and train on several devices, one current device per call.  A packed weight stream
 * (zest_mlp_pack, zest_mlp_train16_pack) lives in the caller's buffer on the device it was packed on and is valid
 * there only.
 */
#ifndef ZEST_RENDER_H
#define ZEST_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZEST_ABI_VERSION 3

/* arithmetic of the MLP contraction */
#define ZEST_PREC_F32  0   /* v_mfma_f32_32x32x2_f32: exact fp32 products, parity mode */
#define ZEST_PREC_BF16 1   /* v_mfma_f32_16x16x32_bf16: bf16 operands, fp32 accumulate  */
#define ZEST_PREC_F16  2   /* v_mfma_f32_16x16x32_f16: fp16 operands (11-bit significand), */
                           /* fp32 accumulate; activations beyond 65504 saturate          */
#define ZEST_PREC_F16X3 3  /* fp32-class results on the fp16 matrix pipe: every operand is */
                           /* the pair (hi, lo) = (fp16(v), fp16((v - hi) * 2^11)) and a  */
                           /* product is hi*hi + 2^-11 (hi*lo + lo*hi): 22 significant    */
                           /* bits per operand, 3 MFMAs per product, fp32 accumulate       */

/* extra heads of the MLP (reference networks.py:115-123) */
#define ZEST_HEAD_NONE    0  /* out = rgb(3) sigma(1)                                     */
#define ZEST_HEAD_BLEND   1  /* static net with scene flow: + sigmoid(w_linear)   -> 5   */
#define ZEST_HEAD_DYNAMIC 2  /* dynamic net: + tanh(sf_linear)(6) sigmoid(prob)(2) -> 12 */

/* Shape of one NeRF MLP (reference networks.py:73-132 `Renderer`, :223-265 `Renderer_linear`).
 * depth / width / skip_mask all zero = the shape every shipped config uses (D=8, W=256, skips=[4]),
 * the only one the MFMA engine (bf16 / fp16 precisions, the fused renderer, the train16 path) covers;
 * other shapes (reference opt.py:54-57 --netdepth / --netwidth, networks.py:93-100) run on the fp32
 * kernel (zest_mlp_fwd with ZEST_PREC_F32) and the fp32 training path (zest_mlp_train_*). */
typedef struct zest_mlp_desc {
    int32_t in_ch_pts;    /* encoded point width: 63 (xyz, L=10) or 84 (xyzt)            */
    int32_t in_ch_feat;   /* F = 8 + 4V per-sample feature width (ignored if !use_feat)   */
    int32_t in_ch_views;  /* encoded direction width: 27                                  */
    int32_t use_feat;     /* 1: input carries features, trunk is modulated by pts_bias    */
    int32_t net_type;     /* 0 = 'v0' multiplicative modulation; 2 = 'v2' additive,       */
                          /*     relu(alpha), sigmoid(rgb); 3 = the 'v2' trunk with raw   */
                          /*     outputs (Renderer_linear.forward_alpha; zest_mlp_fwd only) */
    int32_t head;         /* ZEST_HEAD_*                                                   */
    int32_t depth;        /* D: trunk layers, 2..8 (0 with width = skip_mask = 0: the default shape) */
    int32_t width;        /* W: trunk width, 64 | 128 | 192 | 256                             */
    int32_t skip_mask;    /* bit i set: layer i+1 takes [pts | h] (reference `skips` holds i)  */
} zest_mlp_desc;

/* Parameter order for zest_mlp_pack: weight then bias of each nn.Linear, [out,in]
 * row-major fp32 exactly as in the reference state dict. */
enum {
    ZEST_P_PTS0 = 0,          /* pts_linears.0 .. pts_linears.(D-1) -> 0..D-1 */
    ZEST_P_PTS_BIAS = 8,      /* pts_bias        [W, F]                */
    ZEST_P_VIEWS = 9,         /* views_linears.0 [W/2, W+27]           */
    ZEST_P_FEATURE = 10,      /* feature_linear  [W, W]                */
    ZEST_P_ALPHA = 11,        /* alpha_linear    [1, W]                */
    ZEST_P_RGB = 12,          /* rgb_linear      [3, W/2]              */
    ZEST_P_HEAD0 = 13,        /* w_linear [1,W]    | sf_linear [6,W]   */
    ZEST_P_HEAD1 = 14,        /* (unused)          | prob_linear [2,W] */
    ZEST_P_COUNT = 15
};

int         zest_abi_version(void);
const char *zest_last_error(void);
/* number of compute units / multiprocessor clock (kHz) of the current device */
int         zest_device_info(int *cu_count, int *clock_khz, char *name, size_t name_len);

/* ---- compositing -------------------------------------------------------------
 * Replaces depth2dist + raw2outputs + raw2alpha (reference renderer.py:74-164).
 * raw [R,S,4] (rgb logits, sigma), z [R,S], rays_dir [R,3] (un-normalised; the
 * sample distance is dz * |dir|, last sample 1e10*|dir|).  dists [R,S] or NULL: when given,
 * the sample distances are taken from it instead (the `dists` argument of the reference's
 * raw2outputs, whatever produced it) and rays_dir may be NULL.  noise [R,S] or NULL is
 * added to sigma after scaling by noise_std.  Any output pointer may be NULL.
 * rgb_map [R,3], depth/acc/disp [R], weights/alpha [R,S]. */
int zest_composite_fwd(const float *raw, const float *z, const float *rays_dir, const float *dists,
                       const float *noise, float noise_std, int white_bkgd,
                       int R, int S,
                       float *rgb_map, float *depth_map, float *acc_map, float *disp_map,
                       float *weights, float *alpha, void *stream);

/* Replaces raw2outputs_blending (reference renderer.py:166-219).
 * raw_dy, raw_st [R,S,4]; blend [R,S].  Outputs: blended rgb_map [R,3], depth_map [R];
 * dynamic-only rgb_map_fg [R,3], depth_map_fg [R], weights_fg [R,S]; weights_dy [R,S]
 * (dynamic share of the blended weights) and its per-ray sum weights_dd_sum [R]. */
int zest_composite_blend_fwd(const float *raw_dy, const float *raw_st, const float *blend,
                             const float *z, const float *rays_dir, const float *dists,
                             const float *noise, float noise_std, int R, int S,
                             float *rgb_map, float *depth_map, float *rgb_map_fg,
                             float *depth_map_fg, float *weights_fg, float *weights_dy,
                             float *weights_dd_sum, void *stream);

/* Weighted per-ray sum  out[r] = sum_s w[r,s] * (1 - p[r,s])
 * (compute_2d_prob, reference renderer.py:22-32). */
int zest_weighted_complement_sum(const float *w, const float *p, int R, int S, float *out,
                                 void *stream);

/* ---- per-sample operators -------------------------------------------------------
 * Positional encoding, Embedding.forward (reference networks.py:48-65), log-scale
 * bands 2^0..2^(L-1):  x [M,C] -> y [M, C*(2L+1)]. */
int zest_embed_fwd(const float *x, int M, int C, int L, float *y, void *stream);

/* One-off layout changes made when a volume / image set is first seen:
 * volume [8,D,H,W] -> channels-last, depth innermost [H,W,D,8] (one 32-byte read per trilinear corner; the corners of
 * consecutive samples of a ray are contiguous runs);
 * imgs [V,3,H,W] -> [V,H,W,4] (rgb + pad, one 16-byte read per bilinear corner). */
int zest_volume_to_cl(const float *vol, int D, int H, int W, float *vol_cl, void *stream);
int zest_images_to_cl(const float *imgs, int V, int H, int W, float *imgs_cl, void *stream);

/* ---- MVS volume builder: plane-sweep cost volume --------------------
 * zest_nchw_to_nhwc: [N,C,H,W] -> [N,H,W,C] (feature maps of FeatureNet, reference
 * networks.py:962-1001, to channels-last: one bilinear tap = C contiguous floats).
 *
 * zest_volume_cost_fwd: MVSNet.build_volume_cost (reference networks.py:1077-1140), batch 1.
 *   feats_cl [V,H,W,32], imgs_cl [V,H,W,4] (images already resized to H x W, rgb + pad),
 *   proj [V-1,3,4] = src_proj @ ref_proj_inv of source views 1..V-1, depth [D] plane depths.
 *   -> img_feat [3V+32, D, H+2pad, W+2pad]: reference image, warped source images, variance of
 *   the (warped) features over the views whose projection is in frame; in_masks
 *   [V, D, H+2pad, W+2pad].  The reference leaves channels 0-2 of the padding ring
 *   uninitialised (torch.empty, networks.py:1097-1099); they are written as 0 here.
 *
 * zest_homo_warp_fwd: utils.homo_warp (reference utils.py:49-99).  src [C,H,W]; either
 *   proj [3,4] + depth [D] (grid_out [D,Hp,Wp,2] receives the normalised sampling positions)
 *   or grid_in [D,Hp,Wp,2] from an earlier call; warped [C,D,Hp,Wp]. */
int zest_nchw_to_nhwc(const float *in, int N, int C, int H, int W, float *out, void *stream);
int zest_volume_cost_fwd(const float *feats_cl, const float *imgs_cl, const float *proj,
                         const float *depth, int V, int C, int D, int H, int W, int pad,
                         float *img_feat, float *in_masks, void *stream);
int zest_homo_warp_fwd(const float *src, const float *proj, const float *depth, const float *grid_in,
                       int C, int D, int H, int W, int Hp, int Wp, int pad, float *warped,
                       float *grid_out, void *stream);

/* ---- MVS volume builder: 3-D regularisation net without autograd ----------
 * CostRegNet (reference networks.py:1003-1059) on channels-last fp32 tensors [D,H,W,C]; a layer's batch norm +
 * leaky ReLU(0.01) (InPlaceABN, networks.py:938-960) is applied by the layer that READS its output, from two
 * constants per channel (pre [2,C]: scale, shift).
 * zest_volume_cost_cl_fwd: zest_volume_cost_fwd's img_feat as cost_cl [D, H+2pad, W+2pad, 48] (the 3V+32 channels
 *   of a voxel, then zeros; V <= 5), no masks.
 * zest_costreg_conv_fwd: Conv3d(cin -> cout, 3, stride, padding 1, no bias) on act(norm(in)) (pre NULL: on `in`
 *   itself - first layer only).  w_packed: zest_costreg_packed_bytes(cin, cout, passes) bytes in the MFMA operand
 *   order (zest_networks.CostRegNet packs them).  passes 1: bf16 operands; 3: split bf16 pairs (16 significant
 *   bits).  out [Do,Ho,Wo,cout] raw; stats [zest_costreg_stat_rows(), 2, cout] doubles: every workgroup writes the
 *   sum and sum of squares of its voxels to a row of its own (the last row holds the number of rows in use), and
 *   zest_costreg_bn adds the rows in a fixed order - the result does not depend on the order workgroups finish in.
 *   Shapes: the seven layers of CostRegNet (48->8/1, 8->16/2, 16->16/1, 16->32/2, 32->32/1, 32->64/2, 64->64/1).
 * zest_costreg_deconv_fwd: ConvTranspose3d(cin -> cout, 3, stride 2, padding 1, output_padding 1, no bias) on
 *   act(norm(in0)) [+ act(norm(in1))]; w_packed: zest_costreg_deconv_packed_bytes(cin, cout, passes) bytes, the
 *   taps of the eight output parity classes in the MFMA operand order (zest_networks.CostRegNet packs them);
 *   out [2Di,2Hi,2Wi,cout] raw; stats and passes as above.  Shapes: 64->32, 32->16, 16->8.
 * zest_costreg_bn: pre [2,C] (C <= 64) from the table of batch statistics of `count` voxels (batch_stats != 0; running_mean /
 *   running_var / steps, when given, are updated as nn.BatchNorm does in training mode) or from the running ones;
 *   moments [2,C] or NULL receives mean and 1/sqrt(var + eps) (what a batch-norm backward needs).
 * zest_costreg_out: encoding volume [8,D,H,W] = act(norm(raw_a)) + act(norm(raw_b)) from two [D,H,W,8] tensors.
 * zest_conv2d_fwd: the same kernel on a batch of N images [N,H,W,cin] channels-last - Conv2d(cin -> cout, k, stride,
 *   padding k/2, no bias) on act(norm(in)) (pre NULL: on `in` itself - first layer only), the layers of FeatureNet
 *   (reference networks.py:962-1001): 8->8 k3 (3 input channels padded to 8), 8->16 k5/2, 16->16 k3, 16->32 k5/2,
 *   32->32 k3.  w_packed: zest_conv2d_packed_bytes(cin, cout, k, passes) bytes; out, stats, passes as above. */
int zest_costreg_stat_rows(void);
size_t zest_conv2d_packed_bytes(int cin, int cout, int k, int passes);
int zest_conv2d_fwd(const float *in, const float *pre, const void *w_packed, int cin, int cout, int k, int stride,
                    int passes, int N, int Hi, int Wi, float *out, double *stats, void *stream);
size_t zest_costreg_packed_bytes(int cin, int cout, int passes);
int zest_volume_cost_cl_fwd(const float *feats_cl, const float *imgs_cl, const float *proj, const float *depth,
                            int V, int C, int D, int H, int W, int pad, float *cost_cl, void *stream);
int zest_costreg_conv_fwd(const float *in, const float *pre, const void *w_packed, int cin, int cout, int stride,
                          int passes, int Di, int Hi, int Wi, float *out, double *stats, void *stream);
size_t zest_costreg_deconv_packed_bytes(int cin, int cout, int passes);
int zest_costreg_deconv_fwd(const float *in0, const float *pre0, const float *in1, const float *pre1,
                            const void *w_packed, int cin, int cout, int passes, int Di, int Hi, int Wi,
                            float *out, double *stats, void *stream);
int zest_costreg_bn(const double *stats, int C, long long count, const float *gamma, const float *beta, float eps,
                    int batch_stats, float *running_mean, float *running_var, float momentum,
                    long long *steps, float *pre, float *moments, void *stream);
/* Backward of act(norm(raw)) with batch statistics (training; zest_autograd.CostRegFn): raw, g_act, g_raw [M,C]
 * channels-last (C = 8, 16, 32, 64); pre / moments [2,C] of the forward (zest_costreg_bn); stats: a table as above
 * (workspace); totals [2,C] receives g_beta = sum g_y and g_gamma = sum g_y xhat. */
int zest_costreg_bn_bwd(const float *raw, const float *g_act, const float *pre, const float *moments,
                        const float *gamma, int C, long long M, double *stats, float *totals, float *g_raw,
                        void *stream);
int zest_costreg_out(const float *raw_a, const float *pre_a, const float *raw_b, const float *pre_b, int D,
                     int H, int W, float *volume, void *stream);

/* Backward of the plane sweep (the MVSNet trains through it: reference train.py:270 optimises
 * generator.parameters(), networks.py:1077-1140 runs under autograd).  The trainable input is the
 * feature maps; images, homographies, depths and the in-frame counts carry no gradient.
 * zest_volume_cost_bwd: g_img_feat [3V+32, D, Hp, Wp] (gradient of zest_volume_cost_fwd's img_feat;
 *   only the 32 variance channels are read) -> g_feats_cl [V,H,W,32], ACCUMULATED with float atomics
 *   (zero it first).  The warped features are gathered again from feats_cl, not stored.
 * zest_homo_warp_bwd: g_warped [C,D,Hp,Wp] -> g_src [C,H,W], accumulated (zero it first); grid_in or
 *   proj + depth as in the forward call. */
int zest_volume_cost_bwd(const float *feats_cl, const float *proj, const float *depth, int V, int C,
                         int D, int H, int W, int pad, const float *g_img_feat, float *g_feats_cl,
                         void *stream);
int zest_homo_warp_bwd(const float *proj, const float *depth, const float *grid_in, int C, int D, int H,
                       int W, int Hp, int Wp, int pad, const float *g_warped, float *g_src, void *stream);

/* ---- loss-side reductions over the samples of a ray -------------------
 * zest_distortion_fwd: distortion_loss (reference losses.py:53-87) per ray.  weights [R,S];
 *   t_vals [t_rows,S] with t_rows 1 or R.  loss_ray [R] (the reference returns their sum);
 *   grad_w [R,S] or NULL receives d loss_ray / d weights (t_vals carry no gradient).
 * zest_project_rays_fwd/bwd: projection_from_ndc (reference utils.py:507-539).  weights [R,S],
 *   pts [R,S,3] NDC points, w2c [>=12] row-major rows of (R | t), image size H x W, focal.
 *   -> out [R,2] pixel positions; bwd: grad_out [R,2] -> d_weights [R,S], d_pts [R,S,3]
 *   (either may be NULL). */
int zest_distortion_fwd(const float *weights, const float *t_vals, int t_rows, int R, int S,
                        float *loss_ray, float *grad_w, void *stream);
int zest_project_rays_fwd(const float *weights, const float *pts, const float *w2c, int H, int W,
                          float focal, int R, int S, float *out, void *stream);
int zest_project_rays_bwd(const float *weights, const float *pts, const float *w2c, int H, int W,
                          float focal, const float *grad_out, int R, int S, float *d_weights,
                          float *d_pts, void *stream);

/* Trilinear lookup, zero padding, align_corners: index_point_feature
 * (reference utils.py:433-459).  vol_cl [H,W,D,8]; ndc [M,3] -> out [M,8]. */
int zest_volume_lookup_fwd(const float *vol_cl, int D, int H, int W, const float *ndc, int M,
                           float *out, void *stream);

/* Per-view projection + bilinear colour (border clamp) + strict in-frame mask:
 * build_color_volume(with_mask=True) + get_ndc_coordinate (reference utils.py:461-505,
 * 262-269).  imgs_cl [V,H,W,4]; w2cs [>=V,4,4]; intrinsics [>=V,3,3]; pts [M,3] world
 * -> out [M,4V] as (r,g,b,mask) per view. */
int zest_color_lookup_fwd(const float *imgs_cl, int V, int H, int W, const float *w2cs,
                          const float *intrinsics, const float *pts, int M, float *out,
                          void *stream);

/* Assemble the MLP input exactly as prepare_pts / prepare_dynamic_pts do (reference
 * renderer.py:246-318): x[m] = PE_L10(ndc[,t]) | vol(8) | colours(4V) | PE_L4(dir_ref).
 * ndc, pts [R,S,3]; rays_dir [R,3]; w2cs/intrinsics of the view set (view 0 rotates the
 * direction, renderer.py:256-258); has_time appends the constant frame index t as 4th
 * coordinate; vol_cl/imgs_cl NULL => no feature columns; w2cs NULL => direction is not
 * rotated.  x [R*S, C_in]. */
int zest_encode_fwd(const float *ndc, const float *pts, const float *rays_dir, int R, int S,
                    int has_time, float t,
                    const float *vol_cl, int D, int Hv, int Wv,
                    const float *imgs_cl, int V, int H, int W,
                    const float *w2cs, const float *intrinsics,
                    float *x, void *stream);
/* The same entry point under the name SURVEY.md 8(b) gives it (the survey lists the launchers a replacement exports
 * as zest_gather_encode_fwd and zest_pack_weights; the reference itself has no FFI that would bind either name). */
int zest_gather_encode_fwd(const float *ndc, const float *pts, const float *rays_dir, int R, int S,
                           int has_time, float t,
                           const float *vol_cl, int D, int Hv, int Wv,
                           const float *imgs_cl, int V, int H, int W,
                           const float *w2cs, const float *intrinsics,
                           float *x, void *stream);

/* ---- ray sampling (the step in front of the renderer) -----------------------------
 * Per-sample part of build_rays_base (reference utils.py:361-387): depth candidates
 * near*(1-t)+far*t over t = linspace(0,1,S), jittered inside their strata when t_rand [R,S]
 * is given (the reference's torch.rand draw), points o + z d along the target camera's rays
 * through pixels (xs, ys) [R], and their coordinates in the reference view's padded volume
 * (get_ndc_coordinate, utils.py:232-288).  Camera matrices (row-major 3x3 / 4x4) and the
 * (near, far) pairs are DEVICE pointers into the batch's camera tensors: no host round trip.
 * Outputs: rays_dir [R,3], depth [R,S], pts [R,S,3], ndc [R,S,3]. */
int zest_build_rays_fwd(const float *xs, const float *ys, const float *t_rand, int R, int S,
                        const float *k_tgt, const float *c2w_tgt, const float *w2c_ref,
                        const float *k_ref, const float *near_far_tgt, const float *near_far_ref,
                        int pad, int W, int H, float *rays_dir, float *depth, float *pts, float *ndc,
                        void *stream);

/* Hierarchical resampling (`sample_pdf` of BASELINE.json's north_star; the reference itself has no such
 * function - a build extension, canonical NeRF inverse-CDF semantics, parity unpinned).  bins [R, n_bins + 1]
 * ascending bin edges, weights [R, n_bins] >= 0, u [R, n_samples] in [0, 1) or NULL (deterministic:
 * linspace(0, 1, n_samples)) -> samples [R, n_samples]: ascending when u is, inside [bins[0], bins[n_bins]]. */
int zest_sample_pdf_fwd(const float *bins, const float *weights, const float *u, int R, int n_bins,
                        int n_samples, float *samples, void *stream);

/* get_ndc_coordinate (reference utils.py:232-288): world pts [M,3] -> (u, v, z) normalised by
 * inv_scale = (inv_w, inv_h) and [near, far] (or inverse depth when lindisp), with the padded
 * feature-map rescale when pad > 0.  w2c [4,4] (device) may be NULL (points already in the
 * camera frame); k [3,3] device. */
int zest_ndc_fwd(const float *pts, int M, const float *w2c, const float *k, float inv_w,
                 float inv_h, float near, float far, int pad, int lindisp, float *out, void *stream);

/* ---- compositing backward (training path) -----------------------------------------
 * Gradient of zest_composite_fwd with respect to raw, given the gradients of the outputs
 * the reference's losses consume: rgb_map [R,3], depth_map [R], acc_map [R], weights [R,S]
 * (any may be NULL = zero).  disp_map and alpha have no consumer and carry no gradient.
 * Same raw / z / rays_dir / dists / noise as the forward call.  g_raw [R,S,4]. */
int zest_composite_bwd(const float *raw, const float *z, const float *rays_dir, const float *dists,
                       const float *noise, float noise_std, int white_bkgd, int R, int S, const float *g_rgb_map,
                       const float *g_depth_map, const float *g_acc_map, const float *g_weights,
                       float *g_raw, void *stream);

/* Gradient of zest_composite_blend_fwd: given g of rgb_map, depth_map, rgb_map_fg,
 * depth_map_fg, weights_fg, weights_dy (NULL = zero; weights_dd_sum is detached in the
 * reference) -> g_raw_dy, g_raw_st [R,S,4], g_blend [R,S]. */
int zest_composite_blend_bwd(const float *raw_dy, const float *raw_st, const float *blend,
                             const float *z, const float *rays_dir, const float *dists,
                             const float *noise, float noise_std, int R, int S, const float *g_rgb_map,
                             const float *g_depth_map, const float *g_rgb_map_fg,
                             const float *g_depth_map_fg, const float *g_weights_fg,
                             const float *g_weights_dy, float *g_raw_dy, float *g_raw_st,
                             float *g_blend, void *stream);

/* Gradient of zest_encode_fwd for the inputs that carry gradients in the reference's training
 * graph: g_x [R*S, C_in] -> g_ndc [R,S,3] (through the positional encoding and the trilinear
 * lookup) and, if g_vol_cl is given, += the channels-last volume gradient [Hv,Wv,D,8]
 * (atomic scatter-add; zero it first).  vol_cl / V as in the forward call (NULL / 0: no
 * feature columns).  zest_volume_from_cl converts that gradient back to [8,D,H,W]. */
int zest_encode_bwd(const float *g_x, const float *ndc, int R, int S, int has_time, float t,
                    const float *vol_cl, int D, int Hv, int Wv, int V, float *g_ndc, float *g_vol_cl,
                    void *stream);
int zest_volume_from_cl(const float *vol_cl, int D, int H, int W, float *vol, void *stream);

/* ---- MLP training path --------------------------------------------------------------
 * fp32 forward that keeps what backward needs, and the backward: gradients with respect to
 * the input x (point-encoding and feature columns; the direction columns are data) and to
 * every parameter.  params / g_params: 2*ZEST_P_COUNT device pointers (weight, bias per
 * ZEST_P_* slot, same layout as zest_mlp_pack; g_params entries are overwritten).
 * saved / workspace: caller-owned scratch of zest_mlp_train_saved_floats /
 * zest_mlp_train_workspace_floats floats; `saved` and `out` of the forward call must be handed
 * unchanged to the backward call.  GEMMs run in rocBLAS sgemm (plain library shapes). */
size_t zest_mlp_train_saved_floats(const zest_mlp_desc *desc, int M);
size_t zest_mlp_train_workspace_floats(const zest_mlp_desc *desc, int M);
int zest_mlp_train_fwd(const zest_mlp_desc *desc, const float *const *params, const float *x, int M,
                       float *saved, float *workspace, float *out, void *stream);
int zest_mlp_train_bwd(const zest_mlp_desc *desc, const float *const *params, const float *x, int M,
                       const float *saved, const float *out, const float *g_out, float *workspace,
                       float *g_x, float *const *g_params, void *stream);

/* ---- MLP training path, bf16 operands on the MFMA engine (hand-written forward AND backward) ----
 * The fast training mode ('v0' nets).  Forward: the inference engine kernel, which additionally
 * stashes every layer's output tiles and the encoder's operand tiles (bf16, 5.2 KB per sample) and
 * ReLU masks.  Backward: a data kernel that walks the transposed weight stream on the same engine
 * (gradient tiles to `work`, point-encoding gradients into g_x), a modulation kernel (d pts_bias input
 * = feature columns of g_x) and a weight-gradient kernel that contracts the stash tiles over the
 * samples with transposing LDS reads (per-workgroup partial sums to `work`, then a reduce kernel);
 * fp32 accumulation everywhere, fp32 gradients out, bit-reproducible from run to run.
 *   packed_fwd: zest_mlp_pack(..., ZEST_PREC_BF16); packed_bwd: zest_mlp_train16_pack (same params).
 *   stash / work: caller-owned, zest_mlp_train16_stash_bytes / _work_bytes; the forward call's stash
 *   and out go unchanged to the backward call.  g_x [M, C_in] and every g_params[i] must be ZERO on
 *   entry (gradients are added to them); the direction columns of g_x stay zero.
 *   stages: bit 0 data kernel, bit 1 modulation kernel, bit 2 weight kernel (7 = all; tests run them apart). */
size_t zest_mlp_train16_stash_bytes(const zest_mlp_desc *desc, int M);
size_t zest_mlp_train16_work_bytes(const zest_mlp_desc *desc, int M);
size_t zest_mlp_train16_packed_bytes(const zest_mlp_desc *desc);
int zest_mlp_train16_pack(const zest_mlp_desc *desc, const float *const *params, void *packed_bwd, void *stream);
int zest_mlp_train16_fwd(const zest_mlp_desc *desc, const void *packed_fwd, const float *x, int M, void *stash,
                         float *out, void *stream);
int zest_mlp_train16_bwd(const zest_mlp_desc *desc, const void *packed_bwd, const float *const *params,
                         const float *x, int M, const void *stash, const float *out, const float *g_out,
                         void *work, float *g_x, float *const *g_params, int stages, void *stream);

/* ---- MLP ---------------------------------------------------------------------------
 * Weights are re-packed once per parameter update into the order the MFMA engine
 * streams them.  zest_mlp_packed_bytes gives the buffer size; params is an array of
 * 2*ZEST_P_COUNT device pointers (weight, bias per ZEST_P_* slot, NULL where absent). */
size_t zest_mlp_packed_bytes(const zest_mlp_desc *desc, int precision);
int    zest_mlp_pack(const zest_mlp_desc *desc, int precision, const float *const *params,
                     void *packed, void *stream);
/* zest_mlp_pack under SURVEY.md 8(b)'s name */
int    zest_pack_weights(const zest_mlp_desc *desc, int precision, const float *const *params,
                     void *packed, void *stream);

/* MVSNeRF.forward / Renderer.forward (reference networks.py:150-221, 283-319):
 * x [M, C_in] -> out [M, C_out], C_out = 4 | 5 | 12 by desc->head. */
int zest_mlp_fwd(const zest_mlp_desc *desc, int precision, const void *packed,
                 const float *x, int M, float *out, void *stream);

/* ---- fused inference path ---------------------------------------------------------
 * rendering(..., val=True) (reference renderer.py:579-626 with the early return at
 * :444-445): encode + feature gathers + static MLP [+ dynamic MLP] + per-block compositing
 * in one launch, nothing per-sample written to HBM.  precision: ZEST_PREC_BF16, _F16 or _F16X3
 * (the packed weights must have been packed for it); _F16X3 is the fp32 mode of this path
 * (results within 1e-4 abs / 1e-3 rel of the fp32 reference).  A ray is cut into blocks of 32
 * samples (16 for _F16X3), one per wave of an 8-wave workgroup pass.  The blocks of a ray are
 * chained inside the launch: a workgroup owns a range of whole rays and carries a ray that spans
 * two of its passes in LDS (any S); only where that would need more rounds of passes than spreading
 * all blocks over the device (a few long rays) are the blocks chained by a second tiny launch reading
 * the block records from `workspace` (see zest_render_fused_pass_shape).
 * out [R,16]: 0-2 rgb_map, 3 depth_map, 4 acc_map; with the dynamic net also
 * 5-7 rgb_map_ref, 8 depth_map_ref, 9-11 rgb_map_ref_dy, 12 depth_map_ref_dy,
 * 13 weights_map_dd; 14,15 reserved. */
typedef struct zest_view_set {
    const float *vol_cl;      /* [Hv,Wv,D,8] or NULL */
    int32_t D, Hv, Wv;
    const float *imgs_cl;     /* [V,H,W,4] or NULL   */
    int32_t V, H, W;
    const float *w2cs;        /* [>=max(V,1),4,4] or NULL */
    const float *intrinsics;  /* [>=V,3,3]           */
} zest_view_set;

/* bytes of caller-owned scratch zest_render_fused_fwd needs for R rays of S samples
 * (one 80-byte record per block of 16 samples, the smallest block of any precision) */
size_t zest_render_fused_workspace(int R, int S);

/* Pass shape of zest_render_fused_fwd for the whole process: 0 = chosen per launch (default),
 * 1 = dense (equal shares of all blocks per workgroup + the combine launch), 2 = ray ranges.
 * Results are identical; a knob for tests and measurements. */
int zest_render_fused_set_passes(int shape);
/* What zest_render_fused_fwd will do for R rays of S samples on a device of `cus` compute units (0: the
 * current device): *ray_ranges = 1: every workgroup owns a range of whole rays, walks their blocks 8 at
 * a time and finishes the rays in the kernel, carrying a ray that spans two passes in LDS (the default
 * wherever it needs no more rounds than the dense shape); 0 = dense + the combine launch.
 * *n_wg (may be NULL) = workgroups launched, *rounds (may be NULL) = passes of the busiest workgroup.
 * Host arithmetic only. */
int zest_render_fused_pass_shape(int R, int S, int precision, int cus, int *ray_ranges, int *n_wg, int *rounds);

int zest_render_fused_fwd(const float *ndc, const float *pts, const float *z,
                          const float *rays_dir, int R, int S,
                          const zest_mlp_desc *desc_static, const void *packed_static,
                          const zest_view_set *views_static,
                          const zest_mlp_desc *desc_dynamic, const void *packed_dynamic,
                          const zest_view_set *views_dynamic, float frame_idx,
                          int precision, int white_bkgd, void *workspace, float *out,
                          void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ZEST_RENDER_H */
