// Loss-side reductions over the samples of a ray: the two places where the
// reference's training step re-reads per-sample renderer outputs and builds [R,S,S] / [R,S,3]
// temporaries for a per-ray scalar or 2-vector.
//   zest_distortion_*     distortion_loss (reference losses.py:53-87): O(S^2) pair sum per ray,
//                         here in registers with the ray's (w, midpoint) broadcast through LDS
//   zest_project_rays_*   projection_from_ndc (reference utils.py:507-539): expected 3-D point
//                         sum_s w_s p_s -> NDC2Euclidean -> rigid transform -> pinhole projection
// One wave per ray; both forward kernels can emit what the backward needs.
#include "zest_common.cuh"
#include "../../include/zest_render.h"

namespace {

constexpr int kWaves = 4;            // rays per workgroup
constexpr int kMaxS = 1024;          // samples per ray the LDS staging covers

// loss_r = 1/2 sum_{i,j<S-1} w_i w_j |m_i - m_j| + 1/3 sum_{i<S-1} w_i^2 (t_{i+1} - t_i),
// m_i = (t_i + t_{i+1}) / 2;  dloss_r/dw_i = sum_j w_j |m_i - m_j| + 2/3 w_i (t_{i+1} - t_i)
__global__ __launch_bounds__(kWaves * 64) void distortion_kernel(
    const float *__restrict__ w, const float *__restrict__ t, int t_rows, int R, int S,
    float *__restrict__ loss_ray, float *__restrict__ grad_w) {
    __shared__ float2 wm[kWaves][kMaxS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_raw = blockIdx.x * kWaves + wave;
    const bool live = r_raw < R;
    const int r = live ? r_raw : R - 1;
    const float *wr = w + (size_t)r * S, *tr = t + (size_t)(t_rows == 1 ? 0 : r) * S;
    const int n = S - 1;                                        // intervals
    for (int i = lane; i < n; i += 64) wm[wave][i] = make_float2(wr[i], 0.5f * (tr[i] + tr[i + 1]));
    __syncthreads();
    if (!live) return;
    float acc = 0.0f;
    for (int i = lane; i < n; i += 64) {
        const float2 me = wm[wave][i];
        float a = 0.0f;
        for (int j = 0; j < n; j++) {
            const float2 o = wm[wave][j];                       // same address on every lane: broadcast
            a = fmaf(o.x, fabsf(me.y - o.y), a);
        }
        const float dt = tr[i + 1] - tr[i];
        acc += 0.5f * me.x * a + (1.0f / 3.0f) * me.x * me.x * dt;
        if (grad_w) grad_w[(size_t)r * S + i] = a + (2.0f / 3.0f) * me.x * dt;
    }
    if (grad_w && lane == 0) grad_w[(size_t)r * S + n] = 0.0f;  // the last weight does not enter the loss
    acc = wave_sum(acc);
    if (lane == 0) loss_ray[r] = acc;
}

struct Proj {                        // per-ray projection chain and its Jacobian pieces
    float u, v;
    float dpx[2], dpy[2], dpz[2];    // d(u, v) / d(p.x, p.y, p.z)
};

// p: expected NDC point.  M: w2c rows (R | t).  Reference: NDC2Euclidean (utils.py:507-514),
// se3_transform_points (:516-518), perspective_projection (:521-525).
__device__ __forceinline__ Proj project_chain(float px, float py, float pz, const float *__restrict__ M,
                                              float H, float W, float f, bool want_jac) {
    const bool clamped = pz < -1.0f || pz > 0.99f;
    const float zc = fminf(fmaxf(pz, -1.0f), 0.99f);
    const float ze = 2.0f / (zc - 1.0f);
    const float kx = W / (2.0f * f), ky = H / (2.0f * f);
    const float xe = -px * ze * kx, ye = -py * ze * ky;
    const float lx = M[0] * xe + M[1] * ye + M[2] * ze + M[3];
    const float ly = M[4] * xe + M[5] * ye + M[6] * ze + M[7];
    const float lz = M[8] * xe + M[9] * ye + M[10] * ze + M[11];
    Proj o;
    o.u = lx * f / -lz + W / 2.0f;
    o.v = -ly * f / -lz + H / 2.0f;
    if (want_jac) {
        // d(xe, ye, ze)/dp
        const float dze = clamped ? 0.0f : -2.0f / ((zc - 1.0f) * (zc - 1.0f));
        const float dxe_dpx = -ze * kx, dxe_dpz = -px * kx * dze;
        const float dye_dpy = -ze * ky, dye_dpz = -py * ky * dze;
        // d(u, v)/d(lx, ly, lz):  u = -f lx / lz + W/2,  v = f ly / lz + H/2
        const float du_dlx = -f / lz, du_dlz = f * lx / (lz * lz);
        const float dv_dly = f / lz, dv_dlz = -f * ly / (lz * lz);
        float dl[3][3];                                        // d(lx, ly, lz)/d(px, py, pz)
        for (int k = 0; k < 3; k++) {
            dl[k][0] = M[4 * k] * dxe_dpx;
            dl[k][1] = M[4 * k + 1] * dye_dpy;
            dl[k][2] = M[4 * k] * dxe_dpz + M[4 * k + 1] * dye_dpz + M[4 * k + 2] * dze;
        }
        o.dpx[0] = du_dlx * dl[0][0] + du_dlz * dl[2][0], o.dpx[1] = dv_dly * dl[1][0] + dv_dlz * dl[2][0];
        o.dpy[0] = du_dlx * dl[0][1] + du_dlz * dl[2][1], o.dpy[1] = dv_dly * dl[1][1] + dv_dlz * dl[2][1];
        o.dpz[0] = du_dlx * dl[0][2] + du_dlz * dl[2][2], o.dpz[1] = dv_dly * dl[1][2] + dv_dlz * dl[2][2];
    }
    return o;
}

__device__ __forceinline__ void expected_point(const float *__restrict__ wr, const float *__restrict__ pr,
                                               int S, int lane, float &px, float &py, float &pz) {
    px = py = pz = 0.0f;
    for (int s = lane; s < S; s += 64) {
        const float ws = wr[s];
        px = fmaf(ws, pr[3 * s], px), py = fmaf(ws, pr[3 * s + 1], py), pz = fmaf(ws, pr[3 * s + 2], pz);
    }
    px = wave_sum(px), py = wave_sum(py), pz = wave_sum(pz);
}

__global__ __launch_bounds__(kWaves * 64) void project_rays_kernel(
    const float *__restrict__ w, const float *__restrict__ pts, const float *__restrict__ w2c, float H,
    float W, float f, int R, int S, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (r >= R) return;
    float px, py, pz;
    expected_point(w + (size_t)r * S, pts + (size_t)r * S * 3, S, lane, px, py, pz);
    const Proj o = project_chain(px, py, pz, w2c, H, W, f, false);
    if (lane == 0) out[2 * r] = o.u, out[2 * r + 1] = o.v;
}

__global__ __launch_bounds__(kWaves * 64) void project_rays_bwd_kernel(
    const float *__restrict__ w, const float *__restrict__ pts, const float *__restrict__ w2c, float H,
    float W, float f, const float *__restrict__ g, int R, int S, float *__restrict__ dw,
    float *__restrict__ dpts) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (r >= R) return;
    const float *wr = w + (size_t)r * S, *pr = pts + (size_t)r * S * 3;
    float px, py, pz;
    expected_point(wr, pr, S, lane, px, py, pz);
    const Proj o = project_chain(px, py, pz, w2c, H, W, f, true);
    const float gu = g[2 * r], gv = g[2 * r + 1];
    const float dx = gu * o.dpx[0] + gv * o.dpx[1], dy = gu * o.dpy[0] + gv * o.dpy[1],
                dz = gu * o.dpz[0] + gv * o.dpz[1];            // dL/dp
    for (int s = lane; s < S; s += 64) {
        const float ws = wr[s];
        if (dw) dw[(size_t)r * S + s] = dx * pr[3 * s] + dy * pr[3 * s + 1] + dz * pr[3 * s + 2];
        if (dpts) {
            float *d = dpts + ((size_t)r * S + s) * 3;
            d[0] = ws * dx, d[1] = ws * dy, d[2] = ws * dz;
        }
    }
}

}  // namespace

extern "C" int zest_distortion_fwd(const float *weights, const float *t_vals, int t_rows, int R, int S,
                                   float *loss_ray, float *grad_w, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(weights && t_vals && loss_ray, "zest_distortion_fwd: null pointer");
    ZEST_CHECK_ARG(R >= 0 && S >= 2 && S <= kMaxS + 1, "zest_distortion_fwd: bad shape R=%d S=%d (S <= %d)", R, S, kMaxS + 1);
    ZEST_CHECK_ARG(t_rows == 1 || t_rows == R, "zest_distortion_fwd: t_vals has %d rows, expected 1 or %d", t_rows, R);
    if (R == 0) return 0;
    hipLaunchKernelGGL(distortion_kernel, dim3(zest_div_up(R, kWaves)), dim3(kWaves * 64), 0,
                       (hipStream_t)stream, weights, t_vals, t_rows, R, S, loss_ray, grad_w);
    ZEST_RETURN_LAUNCH("zest_distortion_fwd");
}

extern "C" int zest_project_rays_fwd(const float *weights, const float *pts, const float *w2c, int H, int W,
                                     float focal, int R, int S, float *out, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(weights && pts && w2c && out, "zest_project_rays_fwd: null pointer");
    ZEST_CHECK_ARG(R >= 0 && S >= 1, "zest_project_rays_fwd: bad shape R=%d S=%d", R, S);
    if (R == 0) return 0;
    hipLaunchKernelGGL(project_rays_kernel, dim3(zest_div_up(R, kWaves)), dim3(kWaves * 64), 0,
                       (hipStream_t)stream, weights, pts, w2c, (float)H, (float)W, focal, R, S, out);
    ZEST_RETURN_LAUNCH("zest_project_rays_fwd");
}

extern "C" int zest_project_rays_bwd(const float *weights, const float *pts, const float *w2c, int H, int W,
                                     float focal, const float *grad_out, int R, int S, float *d_weights,
                                     float *d_pts, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(weights && pts && w2c && grad_out && (d_weights || d_pts), "zest_project_rays_bwd: null pointer");
    ZEST_CHECK_ARG(R >= 0 && S >= 1, "zest_project_rays_bwd: bad shape R=%d S=%d", R, S);
    if (R == 0) return 0;
    hipLaunchKernelGGL(project_rays_bwd_kernel, dim3(zest_div_up(R, kWaves)), dim3(kWaves * 64), 0,
                       (hipStream_t)stream, weights, pts, w2c, (float)H, (float)W, focal, grad_out, R, S,
                       d_weights, d_pts);
    ZEST_RETURN_LAUNCH("zest_project_rays_bwd");
}
