// Training path of the NeRF MLP, any depth / width / skips the desc allows: fp32 forward that keeps the
// activations backward needs, and the backward itself.  The GEMMs here are plain library
// shapes ([M x K] . [K x W] with M = rays x samples), so they go to rocBLAS sgemm (exact
// fp32 products on gfx950: there is no xf32 path); everything between them - bias,
// multiplicative / additive modulation by pts_bias(feats), ReLU and its mask, the skip
// concatenation, head activations - is fused into three small HIP kernels.
//
// Layout of one sample's saved activations (saved_per_sample floats), row-major per buffer:
//   pre[l] = Linear_l(h) + b_l  for the D trunk layers, m = pts_bias(feats), the
//   feature_linear output, and the pre-activation of the view layer (W/2 wide).
// Replaces the autograd of Renderer.forward / Renderer_linear.forward
// (reference networks.py:150-221, 283-319).
#include <rocblas/rocblas.h>
#include <map>
#include <mutex>
#include "mlp_plan.h"
#include "zest_common.cuh"

namespace {

inline int saved_per_sample(int D, int W) { return D * W + W + W + W / 2; }      // 2688 at the default shape
inline int work_per_sample(int W) {
    return 2 * W /*act ping-pong*/ + 4 * W /*d_a, d_b, act_re, dm*/ + W /*d_feat*/ + W / 2 /*d_hv*/ +
           16 /*head pre / d*/ + 1 /*ones*/;
}

constexpr int kSplitRows = 2048;                       // rows of M per weight-gradient partial product
constexpr size_t kPartialFloats = (size_t)64 * 256 * 320;   // split-K partials: up to 64 x [256 x 319]

struct Shape {
    int P, F, V, C_in, C_out, head, v2, mod, n_extra;
    int D, W, HW, skip_mask;
    bool skip_in(int l) const { return l > 0 && (skip_mask >> (l - 1) & 1); }      // layer l reads [pts | h]
};

bool shape_of(const zest_mlp_desc &d, Shape *out) {
    Shape &s = *out;
    zest::MlpShape ms;
    const char *err = nullptr;
    if (!zest::mlp_shape(d, &ms, &err)) {
        zest_set_error("zest_mlp_train: %s", err);
        return false;
    }
    s.D = ms.D, s.W = ms.W, s.HW = ms.W / 2, s.skip_mask = ms.skip_mask;
    s.P = d.in_ch_pts, s.mod = d.use_feat ? 1 : 0, s.F = s.mod ? d.in_ch_feat : 0, s.V = d.in_ch_views;
    s.C_in = s.P + s.F + s.V, s.head = d.head, s.v2 = d.net_type == 2;
    s.n_extra = d.head == ZEST_HEAD_BLEND ? 1 : (d.head == ZEST_HEAD_DYNAMIC ? 8 : 0);
    s.C_out = 4 + s.n_extra;
    return true;
}

std::map<int, rocblas_handle> g_handles;        // one per device: a handle belongs to the device it was created on
std::mutex g_mu;

rocblas_handle handle_for(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(g_mu);
    rocblas_handle &h = g_handles[dev];
    if (!h && rocblas_create_handle(&h) != rocblas_status_success) {
        h = nullptr;
        return nullptr;
    }
    rocblas_set_stream(h, st);
    rocblas_set_pointer_mode(h, rocblas_pointer_mode_host);
    return h;
}

// Row-major helpers on top of column-major sgemm.
// Y[M,N] (ldy) = X[M,K] (ldx) . Wt[N,K]^T (ldw)  (+ beta Y)
bool gemm_xwT(rocblas_handle h, int M, int N, int K, const float *X, int ldx, const float *Wt, int ldw,
              float *Y, int ldy, float beta) {
    const float one = 1.0f;
    return rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, N, M, K, &one, Wt, ldw, X, ldx,
                         &beta, Y, ldy) == rocblas_status_success;
}
// dX[M,K] (ldx) = dY[M,N] (ldy) . Wt[N,K] (ldw)  (+ beta dX)
bool gemm_dx(rocblas_handle h, int M, int N, int K, const float *dY, int ldy, const float *Wt, int ldw,
             float *dX, int ldx, float beta) {
    const float one = 1.0f;
    return rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, K, M, N, &one, Wt, ldw, dY, ldy,
                         &beta, dX, ldx) == rocblas_status_success;
}
// out[n] = sum_m A[m, n]  for n < N (N <= 256), A row-major with leading dimension lda.
// One thread per column, 256 rows per block, one float atomic per column and block.
__global__ void colsum_kernel(const float *__restrict__ A, int M, int N, int lda, float *__restrict__ out) {
    const int n = threadIdx.x;
    if (n >= N) return;
    const int m0 = blockIdx.x * 256, m1 = min(m0 + 256, M);
    float acc = 0.f;
    for (int m = m0; m < m1; m++) acc += A[(size_t)m * lda + n];
    atomicAdd(out + n, acc);
}
__global__ void zero_kernel(float *__restrict__ p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0.f;
}
// dW[n, k] (ldw) = sum_b partial[b][n][k]
__global__ void reduce_partials_kernel(const float *__restrict__ part, int nb, int N, int K, float *__restrict__ dW,
                                       int ldw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * K) return;
    float acc = 0.f;
    for (int b = 0; b < nb; b++) acc += part[(size_t)b * N * K + i];
    dW[(size_t)(i / K) * ldw + (i % K)] = acc;
}

struct Ctx {
    rocblas_handle h;
    hipStream_t st;
    float *partials;        // kPartialFloats of scratch for the split weight-gradient products
};

// dWt[N,K] (ldw) = dY[M,N]^T (ldy) . X[M,K] (ldx).  The reduction runs over M = rays x samples
// (131 072 at the headline shape) into a 256 x 256 output, so it is split into kSplitRows-row
// partial products (strided-batched sgemm: enough workgroups to fill the chip) and a reduce.
bool gemm_dw(const Ctx &c, int M, int N, int K, const float *dY, int ldy, const float *X, int ldx, float *dWt,
             int ldw) {
    const float one = 1.0f, zero = 0.0f;
    const int nb = M / kSplitRows, rem = M % kSplitRows;
    if (nb == 0 || (size_t)(nb + 1) * N * K > kPartialFloats)
        return rocblas_sgemm(c.h, rocblas_operation_none, rocblas_operation_transpose, K, N, M, &one, X, ldx, dY,
                             ldy, &zero, dWt, ldw) == rocblas_status_success;
    if (rocblas_sgemm_strided_batched(c.h, rocblas_operation_none, rocblas_operation_transpose, K, N, kSplitRows,
                                      &one, X, ldx, (rocblas_stride)kSplitRows * ldx, dY, ldy,
                                      (rocblas_stride)kSplitRows * ldy, &zero, c.partials, K,
                                      (rocblas_stride)N * K, nb) != rocblas_status_success)
        return false;
    int np = nb;
    if (rem) {
        if (rocblas_sgemm(c.h, rocblas_operation_none, rocblas_operation_transpose, K, N, rem, &one,
                          X + (size_t)nb * kSplitRows * ldx, ldx, dY + (size_t)nb * kSplitRows * ldy, ldy, &zero,
                          c.partials + (size_t)nb * N * K, K) != rocblas_status_success)
            return false;
        np++;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((N * K + 255) / 256), dim3(256), 0, c.st, c.partials, np, N, K,
                       dWt, ldw);
    return true;
}
// db[N] = column sums of dY[M,N] (ldy)
bool col_sums(const Ctx &c, int M, int N, const float *dY, int ldy, float *db) {
    hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, c.st, db, N);
    hipLaunchKernelGGL(colsum_kernel, dim3((M + 255) / 256), dim3(256), 0, c.st, dY, M, N, ldy, db);
    return true;
}

// pre = y + b (in place);  act = relu(mod(pre, m))          [n columns per row]
__global__ void bias_act_kernel(float *__restrict__ pre, const float *__restrict__ b,
                                const float *__restrict__ m, float *__restrict__ act, long long n_elem,
                                int n, int v2, int relu) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_elem) return;
    const float p = pre[i] + b[i % n];
    pre[i] = p;
    if (act) {
        float v = m ? (v2 ? p + m[i] : p * m[i]) : p;
        act[i] = relu ? fmaxf(v, 0.f) : v;
    }
}
// act = relu(mod(pre, m)) from saved pre (backward recompute)
__global__ void react_kernel(const float *__restrict__ pre, const float *__restrict__ m,
                             float *__restrict__ act, long long n_elem, int v2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_elem) return;
    const float p = pre[i];
    act[i] = fmaxf(m ? (v2 ? p + m[i] : p * m[i]) : p, 0.f);
}
// d (in: dL/dact, out: dL/dpre) and dm += dL/dm, through relu(mod(pre, m)), with the bias gradient
// folded in: db[n] += sum over the block's rows of dL/dpre.
// One thread per column (N = 128 or 256 columns = blockDim.x), kBwdRows rows per block: the
// separate column-sum pass over dL/dpre (11 % of a training step) disappears.
constexpr int kBwdRows = 64;
__global__ void act_bwd_colsum_kernel(float *__restrict__ d, const float *__restrict__ pre,
                                      const float *__restrict__ m, float *__restrict__ dm, int M, int N, int v2,
                                      float *__restrict__ db) {
    const int n = threadIdx.x;
    const int m0 = blockIdx.x * kBwdRows, m1 = min(m0 + kBwdRows, M);
    float acc = 0.f;
    for (int r = m0; r < m1; r++) {
        const size_t i = (size_t)r * N + n;
        const float p = pre[i], g = d[i];
        float out;
        if (!m) {
            out = p > 0.f ? g : 0.f;
        } else {
            const float mv = m[i];
            const bool on = (v2 ? p + mv : p * mv) > 0.f;
            const float go = on ? g : 0.f;
            out = v2 ? go : go * mv;
            dm[i] += v2 ? go : go * p;
        }
        d[i] = out;
        acc += out;
    }
    atomicAdd(db + n, acc);
}
__global__ void fill_kernel(float *__restrict__ p, long long n, float v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// head pre-activations hp[M,16] (0-2 rgb, 3 alpha, 4.. extras) -> out[M,C_out] with the
// network's output activations (v0: raw rgb/alpha; v2: sigmoid rgb, relu alpha;
// sigmoid blend / tanh scene flow / sigmoid prob)
__global__ void heads_fwd_kernel(const float *__restrict__ hp, int M, int C_out, int head, int v2,
                                 float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)M * C_out) return;
    const int c = (int)(i % C_out);
    const float v = hp[(i / C_out) * 16 + c];
    float o = v;
    if (c < 3) o = v2 ? zest_sigmoid(v) : v;
    else if (c == 3) o = v2 ? fmaxf(v, 0.f) : v;
    else if (head == ZEST_HEAD_DYNAMIC && c <= 9) o = tanhf(v);
    else o = zest_sigmoid(v);
    out[i] = o;
}
// dL/dout -> dL/d(head pre-activation) using the saved outputs
__global__ void heads_bwd_kernel(const float *__restrict__ g_out, const float *__restrict__ out, int M,
                                 int C_out, int head, int v2, float *__restrict__ dhp) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)M * 16) return;
    const int c = (int)(i % 16);
    const long long m = i / 16;
    float d = 0.f;
    if (c < C_out) {
        const float g = g_out[m * C_out + c], o = out[m * C_out + c];
        if (c < 3) d = v2 ? g * o * (1.f - o) : g;
        else if (c == 3) d = v2 ? (o > 0.f ? g : 0.f) : g;
        else if (head == ZEST_HEAD_DYNAMIC && c <= 9) d = g * (1.f - o * o);
        else d = g * o * (1.f - o);
    }
    dhp[i] = d;
}

inline dim3 grid1(long long n) { return dim3((unsigned)((n + 255) / 256)); }

struct Params {                 // weight / bias pointers by ZEST_P_* slot
    const float *w[ZEST_P_COUNT], *b[ZEST_P_COUNT];
};
struct GradParams {
    float *w[ZEST_P_COUNT], *b[ZEST_P_COUNT];
};

}  // namespace

extern "C" size_t zest_mlp_train_saved_floats(const zest_mlp_desc *desc, int M) {
    Shape s;
    return desc && M > 0 && shape_of(*desc, &s) ? (size_t)M * saved_per_sample(s.D, s.W) : 0;
}
extern "C" size_t zest_mlp_train_workspace_floats(const zest_mlp_desc *desc, int M) {
    Shape s;
    return desc && M > 0 && shape_of(*desc, &s) ? (size_t)M * work_per_sample(s.W) + kPartialFloats : 0;
}

#define RB(x)                                                       \
    do {                                                            \
        if (!(x)) {                                                 \
            zest_set_error("zest_mlp_train: rocBLAS call failed");  \
            return (int)hipErrorUnknown;                            \
        }                                                           \
    } while (0)
#define EW(kernel, n, ...) hipLaunchKernelGGL(kernel, grid1(n), dim3(256), 0, st, __VA_ARGS__)

extern "C" int zest_mlp_train_fwd(const zest_mlp_desc *desc, const float *const *params, const float *x,
                                  int M, float *saved, float *workspace, float *out, void *stream) {
    if (M == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(desc && params && x && saved && workspace && out && M > 0, "zest_mlp_train_fwd: bad argument");
    ZEST_CHECK_ARG(desc->net_type == 0 || desc->net_type == 2, "zest_mlp_train_fwd: net_type must be 0 or 2");
    Shape s;
    if (!shape_of(*desc, &s)) return (int)hipErrorInvalidValue;
    const int D = s.D, W = s.W, HW = s.HW;
    hipStream_t st = (hipStream_t)stream;
    rocblas_handle h = handle_for(st);
    ZEST_CHECK_ARG(h, "zest_mlp_train_fwd: cannot create a rocBLAS handle");
    Params p;
    for (int i = 0; i < ZEST_P_COUNT; i++) p.w[i] = params[2 * i], p.b[i] = params[2 * i + 1];
    const long long MW = (long long)M * W;
    float *pre[8];
    for (int l = 0; l < D; l++) pre[l] = saved + (size_t)l * MW;
    float *mbuf = saved + D * MW, *featl = saved + (D + 1) * MW, *hvpre = saved + (D + 2) * MW;
    float *actA = workspace, *actB = workspace + MW, *hp = workspace + 7 * MW + (long long)M * HW;
    const float *xp = x, *xf = x + s.P, *xv = x + s.P + s.F;
    const float *m = nullptr;
    if (s.mod) {
        RB(gemm_xwT(h, M, W, s.F, xf, s.C_in, p.w[ZEST_P_PTS_BIAS], s.F, mbuf, W, 0.f));
        EW(bias_act_kernel, MW, mbuf, p.b[ZEST_P_PTS_BIAS], (const float *)nullptr, (float *)nullptr, MW, W, 0, 0);
        m = mbuf;
    }
    float *cur = actA, *nxt = actB;
    for (int l = 0; l < D; l++) {
        if (l == 0) {
            RB(gemm_xwT(h, M, W, s.P, xp, s.C_in, p.w[0], s.P, pre[0], W, 0.f));
        } else if (s.skip_in(l)) {      // input = [pts | h]
            RB(gemm_xwT(h, M, W, s.P, xp, s.C_in, p.w[l], W + s.P, pre[l], W, 0.f));
            RB(gemm_xwT(h, M, W, W, cur, W, p.w[l] + s.P, W + s.P, pre[l], W, 1.f));
        } else {
            RB(gemm_xwT(h, M, W, W, cur, W, p.w[l], W, pre[l], W, 0.f));
        }
        EW(bias_act_kernel, MW, pre[l], p.b[l], m, nxt, MW, W, s.v2, 1);
        float *t = cur;
        cur = nxt, nxt = t;
    }
    // heads on the trunk output `cur`: alpha (col 3), extras (cols 4..)
    EW(fill_kernel, (long long)M * 16, hp, (long long)M * 16, 0.f);
    RB(gemm_xwT(h, M, 1, W, cur, W, p.w[ZEST_P_ALPHA], W, hp + 3, 16, 0.f));
    float hb[16] = {0};
    // biases of the heads are added below through a tiny bias vector on the device: reuse nxt[0..15]
    if (s.head == ZEST_HEAD_BLEND) RB(gemm_xwT(h, M, 1, W, cur, W, p.w[ZEST_P_HEAD0], W, hp + 4, 16, 0.f));
    if (s.head == ZEST_HEAD_DYNAMIC) {
        RB(gemm_xwT(h, M, 6, W, cur, W, p.w[ZEST_P_HEAD0], W, hp + 4, 16, 0.f));
        RB(gemm_xwT(h, M, 2, W, cur, W, p.w[ZEST_P_HEAD1], W, hp + 10, 16, 0.f));
    }
    (void)hb;
    // feature -> views -> rgb
    RB(gemm_xwT(h, M, W, W, cur, W, p.w[ZEST_P_FEATURE], W, featl, W, 0.f));
    EW(bias_act_kernel, MW, featl, p.b[ZEST_P_FEATURE], (const float *)nullptr, (float *)nullptr, MW, W, 0, 0);
    RB(gemm_xwT(h, M, HW, W, featl, W, p.w[ZEST_P_VIEWS], W + s.V, hvpre, HW, 0.f));
    RB(gemm_xwT(h, M, HW, s.V, xv, s.C_in, p.w[ZEST_P_VIEWS] + W, W + s.V, hvpre, HW, 1.f));
    EW(bias_act_kernel, (long long)M * HW, hvpre, p.b[ZEST_P_VIEWS], (const float *)nullptr, nxt, (long long)M * HW,
       HW, 0, 1);
    RB(gemm_xwT(h, M, 3, HW, nxt, HW, p.w[ZEST_P_RGB], HW, hp, 16, 0.f));
    // head biases: assemble the 16-entry bias row on the device from the parameter tensors
    {
        float *brow = nxt + (long long)M * HW;      // 16 floats of scratch behind the view activations
        EW(fill_kernel, 16, brow, 16, 0.f);
        (void)hipMemcpyAsync(brow, p.b[ZEST_P_RGB], 3 * sizeof(float), hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(brow + 3, p.b[ZEST_P_ALPHA], sizeof(float), hipMemcpyDeviceToDevice, st);
        if (s.head == ZEST_HEAD_BLEND)
            (void)hipMemcpyAsync(brow + 4, p.b[ZEST_P_HEAD0], sizeof(float), hipMemcpyDeviceToDevice, st);
        if (s.head == ZEST_HEAD_DYNAMIC) {
            (void)hipMemcpyAsync(brow + 4, p.b[ZEST_P_HEAD0], 6 * sizeof(float), hipMemcpyDeviceToDevice, st);
            (void)hipMemcpyAsync(brow + 10, p.b[ZEST_P_HEAD1], 2 * sizeof(float), hipMemcpyDeviceToDevice, st);
        }
        EW(bias_act_kernel, (long long)M * 16, hp, brow, (const float *)nullptr, (float *)nullptr, (long long)M * 16,
           16, 0, 0);
    }
    EW(heads_fwd_kernel, (long long)M * s.C_out, hp, M, s.C_out, s.head, s.v2, out);
    ZEST_RETURN_LAUNCH("zest_mlp_train_fwd");
}

extern "C" int zest_mlp_train_bwd(const zest_mlp_desc *desc, const float *const *params, const float *x,
                                  int M, const float *saved, const float *out, const float *g_out,
                                  float *workspace, float *g_x, float *const *g_params, void *stream) {
    if (M == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(desc && params && x && saved && out && g_out && workspace && g_params && M > 0,
                   "zest_mlp_train_bwd: bad argument");
    Shape s;
    if (!shape_of(*desc, &s)) return (int)hipErrorInvalidValue;
    const int D = s.D, W = s.W, HW = s.HW;
    hipStream_t st = (hipStream_t)stream;
    rocblas_handle h = handle_for(st);
    ZEST_CHECK_ARG(h, "zest_mlp_train_bwd: cannot create a rocBLAS handle");
    Params p;
    GradParams g;
    for (int i = 0; i < ZEST_P_COUNT; i++) {
        p.w[i] = params[2 * i], p.b[i] = params[2 * i + 1];
        g.w[i] = g_params[2 * i], g.b[i] = g_params[2 * i + 1];
    }
    const long long MW = (long long)M * W, MH = (long long)M * HW;
    const float *pre[8];
    for (int l = 0; l < D; l++) pre[l] = saved + (size_t)l * MW;
    const float *mbuf = s.mod ? saved + D * MW : nullptr, *featl = saved + (D + 1) * MW, *hvpre = saved + (D + 2) * MW;
    float *da = workspace + 2 * MW, *db_ = workspace + 3 * MW, *act = workspace + 4 * MW, *dm = workspace + 5 * MW;
    float *dfeat = workspace + 6 * MW, *dhv = workspace + 7 * MW, *dhp = dhv + MH;
    const Ctx cx{h, st, workspace + (size_t)M * work_per_sample(W)};
    const float *xp = x, *xf = x + s.P, *xv = x + s.P + s.F;
    float *gxp = g_x, *gxf = g_x ? g_x + s.P : nullptr;
    if (s.mod) EW(fill_kernel, MW, dm, MW, 0.f);
    if (g_x) EW(fill_kernel, (long long)M * s.C_in, g_x, (long long)M * s.C_in, 0.f);

    // heads: dL/d(pre-activations)
    EW(heads_bwd_kernel, (long long)M * 16, g_out, out, M, s.C_out, s.head, s.v2, dhp);
    // rgb_linear on hv = relu(hvpre)
    EW(react_kernel, MH, hvpre, (const float *)nullptr, act, MH, 0);
    RB(gemm_dw(cx, M, 3, HW, dhp, 16, act, HW, g.w[ZEST_P_RGB], HW));
    RB(col_sums(cx, M, 3, dhp, 16, g.b[ZEST_P_RGB]));
    RB(gemm_dx(h, M, 3, HW, dhp, 16, p.w[ZEST_P_RGB], HW, dhv, HW, 0.f));
    hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, st, g.b[ZEST_P_VIEWS], HW);
    hipLaunchKernelGGL(act_bwd_colsum_kernel, dim3((M + kBwdRows - 1) / kBwdRows), dim3(HW), 0, st, dhv, hvpre,
                       (const float *)nullptr, (float *)nullptr, M, HW, 0, g.b[ZEST_P_VIEWS]);
    // views_linears.0 on [feature | views]
    RB(gemm_dw(cx, M, HW, W, dhv, HW, featl, W, g.w[ZEST_P_VIEWS], W + s.V));
    RB(gemm_dw(cx, M, HW, s.V, dhv, HW, xv, s.C_in, g.w[ZEST_P_VIEWS] + W, W + s.V));
    RB(gemm_dx(h, M, HW, W, dhv, HW, p.w[ZEST_P_VIEWS], W + s.V, dfeat, W, 0.f));
    // trunk output h(D-1) = relu(mod(pre(D-1), m)): feature_linear, alpha and the extra heads read it
    EW(react_kernel, MW, pre[D - 1], mbuf, act, MW, s.v2);
    RB(gemm_dw(cx, M, W, W, dfeat, W, act, W, g.w[ZEST_P_FEATURE], W));
    RB(col_sums(cx, M, W, dfeat, W, g.b[ZEST_P_FEATURE]));
    RB(gemm_dx(h, M, W, W, dfeat, W, p.w[ZEST_P_FEATURE], W, da, W, 0.f));
    RB(gemm_dw(cx, M, 1, W, dhp + 3, 16, act, W, g.w[ZEST_P_ALPHA], W));
    RB(col_sums(cx, M, 1, dhp + 3, 16, g.b[ZEST_P_ALPHA]));
    RB(gemm_dx(h, M, 1, W, dhp + 3, 16, p.w[ZEST_P_ALPHA], W, da, W, 1.f));
    if (s.head != ZEST_HEAD_NONE) {
        const int n0 = s.head == ZEST_HEAD_BLEND ? 1 : 6;
        RB(gemm_dw(cx, M, n0, W, dhp + 4, 16, act, W, g.w[ZEST_P_HEAD0], W));
        RB(col_sums(cx, M, n0, dhp + 4, 16, g.b[ZEST_P_HEAD0]));
        RB(gemm_dx(h, M, n0, W, dhp + 4, 16, p.w[ZEST_P_HEAD0], W, da, W, 1.f));
        if (s.head == ZEST_HEAD_DYNAMIC) {
            RB(gemm_dw(cx, M, 2, W, dhp + 10, 16, act, W, g.w[ZEST_P_HEAD1], W));
            RB(col_sums(cx, M, 2, dhp + 10, 16, g.b[ZEST_P_HEAD1]));
            RB(gemm_dx(h, M, 2, W, dhp + 10, 16, p.w[ZEST_P_HEAD1], W, da, W, 1.f));
        }
    }
    // trunk, last layer first.  da = dL/d(act_l) on entry to layer l.
    float *dcur = da, *dnxt = db_;
    for (int l = D - 1; l >= 0; l--) {
        // dcur = dL/dpre_l, and its column sums = the bias gradient
        hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, st, g.b[l], W);
        hipLaunchKernelGGL(act_bwd_colsum_kernel, dim3((M + kBwdRows - 1) / kBwdRows), dim3(W), 0, st, dcur, pre[l],
                           mbuf, dm, M, W, s.v2, g.b[l]);
        if (l == 0) {
            RB(gemm_dw(cx, M, W, s.P, dcur, W, xp, s.C_in, g.w[0], s.P));
            if (g_x) RB(gemm_dx(h, M, W, s.P, dcur, W, p.w[0], s.P, gxp, s.C_in, 1.f));
            break;
        }
        EW(react_kernel, MW, pre[l - 1], mbuf, act, MW, s.v2);              // input of layer l
        if (s.skip_in(l)) {
            RB(gemm_dw(cx, M, W, s.P, dcur, W, xp, s.C_in, g.w[l], W + s.P));
            RB(gemm_dw(cx, M, W, W, dcur, W, act, W, g.w[l] + s.P, W + s.P));
            if (g_x) RB(gemm_dx(h, M, W, s.P, dcur, W, p.w[l], W + s.P, gxp, s.C_in, 1.f));
            RB(gemm_dx(h, M, W, W, dcur, W, p.w[l] + s.P, W + s.P, dnxt, W, 0.f));
        } else {
            RB(gemm_dw(cx, M, W, W, dcur, W, act, W, g.w[l], W));
            RB(gemm_dx(h, M, W, W, dcur, W, p.w[l], W, dnxt, W, 0.f));
        }
        float *t = dcur;
        dcur = dnxt, dnxt = t;
    }
    if (s.mod) {        // m = pts_bias(feats)
        RB(gemm_dw(cx, M, W, s.F, dm, W, xf, s.C_in, g.w[ZEST_P_PTS_BIAS], s.F));
        RB(col_sums(cx, M, W, dm, W, g.b[ZEST_P_PTS_BIAS]));
        if (g_x) RB(gemm_dx(h, M, W, s.F, dm, W, p.w[ZEST_P_PTS_BIAS], s.F, gxf, s.C_in, 1.f));
    }
    ZEST_RETURN_LAUNCH("zest_mlp_train_bwd");
}
