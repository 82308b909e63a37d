// Standalone per-sample operators: positional encoding, encoding-volume lookup, colour
// gather, and the assembled MLP input.  These back the module-level API of the reference
// (Embedding.forward, utils.index_point_feature, utils.build_color_volume, prepare_pts) and
// the training path, where per-sample tensors must exist in HBM.  The fused inference kernel
// (fused.hip) uses the same device functions and never materialises these tensors.
// All four are HBM/L2-bound gathers: one thread per sample, 16/32-byte corner reads from the
// channels-last copies of the volume ([H,W,D,8], depth innermost) and the images.
#include "zest_sample_ops.cuh"

namespace {

constexpr int kThreads = 256;

__global__ void embed_kernel(const float *__restrict__ x, int M, int C, int L,
                             float *__restrict__ y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)M * C) return;
    const int m = (int)(i / C), c = (int)(i % C);
    const float v = x[i];
    float *row = y + (size_t)m * C * (2 * L + 1);
    row[c] = v;
    float f = 1.0f;
    for (int k = 0; k < L; k++) {
        float s, co;
        zest_sincos(v * f, &s, &co);
        row[C * (1 + 2 * k) + c] = s;
        row[C * (2 + 2 * k) + c] = co;
        f *= 2.0f;
    }
}

// [8,D,H,W] -> [H,W,D,8] (zest_vox): a workgroup transposes the tile (one y, 32 x, 16 z) through LDS - 128-byte
// runs along x in, one 512-byte run of 16 voxels per x out.
constexpr int kTx = 32, kTz = 16, kTk = kTz * (kTx + 1) + 4;         // per-channel stride of the LDS tile

__global__ __launch_bounds__(kThreads) void volume_to_cl_kernel(const float *__restrict__ vol, int D, int H, int W,
                                                                float4 *__restrict__ out) {
    __shared__ float tile[8 * kTk];
    const int x0 = blockIdx.x * kTx, y = blockIdx.y, z0 = blockIdx.z * kTz;
    const size_t plane = (size_t)H * W, nvox = plane * D;
    const int tx = threadIdx.x & (kTx - 1), tr = threadIdx.x / kTx;
    if (x0 + tx < W) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            for (int zz = tr; zz < kTz && z0 + zz < D; zz += kThreads / kTx)
                tile[k * kTk + zz * (kTx + 1) + tx] = vol[k * nvox + (size_t)(z0 + zz) * plane + (size_t)y * W + x0 + tx];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kTx * kTz * 2; i += kThreads) {
        const int xx = i / (2 * kTz), zz = (i % (2 * kTz)) >> 1, half = i & 1;
        if (x0 + xx < W && z0 + zz < D) {
            const float *t = tile + 4 * half * kTk + zz * (kTx + 1) + xx;
            out[2 * zest_vox(z0 + zz, y, x0 + xx, D, W) + half] = make_float4(t[0], t[kTk], t[2 * kTk], t[3 * kTk]);
        }
    }
}

__global__ void images_to_cl_kernel(const float *__restrict__ imgs, int V, long long npix,
                                    float4 *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)V * npix) return;
    const long long v = i / npix, p = i % npix;
    const float *b = imgs + (size_t)v * 3 * npix + p;
    out[i] = make_float4(b[0], b[npix], b[2 * npix], 0.0f);
}

__global__ void volume_lookup_kernel(const float4 *__restrict__ vol, int D, int H, int W,
                                     const float *__restrict__ ndc, int M,
                                     float4 *__restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float f[8];
    zest_volume_trilerp<2>(vol, D, H, W, ndc[3 * m], ndc[3 * m + 1], ndc[3 * m + 2], f);
    out[2 * m] = make_float4(f[0], f[1], f[2], f[3]);
    out[2 * m + 1] = make_float4(f[4], f[5], f[6], f[7]);
}

__global__ void color_lookup_kernel(const float4 *__restrict__ imgs, int V, int H, int W,
                                    const float *__restrict__ w2cs,
                                    const float *__restrict__ intr,
                                    const float *__restrict__ pts, int M,
                                    float4 *__restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float px = pts[3 * m], py = pts[3 * m + 1], pz = pts[3 * m + 2];
    for (int v = 0; v < V; v++) {
        const ZestCam c = zest_load_cam(w2cs, intr, v);      // wave-uniform: scalar loads
        out[(size_t)m * V + v] = zest_color_tap(imgs + (size_t)v * H * W, H, W, c, px, py, pz);
    }
}

// x[m] = PE10(ndc[,t]) | vol(8) | colours(4V) | PE4(view dir)
// One wave per 64 consecutive samples, one lane per sample; the rows (90 .. 135 floats each) are assembled in an
// LDS tile (odd row stride: conflict-free column writes) and leave as ONE contiguous run of 64 rows written with
// consecutive lanes on consecutive floats.  (A lane writing its own 520-byte row straight to HBM - the first version
// - touched 64 different lines per store instruction: 98 us per 1024 x 128 batch, 22 % of the HBM rate.)
constexpr int kEncRows = 64, kEncMaxC = 4 * 21 + 8 + 4 * 16 + 27;     // widest row: xyzt + 16 views

__global__ __launch_bounds__(kEncRows) void encode_kernel(
    const float *__restrict__ ndc, const float *__restrict__ pts, const float *__restrict__ rays_dir, int R, int S,
    int has_time, float t, const float4 *__restrict__ vol, int D, int Hv, int Wv, const float4 *__restrict__ imgs, int V,
    int H, int W, const float *__restrict__ w2cs, const float *__restrict__ intr, float *__restrict__ x) {
    extern __shared__ float tile[];                          // [64][stride]
    const long long M = (long long)R * S;
    const long long m0 = (long long)blockIdx.x * kEncRows, m = m0 + threadIdx.x;
    const int C = 3 + has_time;
    const int P = C * 21, F = vol ? 8 + 4 * V : 0, C_in = P + F + 27, stride = C_in | 1;
    if (m < M) {
        const int r = (int)(m / S);
        float *row = tile + threadIdx.x * stride;
        const float p[4] = {ndc[3 * m], ndc[3 * m + 1], ndc[3 * m + 2], t};
        for (int c = 0; c < C; c++) {
            row[c] = p[c];
            float f = 1.0f;
            for (int k = 0; k < 10; k++) {
                float s, co;
                zest_sincos(p[c] * f, &s, &co);
                row[C * (1 + 2 * k) + c] = s;
                row[C * (2 + 2 * k) + c] = co;
                f *= 2.0f;
            }
        }
        if (vol) {
            float f8[8];
            zest_volume_trilerp<2>(vol, D, Hv, Wv, p[0], p[1], p[2], f8);
#pragma unroll
            for (int i = 0; i < 8; i++) row[P + i] = f8[i];
            const float px = pts[3 * m], py = pts[3 * m + 1], pz = pts[3 * m + 2];
            for (int v = 0; v < V; v++) {
                const ZestCam c = zest_load_cam(w2cs, intr, v);
                const float4 o = zest_color_tap(imgs + (size_t)v * H * W, H, W, c, px, py, pz);
                float *d = row + P + 8 + 4 * v;
                d[0] = o.x, d[1] = o.y, d[2] = o.z, d[3] = o.w;
            }
        }
        float dv[3];
        zest_view_dir(rays_dir + 3 * r, w2cs, dv);
        float *dr = row + P + F;
        for (int c = 0; c < 3; c++) {
            dr[c] = dv[c];
            float f = 1.0f;
            for (int k = 0; k < 4; k++) {
                float s, co;
                zest_sincos(dv[c] * f, &s, &co);
                dr[3 * (1 + 2 * k) + c] = s;
                dr[3 * (2 + 2 * k) + c] = co;
                f *= 2.0f;
            }
        }
    }
    __syncthreads();
    // the 64 rows are one contiguous run of x: lane l takes floats l, l + 64, ...
    const long long rows = min((long long)kEncRows, M - m0);
    const int n = (int)rows * C_in;
    float *dst = x + (size_t)m0 * C_in;
    int rr = 0, cc = (int)threadIdx.x;                       // row / column of element threadIdx.x
    while (cc >= C_in) cc -= C_in, rr++;
    const int step_r = kEncRows / C_in, step_c = kEncRows % C_in;
    for (int i = threadIdx.x; i < n; i += kEncRows) {
        dst[i] = tile[rr * stride + cc];
        rr += step_r, cc += step_c;
        if (cc >= C_in) cc -= C_in, rr++;
    }
}

// Backward of encode_kernel for the inputs that carry gradients in the reference's training
// graph: the volume coordinates (through the positional encoding and through the trilinear
// lookup's dependence on them - needed for the scene-flow displaced points, renderer.py:461,488)
// and the encoding volume itself (scatter-add; MVSNet trains through it).  World points, source
// images, cameras and ray directions are data and receive no gradient.
// Eight lanes per sample (lane & 7 = volume channel): a corner's scatter-add is then 8 adjacent floats = one
// 32-byte segment per sample and 8 segments per wave instruction, instead of 64 single floats in 64 different
// rows (the shape at which memory-side float atomics run at 1/17 of their rate: the first version of this
// kernel spent 0.45 ms per 1024 x 128 batch on them).  The octet's lanes 0-2 also take one coordinate each of
// the positional-encoding part; the per-corner dot products are reduced over the octet with shuffles.
__global__ void encode_bwd_kernel(const float *__restrict__ g_x, const float *__restrict__ ndc, int R, int S,
                                  int has_time, float t, const float4 *__restrict__ vol, int D, int Hv, int Wv,
                                  int V, float *__restrict__ g_ndc, float *__restrict__ g_vol) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long m_raw = tid >> 3;
    const int ch = (int)(tid & 7);
    const bool live = m_raw < (long long)R * S;
    const long long m = live ? m_raw : 0;                     // idle octets compute on sample 0 and store nothing
    const int C = 3 + has_time;
    const int P = C * 21, F = vol ? 8 + 4 * V : 0;
    const float *row = g_x + (size_t)m * (P + F + 27);
    const float p[3] = {ndc[3 * m], ndc[3 * m + 1], ndc[3 * m + 2]};
    float gpe = 0.0f;                                          // lanes 0-2: d / d coordinate ch through the encoding
    if (ch < 3) {
        float acc = row[ch], f = 1.0f;
        for (int k = 0; k < 10; k++) {
            float sn, cs;
            zest_sincos(p[ch] * f, &sn, &cs);
            acc += f * (cs * row[C * (1 + 2 * k) + ch] - sn * row[C * (2 + 2 * k) + ch]);
            f *= 2.0f;
        }
        gpe = acc;
    }
    float dxs = 0.f, dys = 0.f, dzs = 0.f;
    if (vol) {
        const float gf = row[P + ch];
        const float fx = zest_unnorm(p[0], Wv), fy = zest_unnorm(p[1], Hv), fz = zest_unnorm(p[2], D);
        const bool inside = fx > -2.0f && fx < (float)Wv + 1.0f && fy > -2.0f && fy < (float)Hv + 1.0f &&
                            fz > -2.0f && fz < (float)D + 1.0f;
        const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
        const float tx = fx - x0f, ty = fy - y0f, tz = fz - z0f;
        const int x0 = inside ? (int)x0f : -8, y0 = inside ? (int)y0f : -8, z0 = inside ? (int)z0f : -8;
        const float *volf = reinterpret_cast<const float *>(vol);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
            const int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
            const bool ok = (unsigned)xi < (unsigned)Wv && (unsigned)yi < (unsigned)Hv && (unsigned)zi < (unsigned)D;
            const size_t vox = ok ? zest_vox(zi, yi, xi, D, Wv) : 0;
            const float wx = dx ? tx : 1.0f - tx, wy = dy ? ty : 1.0f - ty, wz = dz ? tz : 1.0f - tz;
            float dot = ok ? volf[8 * vox + ch] * gf : 0.0f;                    // octet-uniform `ok`
            dot += __shfl_xor(dot, 1, 64), dot += __shfl_xor(dot, 2, 64), dot += __shfl_xor(dot, 4, 64);
            dxs += (dx ? 1.0f : -1.0f) * wy * wz * dot;
            dys += (dy ? 1.0f : -1.0f) * wx * wz * dot;
            dzs += (dz ? 1.0f : -1.0f) * wx * wy * dot;
            if (g_vol && ok && live) atomicAdd(g_vol + 8 * vox + ch, wx * wy * wz * gf);
        }
    }
    if (live && ch < 3) {
        const float scale = ch == 0 ? (float)(Wv - 1) : (ch == 1 ? (float)(Hv - 1) : (float)(D - 1));
        const float dv = ch == 0 ? dxs : (ch == 1 ? dys : dzs);
        g_ndc[3 * m + ch] = gpe + (vol ? dv * scale : 0.0f);
    }
}

// gradient volume [H,W,D,8] (zest_vox) -> the reference's layout [8,D,H,W]: the inverse tile transpose
__global__ __launch_bounds__(kThreads) void volume_from_cl_kernel(const float4 *__restrict__ cl, int D, int H, int W,
                                                                  float *__restrict__ out) {
    __shared__ float tile[8 * kTk];
    const int x0 = blockIdx.x * kTx, y = blockIdx.y, z0 = blockIdx.z * kTz;
    const size_t plane = (size_t)H * W, nvox = plane * D;
    for (int i = threadIdx.x; i < kTx * kTz * 2; i += kThreads) {
        const int xx = i / (2 * kTz), zz = (i % (2 * kTz)) >> 1, half = i & 1;
        if (x0 + xx < W && z0 + zz < D) {
            const float4 v = cl[2 * zest_vox(z0 + zz, y, x0 + xx, D, W) + half];
            float *t = tile + 4 * half * kTk + zz * (kTx + 1) + xx;
            t[0] = v.x, t[kTk] = v.y, t[2 * kTk] = v.z, t[3 * kTk] = v.w;
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & (kTx - 1), tr = threadIdx.x / kTx;
    if (x0 + tx < W) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            for (int zz = tr; zz < kTz && z0 + zz < D; zz += kThreads / kTx)
                out[k * nvox + (size_t)(z0 + zz) * plane + (size_t)y * W + x0 + tx] = tile[k * kTk + zz * (kTx + 1) + tx];
    }
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int zest_encode_bwd(const float *g_x, const float *ndc, int R, int S, int has_time, float t,
                               const float *vol_cl, int D, int Hv, int Wv, int V, float *g_ndc,
                               float *g_vol_cl, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(g_x && ndc && g_ndc, "zest_encode_bwd: g_x, ndc and g_ndc are required");
    ZEST_CHECK_ARG(R >= 0 && S >= 1, "zest_encode_bwd: bad shape");
    ZEST_CHECK_ARG(!vol_cl || (aligned16(vol_cl) && D >= 1 && Hv >= 1 && Wv >= 1 && V >= 1),
                   "zest_encode_bwd: bad volume");
    ZEST_CHECK_ARG(!g_vol_cl || vol_cl, "zest_encode_bwd: g_vol_cl needs vol_cl");
    if (R == 0) return 0;
    hipLaunchKernelGGL(encode_bwd_kernel, dim3(zest_div_up((long long)R * S * 8, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, g_x, ndc, R, S, has_time ? 1 : 0, t, (const float4 *)vol_cl, D, Hv,
                       Wv, V, g_ndc, g_vol_cl);
    ZEST_RETURN_LAUNCH("zest_encode_bwd");
}

extern "C" int zest_volume_from_cl(const float *vol_cl, int D, int H, int W, float *vol, void *stream) {
    ZEST_CHECK_ARG(vol_cl && vol && aligned16(vol_cl), "zest_volume_from_cl: bad pointer");
    ZEST_CHECK_ARG(D >= 1 && H >= 1 && W >= 1, "zest_volume_from_cl: bad shape");
    ZEST_CHECK_ARG(H <= 65535 && D <= 65535 * kTz, "zest_volume_from_cl: volume too large for the launch grid");
    hipLaunchKernelGGL(volume_from_cl_kernel, dim3(zest_div_up(W, kTx), H, zest_div_up(D, kTz)), dim3(kThreads), 0,
                       (hipStream_t)stream, (const float4 *)vol_cl, D, H, W, vol);
    ZEST_RETURN_LAUNCH("zest_volume_from_cl");
}

extern "C" int zest_embed_fwd(const float *x, int M, int C, int L, float *y, void *stream) {
    if (M == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(x && y, "zest_embed_fwd: null pointer");
    ZEST_CHECK_ARG(M >= 0 && C >= 1 && L >= 0 && L <= 16, "zest_embed_fwd: bad shape M=%d C=%d L=%d",
                   M, C, L);
    if (M == 0) return 0;
    const long long n = (long long)M * C;
    hipLaunchKernelGGL(embed_kernel, dim3(zest_div_up(n, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, x, M, C, L, y);
    ZEST_RETURN_LAUNCH("zest_embed_fwd");
}

extern "C" int zest_volume_to_cl(const float *vol, int D, int H, int W, float *vol_cl,
                                 void *stream) {
    ZEST_CHECK_ARG(vol && vol_cl && aligned16(vol_cl), "zest_volume_to_cl: bad pointer");
    ZEST_CHECK_ARG(D >= 1 && H >= 1 && W >= 1, "zest_volume_to_cl: bad shape");
    ZEST_CHECK_ARG(H <= 65535 && D <= 65535 * kTz, "zest_volume_to_cl: volume too large for the launch grid");
    hipLaunchKernelGGL(volume_to_cl_kernel, dim3(zest_div_up(W, kTx), H, zest_div_up(D, kTz)), dim3(kThreads), 0,
                       (hipStream_t)stream, vol, D, H, W, (float4 *)vol_cl);
    ZEST_RETURN_LAUNCH("zest_volume_to_cl");
}

extern "C" int zest_images_to_cl(const float *imgs, int V, int H, int W, float *imgs_cl,
                                 void *stream) {
    ZEST_CHECK_ARG(imgs && imgs_cl && aligned16(imgs_cl), "zest_images_to_cl: bad pointer");
    ZEST_CHECK_ARG(V >= 1 && H >= 1 && W >= 1, "zest_images_to_cl: bad shape");
    const long long n = (long long)V * H * W;
    hipLaunchKernelGGL(images_to_cl_kernel, dim3(zest_div_up(n, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, imgs, V, (long long)H * W, (float4 *)imgs_cl);
    ZEST_RETURN_LAUNCH("zest_images_to_cl");
}

extern "C" int zest_volume_lookup_fwd(const float *vol_cl, int D, int H, int W, const float *ndc,
                                      int M, float *out, void *stream) {
    if (M == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(vol_cl && ndc && out && aligned16(vol_cl) && aligned16(out),
                   "zest_volume_lookup_fwd: bad pointer");
    ZEST_CHECK_ARG(D >= 1 && H >= 1 && W >= 1 && M >= 0, "zest_volume_lookup_fwd: bad shape");
    if (M == 0) return 0;
    hipLaunchKernelGGL(volume_lookup_kernel, dim3(zest_div_up(M, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, (const float4 *)vol_cl, D, H, W, ndc, M, (float4 *)out);
    ZEST_RETURN_LAUNCH("zest_volume_lookup_fwd");
}

extern "C" int zest_color_lookup_fwd(const float *imgs_cl, int V, int H, int W, const float *w2cs,
                                     const float *intrinsics, const float *pts, int M, float *out,
                                     void *stream) {
    if (M == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(imgs_cl && w2cs && intrinsics && pts && out && aligned16(imgs_cl) &&
                       aligned16(out), "zest_color_lookup_fwd: bad pointer");
    ZEST_CHECK_ARG(V >= 1 && H >= 2 && W >= 2 && M >= 0, "zest_color_lookup_fwd: bad shape");
    if (M == 0) return 0;
    hipLaunchKernelGGL(color_lookup_kernel, dim3(zest_div_up(M, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, (const float4 *)imgs_cl, V, H, W, w2cs, intrinsics, pts,
                       M, (float4 *)out);
    ZEST_RETURN_LAUNCH("zest_color_lookup_fwd");
}

extern "C" int zest_encode_fwd(const float *ndc, const float *pts, const float *rays_dir, int R,
                               int S, int has_time, float t, const float *vol_cl, int D, int Hv,
                               int Wv, const float *imgs_cl, int V, int H, int W,
                               const float *w2cs, const float *intrinsics, float *x, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(ndc && rays_dir && x, "zest_encode_fwd: ndc, rays_dir and x are required");
    ZEST_CHECK_ARG(R >= 0 && S >= 1, "zest_encode_fwd: bad shape R=%d S=%d", R, S);
    if (vol_cl) {
        ZEST_CHECK_ARG(imgs_cl && pts && w2cs && intrinsics && aligned16(vol_cl) &&
                           aligned16(imgs_cl),
                       "zest_encode_fwd: features need vol_cl, imgs_cl, pts, w2cs, intrinsics");
        ZEST_CHECK_ARG(D >= 1 && Hv >= 1 && Wv >= 1 && V >= 1 && H >= 2 && W >= 2,
                       "zest_encode_fwd: bad volume/image shape");
    }
    if (R == 0) return 0;
    const int c_in = (3 + (has_time ? 1 : 0)) * 21 + (vol_cl ? 8 + 4 * V : 0) + 27;
    ZEST_CHECK_ARG(c_in <= kEncMaxC, "zest_encode_fwd: %d input channels (at most %d: 16 source views)", c_in, kEncMaxC);
    hipLaunchKernelGGL(encode_kernel, dim3(zest_div_up((long long)R * S, kEncRows)), dim3(kEncRows),
                       (size_t)kEncRows * (c_in | 1) * sizeof(float), (hipStream_t)stream, ndc, pts, rays_dir, R, S,
                       has_time ? 1 : 0, t,
                       (const float4 *)vol_cl, D, Hv, Wv, (const float4 *)imgs_cl, V, H, W, w2cs,
                       intrinsics, x);
    ZEST_RETURN_LAUNCH("zest_encode_fwd");
}
