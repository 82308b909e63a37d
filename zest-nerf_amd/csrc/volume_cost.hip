// Plane-sweep cost volume of the MVS volume builder.
//
// Replaces MVSNet.build_volume_cost (reference networks.py:1077-1140) and utils.homo_warp
// (utils.py:49-99).  The reference materialises, per source view, a [C,D,H,W] warped feature
// volume, its square, a warped image volume and a sampling grid, and reduces them with a dozen
// elementwise passes.  Here every voxel (d, y, x) of the padded reference grid is projected into
// every source view once (homography at the plane's depth), the four bilinear taps of the
// 32-channel feature map (channels-last: 128 contiguous bytes per tap) and of the image are
// taken, the running sum / sum of squares / in-frame count stay in registers, and the 3 V image
// channels, the 32 variance channels and the V masks are written once.  Bound: the HBM write of
// the output (41 x D x Hp x Wp floats); the source maps are a few MB and stay in L2.
#include "zest_common.cuh"
#include "../../include/zest_render.h"

namespace {

constexpr int kThreads = 256;
constexpr int kC = 32;               // feature channels of FeatureNet's top level

// sum over the four lanes of a quad, result on all four: two DPP quad permutes ([1,0,3,2], [2,3,0,1]) instead of
// the two LDS-crossbar shuffles (ds_bpermute) __shfl_xor compiles to
__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    return v;
}

struct Tap4 {                        // bilinear taps, zero padding, align_corners (grid_sample)
    int off[4];                      // pixel offsets y * W + x (clamped when the weight is 0)
    float w[4];
};

// Source-view sampling position of reference pixel (xr, yr) on the plane at `depth`, following
// homo_warp's arithmetic: p = R [xr, yr, 1]^T + T / depth; (sx, sy) = p.xy / p.z; normalised to
// [-1, 1] (the in-frame mask is taken on the normalised value, strictly inside) and mapped
// back to pixels as grid_sample(align_corners=True) does.
__device__ __forceinline__ void project(const float *__restrict__ P, float xr, float yr, float depth,
                                        int H, int W, float &gx, float &gy) {
    const float X = fmaf(P[0], xr, fmaf(P[1], yr, P[2])) + P[3] / depth;
    const float Y = fmaf(P[4], xr, fmaf(P[5], yr, P[6])) + P[7] / depth;
    const float Z = fmaf(P[8], xr, fmaf(P[9], yr, P[10])) + P[11] / depth;
    gx = (X / Z) / ((float)(W - 1) / 2.0f) - 1.0f;
    gy = (Y / Z) / ((float)(H - 1) / 2.0f) - 1.0f;
}

__device__ __forceinline__ Tap4 taps(float gx, float gy, int H, int W) {
    Tap4 t;
    float px = (gx + 1.0f) * 0.5f * (float)(W - 1), py = (gy + 1.0f) * 0.5f * (float)(H - 1);
    // NaN / far-away positions (a plane behind the source camera): no contribution
    const bool finite = fabsf(px) < 1e8f && fabsf(py) < 1e8f;
    px = finite ? px : -4.0f, py = finite ? py : -4.0f;
    const float x0f = floorf(px), y0f = floorf(py);
    const float tx = px - x0f, ty = py - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int dx = c & 1, dy = c >> 1, xi = x0 + dx, yi = y0 + dy;
        const bool ok = (unsigned)xi < (unsigned)W && (unsigned)yi < (unsigned)H;
        t.w[c] = ok ? (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty) : 0.0f;
        t.off[c] = min(max(yi, 0), H - 1) * W + min(max(xi, 0), W - 1);
    }
    return t;
}

__global__ void nchw_to_nhwc_kernel(const float *__restrict__ in, int N, int C, long long npix,
                                    float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over N * npix * C
    if (i >= (long long)N * npix * C) return;
    const int c = (int)(i % C);
    const long long p = (i / C) % npix, n = i / C / npix;
    out[i] = in[((size_t)n * C + c) * npix + p];
}

// feats_cl [V,H,W,32], imgs_cl [V,H,W,4] (rgb + pad), proj [V-1,3,4], depth [D]
// img_feat [3V + 32, D, Hp, Wp], in_masks [V, D, Hp, Wp]
//
// A workgroup is one wave and owns 64 consecutive voxels.  Phase 1, one lane per voxel: the
// homographies, bilinear weights and tap offsets are computed once per voxel and left in LDS.
// Phase 2 walks the voxels 16 at a time with lane = (voxel of the group, channel octet h): a
// bilinear tap of a voxel is then one 128-byte line read by 4 neighbouring lanes (two 16-byte
// loads each, one 32-bit offset per tap), where a lane-per-voxel gather touches 64 lines per
// load instruction and uses 16 bytes of each.  The 4V + 32 output values of a voxel leave from the
// registers of its four lanes with streaming stores: a store instruction writes 64 contiguous bytes
// (the group's 16 voxels) of four planes, and the workgroup's four groups complete each plane's 256
// bytes back to back.  (Round 1 transposed them through a 12 KB LDS tile into 256-byte stores; the tile
// held the CU to 9 one-wave workgroups, and this kernel is bound by issue latency, not by either stream.)
constexpr int kMaxViews = 8;
constexpr int kClChannels = 48;        // channels of the channels-last cost volume (3 V + 32 = 41 for V = 3, padded to octets)
constexpr int kRowStride = 66;      // row stride (floats) of the backward kernel's LDS transpose tile

// Output path.  ZEST_SWEEP_TILE=1 (default): the 4V + 32 values of the 64 voxels are transposed through a 12 KB LDS
// tile so that every plane receives one 256-byte store per workgroup.  =2: a tile of 32 voxels, two rounds of
// 128-byte stores, 14 instead of 9 workgroups per CU.  =0: no tile, 64-byte stores straight from the registers
// of phase 2, 16 workgroups per CU.  Measured on one MI355X (NSFF geometry, same process, two repeats each):
// 183-186 us / 191-198 us / 205-208 us: the width of the stores matters more than the waves in flight.
#ifndef ZEST_SWEEP_TILE
#define ZEST_SWEEP_TILE 1
#endif
#if ZEST_SWEEP_TILE
#if ZEST_SWEEP_TILE == 2          // the tile holds 32 voxels: two rounds of transpose + 128-byte stores, half the LDS
constexpr int kTileStride = 34, kTileIts = 2;
#else
constexpr int kTileStride = 66, kTileIts = 4;
#endif
template <int VT>                   // VT > 0: compile-time view count (loops unroll)
__global__ __launch_bounds__(64) void volume_cost_tile_kernel(
    const float4 *__restrict__ feats, const float4 *__restrict__ imgs, const float *__restrict__ proj,
    const float *__restrict__ depth, int V_rt, int D, int H, int W, int pad, float *__restrict__ img_feat,
    float *__restrict__ in_masks, float *__restrict__ cl_out) {
    const int V = VT > 0 ? VT : V_rt;
    constexpr int VM = VT > 0 ? VT : kMaxViews;
    __shared__ float tile[(4 * VM + kC) * kTileStride];         // [rows][kTileStride]
    __shared__ int4 toff[(VM - 1) * 64];                       // bilinear tap offsets / weights per source view
    __shared__ float4 tw[(VM - 1) * 64];
    __shared__ float4 vox[64];          // ref pixel offset (-1: ring, -2: past the end), 1/count, in-frame bit mask
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const long long nvox = (long long)D * Hp * Wp;
    const int lane = threadIdx.x;
    const long long base = (long long)blockIdx.x * 64;
    {   // ---- phase 1: lane = voxel
        const long long idx = base + lane;
        const bool live = idx < nvox;
        // 32-bit index arithmetic (the host checks nvox < 2^31)
        const unsigned ic = (unsigned)(live ? idx : nvox - 1);
        const unsigned row = ic / (unsigned)Wp;
        const int x = (int)(ic - row * (unsigned)Wp), d = (int)(row / (unsigned)Hp), y = (int)(row - (unsigned)d * (unsigned)Hp);
        const int xr = x - pad, yr = y - pad;
        const bool inside = (unsigned)xr < (unsigned)W && (unsigned)yr < (unsigned)H;
        const float dep = depth[d];
        float count = 1.0f;
        int in_frame = 1;
#pragma unroll
        for (int i = 1; i < V; i++) {
            float gx, gy;
            project(proj + 12 * (i - 1), (float)xr, (float)yr, dep, H, W, gx, gy);
            const bool m = gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f;
            count += m ? 1.0f : 0.0f, in_frame |= m ? (1 << i) : 0;
            const Tap4 t = taps(gx, gy, H, W);
            toff[(i - 1) * 64 + lane] = make_int4(t.off[0], t.off[1], t.off[2], t.off[3]);
            tw[(i - 1) * 64 + lane] = make_float4(t.w[0], t.w[1], t.w[2], t.w[3]);
        }
        vox[lane] = make_float4(__int_as_float(live && inside ? yr * W + xr : -1), 1.0f / count,
                                __int_as_float(in_frame), 0.0f);
    }
    __syncthreads();
    // ---- phase 2: lane = (voxel of a group of 16, channel octet h).  Branch-free, so the four
    // groups form one basic block and the compiler keeps their loads in flight together (voxels
    // past the end of the volume compute on clamped addresses and are dropped at the final
    // store).  One 32-bit byte offset per tap; the octet's two 16-byte halves differ by an
    // immediate.
    const int v16 = lane >> 2, h = lane & 3;
    const char *fbase = reinterpret_cast<const char *>(feats);
  for (int round = 0; round < 4 / kTileIts; round++) {
#pragma unroll
    for (int it0 = 0; it0 < kTileIts; it0++) {
        const int it = round * kTileIts + it0;
        const int v = it * 16 + v16, vt = it0 * 16 + v16;       // voxel of the workgroup / of the tile
        const float4 g = vox[v];
        const int ref_off = __float_as_int(g.x);
        const bool has_ref = ref_off >= 0;
        // reference view: its own feature map, zero in the padding ring
        float4 sum[2], sq[2];
        {
            const unsigned bo = (unsigned)(has_ref ? ref_off : 0) * 128u + (unsigned)h * 32u;
            const float4 r0 = *reinterpret_cast<const float4 *>(fbase + bo);
            const float4 r1 = *reinterpret_cast<const float4 *>(fbase + bo + 16);
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            sum[0] = has_ref ? r0 : z4, sum[1] = has_ref ? r1 : z4;
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
            sq[j] = make_float4(sum[j].x * sum[j].x, sum[j].y * sum[j].y, sum[j].z * sum[j].z, sum[j].w * sum[j].w);
#pragma unroll
        for (int i = 1; i < V; i++) {
            const int4 o4 = toff[(i - 1) * 64 + v];
            const float4 w4 = tw[(i - 1) * 64 + v];
            const unsigned view_b = (unsigned)i * (unsigned)(H * W) * 128u + (unsigned)h * 32u;
            const unsigned b0 = view_b + (unsigned)o4.x * 128u, b1 = view_b + (unsigned)o4.y * 128u,
                           b2 = view_b + (unsigned)o4.z * 128u, b3 = view_b + (unsigned)o4.w * 128u;
#pragma unroll
            for (int j = 0; j < 2; j++) {
#ifdef ZEST_EXPERIMENT_NO_GATHER       // timing experiment only
                const float4 t0 = w4, t1 = w4, t2 = w4, t3 = w4;
#else
                const float4 t0 = *reinterpret_cast<const float4 *>(fbase + b0 + 16 * j),
                             t1 = *reinterpret_cast<const float4 *>(fbase + b1 + 16 * j),
                             t2 = *reinterpret_cast<const float4 *>(fbase + b2 + 16 * j),
                             t3 = *reinterpret_cast<const float4 *>(fbase + b3 + 16 * j);
#endif
                float4 a;
                a.x = fmaf(w4.w, t3.x, fmaf(w4.z, t2.x, fmaf(w4.y, t1.x, w4.x * t0.x)));
                a.y = fmaf(w4.w, t3.y, fmaf(w4.z, t2.y, fmaf(w4.y, t1.y, w4.x * t0.y)));
                a.z = fmaf(w4.w, t3.z, fmaf(w4.z, t2.z, fmaf(w4.y, t1.z, w4.x * t0.z)));
                a.w = fmaf(w4.w, t3.w, fmaf(w4.z, t2.w, fmaf(w4.y, t1.w, w4.x * t0.w)));
                sum[j].x += a.x, sum[j].y += a.y, sum[j].z += a.z, sum[j].w += a.w;
                sq[j].x = fmaf(a.x, a.x, sq[j].x), sq[j].y = fmaf(a.y, a.y, sq[j].y);
                sq[j].z = fmaf(a.z, a.z, sq[j].z), sq[j].w = fmaf(a.w, a.w, sq[j].w);
            }
            // image of this view: lane h takes tap h, two quad shuffles add the four up
            const int off = h == 0 ? o4.x : h == 1 ? o4.y : h == 2 ? o4.z : o4.w;
            const float wgt = h == 0 ? w4.x : h == 1 ? w4.y : h == 2 ? w4.z : w4.w;
            const float4 tv = imgs[(size_t)i * H * W + off];
            float bx = wgt * tv.x, by = wgt * tv.y, bz = wgt * tv.z;
            bx = quad_sum(bx), by = quad_sum(by), bz = quad_sum(bz);
            if (h == 0) {
                tile[(3 * i) * kTileStride + vt] = bx, tile[(3 * i + 1) * kTileStride + vt] = by;
                tile[(3 * i + 2) * kTileStride + vt] = bz;
                tile[(3 * V + kC + i) * kTileStride + vt] = ((__float_as_int(g.z) >> i) & 1) ? 1.0f : 0.0f;
            }
        }
        if (h == 0) {
            // the reference leaves channels 0-2 of the padding ring uninitialised (torch.empty); 0 here
            const float4 c0 = imgs[has_ref ? ref_off : 0];
            tile[0 * kTileStride + vt] = has_ref ? c0.x : 0.f, tile[1 * kTileStride + vt] = has_ref ? c0.y : 0.f;
            tile[2 * kTileStride + vt] = has_ref ? c0.z : 0.f;
            tile[(3 * V + kC) * kTileStride + vt] = 1.0f;
        }
        const float inv = g.y;
        float *o = tile + (size_t)(3 * V + 8 * h) * kTileStride + vt;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            float mean = sum[j].x * inv;
            o[(4 * j) * kTileStride] = sq[j].x * inv - mean * mean;
            mean = sum[j].y * inv, o[(4 * j + 1) * kTileStride] = sq[j].y * inv - mean * mean;
            mean = sum[j].z * inv, o[(4 * j + 2) * kTileStride] = sq[j].z * inv - mean * mean;
            mean = sum[j].w * inv, o[(4 * j + 3) * kTileStride] = sq[j].w * inv - mean * mean;
        }
    }
    __syncthreads();
#if ZEST_SWEEP_TILE == 2
    {
        // lanes 0-31 / 32-63 take even / odd rows: one instruction stores 128 bytes of two planes
        const int vl = lane & 31, rp = lane >> 5;
        const long long idx = base + round * 32 + vl;
        if (idx < nvox) {
            for (int r = rp; r < 4 * V + kC; r += 2) {
                float *dst = r < 3 * V + kC ? img_feat + (size_t)r * nvox : in_masks + (size_t)(r - 3 * V - kC) * nvox;
                __builtin_nontemporal_store(tile[r * kTileStride + vl], dst + idx);
            }
        }
    }
    __syncthreads();
  }
}
#else
    if (cl_out) {
        // channels-last for the HIP regularisation net (costreg.hip): [voxel][kClChannels], the 3 V + 32 values of a
        // voxel followed by zeros; the workgroup's 64 voxels are 64 * kClChannels consecutive floats
        const int live = (int)(nvox - base < 64 ? nvox - base : 64);
        float *dst = cl_out + (size_t)base * kClChannels;
        typedef __attribute__((ext_vector_type(4))) float f4;
#pragma unroll
        for (int k = 0; k < kClChannels / 4; k++) {                 // 16-byte stores: 1 KiB per wave instruction
            const int e = (k * 64 + lane) * 4, v = e / kClChannels, ch = e - v * kClChannels;
            f4 q;
#pragma unroll
            for (int j = 0; j < 4; j++) q[j] = ch + j < 3 * V + kC ? tile[(ch + j) * kTileStride + v] : 0.0f;
            if (v < live) __builtin_nontemporal_store(q, reinterpret_cast<f4 *>(dst + e));
        }
        return;
    }
    const long long idx = base + lane;
    if (idx < nvox) {
    for (int r = 0; r < 3 * V + kC; r++) {
        // streaming stores: 475 MB of output must not evict the few MB of feature maps from L2
        __builtin_nontemporal_store(tile[r * kTileStride + lane], &img_feat[(size_t)r * nvox + idx]);
    }
    for (int i = 0; i < V; i++)
        __builtin_nontemporal_store(tile[(3 * V + kC + i) * kTileStride + lane], &in_masks[(size_t)i * nvox + idx]);
    }
  }
}
#endif

#define volume_cost_kernel volume_cost_tile_kernel
#else
template <int VT>                   // VT > 0: compile-time view count (loops unroll)
#ifdef ZEST_SWEEP_WAVES             // occupancy experiment: cap the registers for this many waves per SIMD
__attribute__((amdgpu_waves_per_eu(ZEST_SWEEP_WAVES, ZEST_SWEEP_WAVES)))
#endif
__global__ __launch_bounds__(64) void volume_cost_kernel(
    const float4 *__restrict__ feats, const float4 *__restrict__ imgs, const float *__restrict__ proj,
    const float *__restrict__ depth, int V_rt, int D, int H, int W, int pad, float *__restrict__ img_feat,
    float *__restrict__ in_masks, float *__restrict__ /* cl_out: tile variant only */) {
    const int V = VT > 0 ? VT : V_rt;
    constexpr int VM = VT > 0 ? VT : kMaxViews;
    __shared__ int4 toff[(VM - 1) * 64];                       // bilinear tap offsets / weights per source view
    __shared__ float4 tw[(VM - 1) * 64];
    __shared__ float4 vox[64];          // ref pixel offset (-1: ring, -2: past the end), 1/count, in-frame bit mask
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const long long nvox = (long long)D * Hp * Wp;
    const int lane = threadIdx.x;
    const long long base = (long long)blockIdx.x * 64;
    {   // ---- phase 1: lane = voxel
        const long long idx = base + lane;
        const bool live = idx < nvox;
        // 32-bit index arithmetic (the host checks nvox < 2^31)
        const unsigned ic = (unsigned)(live ? idx : nvox - 1);
        const unsigned row = ic / (unsigned)Wp;
        const int x = (int)(ic - row * (unsigned)Wp), d = (int)(row / (unsigned)Hp), y = (int)(row - (unsigned)d * (unsigned)Hp);
        const int xr = x - pad, yr = y - pad;
        const bool inside = (unsigned)xr < (unsigned)W && (unsigned)yr < (unsigned)H;
        const float dep = depth[d];
        float count = 1.0f;
        int in_frame = 1;
#pragma unroll
        for (int i = 1; i < V; i++) {
            float gx, gy;
            project(proj + 12 * (i - 1), (float)xr, (float)yr, dep, H, W, gx, gy);
            const bool m = gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f;
            count += m ? 1.0f : 0.0f, in_frame |= m ? (1 << i) : 0;
            const Tap4 t = taps(gx, gy, H, W);
            toff[(i - 1) * 64 + lane] = make_int4(t.off[0], t.off[1], t.off[2], t.off[3]);
            tw[(i - 1) * 64 + lane] = make_float4(t.w[0], t.w[1], t.w[2], t.w[3]);
        }
        vox[lane] = make_float4(__int_as_float(live && inside ? yr * W + xr : -1), 1.0f / count,
                                __int_as_float(in_frame), 0.0f);
    }
    __syncthreads();
    // ---- phase 2: lane = (voxel of a group of 16, channel octet h).  Branch-free, so the four
    // groups form one basic block and the compiler keeps their loads in flight together (voxels
    // past the end of the volume compute on clamped addresses and are dropped at the final
    // store).  One 32-bit byte offset per tap; the octet's two 16-byte halves differ by an
    // immediate.
    const int v16 = lane >> 2, h = lane & 3;
    const char *fbase = reinterpret_cast<const char *>(feats);
#ifndef ZEST_SWEEP_UNROLL
#define ZEST_SWEEP_UNROLL 4         // groups of 16 voxels whose gathers are in flight together
#endif
#pragma unroll ZEST_SWEEP_UNROLL
    for (int it = 0; it < 4; it++) {
        const int v = it * 16 + v16;
        const long long idx = base + v;
        const bool live = idx < nvox;
        const float4 g = vox[v];
        const int ref_off = __float_as_int(g.x);
        const bool has_ref = ref_off >= 0;
        // reference view: its own feature map, zero in the padding ring
        float4 sum[2], sq[2];
        {
            const unsigned bo = (unsigned)(has_ref ? ref_off : 0) * 128u + (unsigned)h * 32u;
            const float4 r0 = *reinterpret_cast<const float4 *>(fbase + bo);
            const float4 r1 = *reinterpret_cast<const float4 *>(fbase + bo + 16);
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            sum[0] = has_ref ? r0 : z4, sum[1] = has_ref ? r1 : z4;
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
            sq[j] = make_float4(sum[j].x * sum[j].x, sum[j].y * sum[j].y, sum[j].z * sum[j].z, sum[j].w * sum[j].w);
#pragma unroll
        for (int i = 1; i < V; i++) {
            const int4 o4 = toff[(i - 1) * 64 + v];
            const float4 w4 = tw[(i - 1) * 64 + v];
            const unsigned view_b = (unsigned)i * (unsigned)(H * W) * 128u + (unsigned)h * 32u;
            const unsigned b0 = view_b + (unsigned)o4.x * 128u, b1 = view_b + (unsigned)o4.y * 128u,
                           b2 = view_b + (unsigned)o4.z * 128u, b3 = view_b + (unsigned)o4.w * 128u;
#pragma unroll
            for (int j = 0; j < 2; j++) {
#ifdef ZEST_EXPERIMENT_NO_GATHER       // timing experiment only
                const float4 t0 = w4, t1 = w4, t2 = w4, t3 = w4;
#else
                const float4 t0 = *reinterpret_cast<const float4 *>(fbase + b0 + 16 * j),
                             t1 = *reinterpret_cast<const float4 *>(fbase + b1 + 16 * j),
                             t2 = *reinterpret_cast<const float4 *>(fbase + b2 + 16 * j),
                             t3 = *reinterpret_cast<const float4 *>(fbase + b3 + 16 * j);
#endif
                float4 a;
                a.x = fmaf(w4.w, t3.x, fmaf(w4.z, t2.x, fmaf(w4.y, t1.x, w4.x * t0.x)));
                a.y = fmaf(w4.w, t3.y, fmaf(w4.z, t2.y, fmaf(w4.y, t1.y, w4.x * t0.y)));
                a.z = fmaf(w4.w, t3.z, fmaf(w4.z, t2.z, fmaf(w4.y, t1.z, w4.x * t0.z)));
                a.w = fmaf(w4.w, t3.w, fmaf(w4.z, t2.w, fmaf(w4.y, t1.w, w4.x * t0.w)));
                sum[j].x += a.x, sum[j].y += a.y, sum[j].z += a.z, sum[j].w += a.w;
                sq[j].x = fmaf(a.x, a.x, sq[j].x), sq[j].y = fmaf(a.y, a.y, sq[j].y);
                sq[j].z = fmaf(a.z, a.z, sq[j].z), sq[j].w = fmaf(a.w, a.w, sq[j].w);
            }
            // image of this view: lane h takes tap h, two quad shuffles add the four up (every lane of the quad
            // then holds the sums); lane h < 3 stores colour channel h, lane 3 the in-frame mask
            const int off = h == 0 ? o4.x : h == 1 ? o4.y : h == 2 ? o4.z : o4.w;
            const float wgt = h == 0 ? w4.x : h == 1 ? w4.y : h == 2 ? w4.z : w4.w;
            const float4 tv = imgs[(size_t)i * H * W + off];
            float bx = wgt * tv.x, by = wgt * tv.y, bz = wgt * tv.z;
            bx = quad_sum(bx), by = quad_sum(by), bz = quad_sum(bz);
            const float m = ((__float_as_int(g.z) >> i) & 1) ? 1.0f : 0.0f;
            if (live)
                __builtin_nontemporal_store(h == 0 ? bx : h == 1 ? by : h == 2 ? bz : m,
                                            (h < 3 ? img_feat + (size_t)(3 * i + h) * nvox : in_masks + (size_t)i * nvox) + idx);
        }
        {
            // reference view: its own colours (the reference leaves channels 0-2 of the padding ring
            // uninitialised (torch.empty); 0 here) and an all-ones mask
            const float4 c0 = imgs[has_ref ? ref_off : 0];
            const float val = !has_ref && h < 3 ? 0.f : (h == 0 ? c0.x : h == 1 ? c0.y : h == 2 ? c0.z : 1.0f);
            if (live) __builtin_nontemporal_store(val, (h < 3 ? img_feat + (size_t)h * nvox : in_masks) + idx);
        }
        // variance of the 8 channels of this lane's octet, straight from registers: one store instruction
        // writes 64 contiguous bytes (16 voxels) of four planes, and the four groups of a workgroup complete
        // each plane's 256 bytes back to back (the lines merge in L2 before they leave for HBM).  No LDS
        // transpose tile: the workgroup keeps 5 KB of LDS instead of 17, which was what limited the CU to 9 waves.
        const float inv = g.y;
        float *o = img_feat + (size_t)(3 * V + 8 * h) * nvox + idx;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float m0 = sum[j].x * inv, m1 = sum[j].y * inv, m2 = sum[j].z * inv, m3 = sum[j].w * inv;
            const float v0 = sq[j].x * inv - m0 * m0, v1 = sq[j].y * inv - m1 * m1;
            const float v2 = sq[j].z * inv - m2 * m2, v3 = sq[j].w * inv - m3 * m3;
            if (live) {
#ifdef ZEST_EXPERIMENT_NO_WRITE        // timing experiment only
                if (v0 == 123456.0f)
#endif
                {
                    __builtin_nontemporal_store(v0, o + (size_t)(4 * j) * nvox);
                    __builtin_nontemporal_store(v1, o + (size_t)(4 * j + 1) * nvox);
                    __builtin_nontemporal_store(v2, o + (size_t)(4 * j + 2) * nvox);
                    __builtin_nontemporal_store(v3, o + (size_t)(4 * j + 3) * nvox);
                }
            }
        }
    }
}

#endif

// src [C,H,W]; grid_in (optional) [D,Hp,Wp,2] normalised positions to reuse; outputs
// warped [C,D,Hp,Wp] and (when computed here) grid_out [D,Hp,Wp,2]
__global__ __launch_bounds__(kThreads) void homo_warp_kernel(
    const float *__restrict__ src, const float *__restrict__ proj, const float *__restrict__ depth,
    const float *__restrict__ grid_in, int C, int D, int H, int W, int Hp, int Wp, int pad,
    float *__restrict__ warped, float *__restrict__ grid_out) {
    const long long nvox = (long long)D * Hp * Wp;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nvox) return;
    float gx, gy;
    if (grid_in) {
        gx = grid_in[2 * idx], gy = grid_in[2 * idx + 1];
    } else {
        const unsigned row = (unsigned)idx / (unsigned)Wp;
        const int x = (int)((unsigned)idx - row * (unsigned)Wp), d = (int)(row / (unsigned)Hp), y = (int)(row - (unsigned)d * (unsigned)Hp);
        project(proj, (float)(x - pad), (float)(y - pad), depth[d], H, W, gx, gy);
        grid_out[2 * idx] = gx, grid_out[2 * idx + 1] = gy;
    }
    const Tap4 t = taps(gx, gy, H, W);
    for (int c = 0; c < C; c++) {
        const float *s = src + (size_t)c * H * W;
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) a = fmaf(t.w[k], s[t.off[k]], a);
        warped[(size_t)c * nvox + idx] = a;
    }
}

// ---- backward of the plane sweep ----------------------------------------------------------
// The trainable input of build_volume_cost is the feature maps (FeatureNet's output; the images,
// homographies and depths are data, and the in-frame counts come from comparisons: no gradient).
// Per voxel and channel, with a_i the (warped) feature of view i, n the in-frame count and
// var = sum a_i^2 / n - (sum a_i / n)^2:      d var / d a_i = (2 / n) (a_i - mean).
// The warped features are not kept by the forward pass (V-1 volumes of 350 MB each at the NSFF
// geometry): they are gathered again here.  A workgroup is one wave and owns 64 consecutive voxels,
// as in the forward kernel: phase 1 (lane = voxel) recomputes the homographies and bilinear taps
// into LDS and transposes the 32 gradient planes of its voxels through LDS (each plane read as 256
// contiguous bytes); phase 2 walks the voxels two at a time with lane = (voxel of the pair,
// channel): a bilinear tap is one 128-byte line per voxel, read - and scattered back with float
// atomics - by 32 neighbouring lanes, the access shape at which memory-side float atomics run at
// full rate (MI355X_MICROARCH.md, Global float atomics: two 128-B segments per wave-instruction).
// feats_cl [V,H,W,32], g_img_feat [3V+32, D, Hp, Wp] (only its last 32 channels are read),
// g_feats_cl [V,H,W,32] accumulated (zero it first).
template <int VT>
__global__ __launch_bounds__(64) void volume_cost_bwd_kernel(
    const float *__restrict__ feats, const float *__restrict__ proj, const float *__restrict__ depth, int V_rt,
    int D, int H, int W, int pad, const float *__restrict__ g_img_feat, float *__restrict__ g_feats) {
    const int V = VT > 0 ? VT : V_rt;
    constexpr int VM = VT > 0 ? VT : kMaxViews;
    __shared__ float gt[kC * kRowStride];                      // gradient of the variance planes [channel][voxel]
    __shared__ int4 toff[(VM - 1) * 64];
    __shared__ float4 tw[(VM - 1) * 64];
    __shared__ float2 vox[64];                                  // ref pixel offset (-1: none), 1/count
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const long long nvox = (long long)D * Hp * Wp;
    const int lane = threadIdx.x;
    const long long base = (long long)blockIdx.x * 64;
    {   // ---- phase 1: lane = voxel
        const long long idx = base + lane;
        const bool live = idx < nvox;
        const unsigned ic = (unsigned)(live ? idx : nvox - 1);
        const unsigned row = ic / (unsigned)Wp;
        const int x = (int)(ic - row * (unsigned)Wp), d = (int)(row / (unsigned)Hp), y = (int)(row - (unsigned)d * (unsigned)Hp);
        const int xr = x - pad, yr = y - pad;
        const bool inside = (unsigned)xr < (unsigned)W && (unsigned)yr < (unsigned)H;
        const float dep = depth[d];
        float count = 1.0f;
#pragma unroll
        for (int i = 1; i < V; i++) {
            float gx, gy;
            project(proj + 12 * (i - 1), (float)xr, (float)yr, dep, H, W, gx, gy);
            count += (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.0f : 0.0f;
            const Tap4 t = taps(gx, gy, H, W);
            toff[(i - 1) * 64 + lane] = make_int4(t.off[0], t.off[1], t.off[2], t.off[3]);
            // a voxel past the end of the volume scatters nothing
            tw[(i - 1) * 64 + lane] = live ? make_float4(t.w[0], t.w[1], t.w[2], t.w[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        vox[lane] = make_float2(__int_as_float(live && inside ? yr * W + xr : -1), 1.0f / count);
        const size_t plane0 = (size_t)(3 * V) * nvox;
        for (int c = 0; c < kC; c++)
            gt[c * kRowStride + lane] = live ? g_img_feat[plane0 + (size_t)c * nvox + idx] : 0.0f;
    }
    __syncthreads();
    // ---- phase 2: lane = (voxel of a pair, channel)
    const int c = lane & 31, vp = lane >> 5;
    const size_t HW = (size_t)H * W;
    for (int it = 0; it < 32; it++) {
        const int v = 2 * it + vp;
        const float2 gv = vox[v];
        const int ref_off = __float_as_int(gv.x);
        const float g = gt[c * kRowStride + v], inv = gv.y;
        float a[VM];
        a[0] = ref_off >= 0 ? feats[(size_t)ref_off * kC + c] : 0.0f;
        float sum = a[0];
#pragma unroll
        for (int i = 1; i < V; i++) {
            const int4 o4 = toff[(i - 1) * 64 + v];
            const float4 w4 = tw[(i - 1) * 64 + v];
            const float *f = feats + (size_t)i * HW * kC + c;
            a[i] = fmaf(w4.w, f[(size_t)o4.w * kC], fmaf(w4.z, f[(size_t)o4.z * kC],
                        fmaf(w4.y, f[(size_t)o4.y * kC], w4.x * f[(size_t)o4.x * kC])));
            sum += a[i];
        }
        const float mean = sum * inv, k = 2.0f * inv * g;
        if (ref_off >= 0) atomicAdd(g_feats + (size_t)ref_off * kC + c, k * (a[0] - mean));
#pragma unroll
        for (int i = 1; i < V; i++) {
            const int4 o4 = toff[(i - 1) * 64 + v];
            const float4 w4 = tw[(i - 1) * 64 + v];
            float *gf = g_feats + (size_t)i * HW * kC + c;
            const float ga = k * (a[i] - mean);
            // wave-uniform per half: a tap with weight 0 (outside the frame) is skipped by all 32 lanes
            if (w4.x != 0.0f) atomicAdd(gf + (size_t)o4.x * kC, w4.x * ga);
            if (w4.y != 0.0f) atomicAdd(gf + (size_t)o4.y * kC, w4.y * ga);
            if (w4.z != 0.0f) atomicAdd(gf + (size_t)o4.z * kC, w4.z * ga);
            if (w4.w != 0.0f) atomicAdd(gf + (size_t)o4.w * kC, w4.w * ga);
        }
    }
}

// backward of homo_warp with respect to the source map: g_warped [C,D,Hp,Wp] -> g_src [C,H,W]
// (accumulated: zero it first).  The sampling positions (grid or homography + depths) are data.
__global__ __launch_bounds__(kThreads) void homo_warp_bwd_kernel(
    const float *__restrict__ proj, const float *__restrict__ depth, const float *__restrict__ grid_in, int C, int D,
    int H, int W, int Hp, int Wp, int pad, const float *__restrict__ g_warped, float *__restrict__ g_src) {
    const long long nvox = (long long)D * Hp * Wp;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nvox) return;
    float gx, gy;
    if (grid_in) {
        gx = grid_in[2 * idx], gy = grid_in[2 * idx + 1];
    } else {
        const unsigned row = (unsigned)idx / (unsigned)Wp;
        const int x = (int)((unsigned)idx - row * (unsigned)Wp), d = (int)(row / (unsigned)Hp), y = (int)(row - (unsigned)d * (unsigned)Hp);
        project(proj, (float)(x - pad), (float)(y - pad), depth[d], H, W, gx, gy);
    }
    const Tap4 t = taps(gx, gy, H, W);
    for (int c = 0; c < C; c++) {
        const float g = g_warped[(size_t)c * nvox + idx];
        float *s = g_src + (size_t)c * H * W;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (t.w[k] != 0.0f) atomicAdd(s + t.off[k], t.w[k] * g);
    }
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int zest_nchw_to_nhwc(const float *in, int N, int C, int H, int W, float *out, void *stream) {
    ZEST_CHECK_ARG(in && out, "zest_nchw_to_nhwc: null pointer");
    ZEST_CHECK_ARG(N >= 1 && C >= 1 && H >= 1 && W >= 1, "zest_nchw_to_nhwc: bad shape");
    const long long n = (long long)N * C * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(zest_div_up(n, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, in, N, C, (long long)H * W, out);
    ZEST_RETURN_LAUNCH("zest_nchw_to_nhwc");
}

// img_feat + in_masks (the reference's planes) or, cl_out given, the channels-last copy for costreg.hip alone
static int volume_cost_launch(const char *who, const float *feats_cl, const float *imgs_cl, const float *proj,
                              const float *depth, int V, int C, int D, int H, int W, int pad, float *img_feat,
                              float *in_masks, float *cl_out, void *stream) {
    ZEST_CHECK_ARG(feats_cl && imgs_cl && proj && depth && ((img_feat && in_masks) || cl_out) && aligned16(feats_cl) &&
                       aligned16(imgs_cl), "%s: bad pointer", who);
    ZEST_CHECK_ARG(C == kC, "%s: %d feature channels (the FeatureNet top level has %d)", who, C, kC);
    ZEST_CHECK_ARG(V >= 2 && D >= 1 && H >= 2 && W >= 2 && pad >= 0, "%s: bad shape", who);
    ZEST_CHECK_ARG(V <= kMaxViews, "%s: at most %d views (LDS transpose tile), got %d", who, kMaxViews, V);
    ZEST_CHECK_ARG(!cl_out || (3 * V + kC <= kClChannels && ZEST_SWEEP_TILE == 1),
                   "%s: the channels-last volume holds %d channels, %d views need %d", who, kClChannels, V, 3 * V + kC);
    const long long nvox = (long long)D * (H + 2 * pad) * (W + 2 * pad);
    ZEST_CHECK_ARG(nvox < (1ll << 31), "%s: %lld voxels exceed the 32-bit index range", who, nvox);
    ZEST_CHECK_ARG((long long)V * H * W * 128 < (1ll << 32), "%s: feature maps of %d x %d x %d "
                   "pixels exceed the 32-bit byte offsets of the gather", who, V, H, W);
#define ZEST_SWEEP(VT)                                                                               \
    hipLaunchKernelGGL(volume_cost_kernel<VT>, dim3(zest_div_up(nvox, 64)), dim3(64), 0,                 \
                       (hipStream_t)stream, (const float4 *)feats_cl, (const float4 *)imgs_cl, proj, depth, \
                       V, D, H, W, pad, img_feat, in_masks, cl_out)
    if (V == 3) ZEST_SWEEP(3);          // the shipped configurations: reference + 2 source views
    else if (V == 4) ZEST_SWEEP(4);
    else ZEST_SWEEP(0);
#undef ZEST_SWEEP
    ZEST_RETURN_LAUNCH(who);
}

extern "C" int zest_volume_cost_fwd(const float *feats_cl, const float *imgs_cl, const float *proj,
                                    const float *depth, int V, int C, int D, int H, int W, int pad,
                                    float *img_feat, float *in_masks, void *stream) {
    ZEST_CHECK_ARG(img_feat && in_masks, "zest_volume_cost_fwd: bad pointer");
    return volume_cost_launch("zest_volume_cost_fwd", feats_cl, imgs_cl, proj, depth, V, C, D, H, W, pad, img_feat,
                              in_masks, nullptr, stream);
}

extern "C" int zest_volume_cost_cl_fwd(const float *feats_cl, const float *imgs_cl, const float *proj,
                                       const float *depth, int V, int C, int D, int H, int W, int pad,
                                       float *cost_cl, void *stream) {
    ZEST_CHECK_ARG(cost_cl && aligned16(cost_cl), "zest_volume_cost_cl_fwd: bad pointer");
    return volume_cost_launch("zest_volume_cost_cl_fwd", feats_cl, imgs_cl, proj, depth, V, C, D, H, W, pad, nullptr,
                              nullptr, cost_cl, stream);
}

extern "C" int zest_homo_warp_fwd(const float *src, const float *proj, const float *depth,
                                  const float *grid_in, int C, int D, int H, int W, int Hp, int Wp, int pad,
                                  float *warped, float *grid_out, void *stream) {
    ZEST_CHECK_ARG(src && warped, "zest_homo_warp_fwd: null pointer");
    ZEST_CHECK_ARG(grid_in || (proj && depth && grid_out),
                   "zest_homo_warp_fwd: either a grid or projection + depths + grid output are needed");
    ZEST_CHECK_ARG(C >= 1 && D >= 1 && H >= 2 && W >= 2 && Hp >= 1 && Wp >= 1 && pad >= 0,
                   "zest_homo_warp_fwd: bad shape");
    ZEST_CHECK_ARG(grid_in || (Hp == H + 2 * pad && Wp == W + 2 * pad),
                   "zest_homo_warp_fwd: padded grid %dx%d does not match %dx%d + 2*%d", Hp, Wp, H, W, pad);
    const long long nvox = (long long)D * Hp * Wp;
    ZEST_CHECK_ARG(nvox < (1ll << 31), "zest_homo_warp_fwd: %lld voxels exceed the 32-bit index range", nvox);
    hipLaunchKernelGGL(homo_warp_kernel, dim3(zest_div_up(nvox, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, src, proj, depth, grid_in, C, D, H, W, Hp, Wp, pad, warped, grid_out);
    ZEST_RETURN_LAUNCH("zest_homo_warp_fwd");
}

extern "C" int zest_volume_cost_bwd(const float *feats_cl, const float *proj, const float *depth, int V, int C,
                                    int D, int H, int W, int pad, const float *g_img_feat, float *g_feats_cl,
                                    void *stream) {
    ZEST_CHECK_ARG(feats_cl && proj && depth && g_img_feat && g_feats_cl, "zest_volume_cost_bwd: null pointer");
    ZEST_CHECK_ARG(C == kC, "zest_volume_cost_bwd: %d feature channels (the FeatureNet top level has %d)", C, kC);
    ZEST_CHECK_ARG(V >= 2 && V <= kMaxViews && D >= 1 && H >= 2 && W >= 2 && pad >= 0, "zest_volume_cost_bwd: bad shape");
    const long long nvox = (long long)D * (H + 2 * pad) * (W + 2 * pad);
    ZEST_CHECK_ARG(nvox < (1ll << 31), "zest_volume_cost_bwd: %lld voxels exceed the 32-bit index range", nvox);
#define ZEST_SWEEP(VT)                                                                                   \
    hipLaunchKernelGGL(volume_cost_bwd_kernel<VT>, dim3(zest_div_up(nvox, 64)), dim3(64), 0, (hipStream_t)stream, \
                       feats_cl, proj, depth, V, D, H, W, pad, g_img_feat, g_feats_cl)
    if (V == 3) ZEST_SWEEP(3);
    else if (V == 4) ZEST_SWEEP(4);
    else ZEST_SWEEP(0);
#undef ZEST_SWEEP
    ZEST_RETURN_LAUNCH("zest_volume_cost_bwd");
}

extern "C" int zest_homo_warp_bwd(const float *proj, const float *depth, const float *grid_in, int C, int D, int H,
                                  int W, int Hp, int Wp, int pad, const float *g_warped, float *g_src, void *stream) {
    ZEST_CHECK_ARG(g_warped && g_src, "zest_homo_warp_bwd: null pointer");
    ZEST_CHECK_ARG(grid_in || (proj && depth), "zest_homo_warp_bwd: either a grid or projection + depths are needed");
    ZEST_CHECK_ARG(C >= 1 && D >= 1 && H >= 2 && W >= 2 && Hp >= 1 && Wp >= 1 && pad >= 0, "zest_homo_warp_bwd: bad shape");
    ZEST_CHECK_ARG(grid_in || (Hp == H + 2 * pad && Wp == W + 2 * pad), "zest_homo_warp_bwd: padded grid does not match");
    const long long nvox = (long long)D * Hp * Wp;
    ZEST_CHECK_ARG(nvox < (1ll << 31), "zest_homo_warp_bwd: %lld voxels exceed the 32-bit index range", nvox);
    hipLaunchKernelGGL(homo_warp_bwd_kernel, dim3(zest_div_up(nvox, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                       proj, depth, grid_in, C, D, H, W, Hp, Wp, pad, g_warped, g_src);
    ZEST_RETURN_LAUNCH("zest_homo_warp_bwd");
}
