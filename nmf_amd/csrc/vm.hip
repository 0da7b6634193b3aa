// TensoRF vector-matrix field for gfx950: density, density gradient (normals) and appearance in one
// pass over a compacted sample list, plus the matching backward (second-order path included).
//
// Replaces TensoRF.forward / TensorVMSplit._compute_densityfeature / _compute_appfeature /
// TensorBase.compute_normals + GridSampler2D.backward of the reference
// (fields/tensoRF.py:161-205,392-405; fields/tensor_base.py:66-129;
//  modules/grid_sample_Cinf.py:109-325).
//
// Data layout (HBM): every factor table is channel-last, so one bilinear tap is ONE contiguous
// run: appearance plane tap = 24 floats (96 B), density tap = 48 floats (192 B) from the packed
// table dpk = (P | dP/dx-stencil | dP/dy-stencil).  The reference's NCHW tables cost one strided
// cache line per channel per tap; here the 18 taps of a sample touch 18 short runs.  The
// derivative tables are rebuilt once per parameter update (nmf_vm_pack_density) instead of a
// conv2d over all planes on every call (grid_sample_Cinf.py:254-259).
//
// Because the sample coordinates carry no gradient, every quantity is linear in the taps:
//   sigma_feat = sum_i sum_c P_i,c L_i,c
//   g_a(i) = sum_c L_i,c DX_i,c   g_b(i) = sum_c L_i,c DY_i,c   g_w(i) = sum_c P_i,c DL_i,c
// and the backward is a pure scatter of (weight x adjoint) into the packed gradient tables, which
// nmf_vm_unpack_density_grad folds back through the transposed stencil.
#include "common.hpp"

namespace {

constexpr int CD = NMF_DENSITY_C;   // 16
constexpr int CA = NMF_APP_C;       // 24
constexpr int AD = NMF_APP_DIM;     // 24
constexpr int DP = 3 * CD;          // 48 floats per packed density texel
constexpr int DL = 2 * CD;          // 32 floats per packed density line entry

// plane i samples coordinates (MAT0[i], MAT1[i]) as (x=width, y=height); line i samples VEC[i]
// (fields/tensoRF.py:40-41)
__device__ __constant__ int MAT0[3] = {0, 0, 1};
__device__ __constant__ int MAT1[3] = {1, 2, 2};
__device__ __constant__ int VEC[3] = {2, 1, 0};

struct Tap2 {   // bilinear footprint on a [G][G][C] table
    int idx[4];  // texel index (y*G+x) or -1 when outside (zero padding)
    float w[4];
};
struct Tap1 {
    int idx[2];
    float w[2];
};

// F.grid_sample(..., mode=bilinear, padding_mode=zeros, align_corners=True): unnormalise with
// ((c+1)/2)*(size-1); weights e=1-w, s=1-n (ATen GridSamplerKernel.cpp, ApplyGridSample bilinear).
__device__ __forceinline__ Tap2 make_tap2(float u, float v, int G) {
    float ix = ((u + 1.f) * 0.5f) * (float)(G - 1);
    float iy = ((v + 1.f) * 0.5f) * (float)(G - 1);
    float fx = floorf(ix), fy = floorf(iy);
    float w = ix - fx, e = 1.f - w, n = iy - fy, s = 1.f - n;
    int x0 = (int)fx, y0 = (int)fy;
    Tap2 t;
    bool xin0 = x0 >= 0 && x0 < G, xin1 = x0 + 1 >= 0 && x0 + 1 < G;
    bool yin0 = y0 >= 0 && y0 < G, yin1 = y0 + 1 >= 0 && y0 + 1 < G;
    t.idx[0] = (xin0 && yin0) ? y0 * G + x0 : -1;            t.w[0] = e * s;   // nw
    t.idx[1] = (xin1 && yin0) ? y0 * G + x0 + 1 : -1;        t.w[1] = w * s;   // ne
    t.idx[2] = (xin0 && yin1) ? (y0 + 1) * G + x0 : -1;      t.w[2] = e * n;   // sw
    t.idx[3] = (xin1 && yin1) ? (y0 + 1) * G + x0 + 1 : -1;  t.w[3] = w * n;   // se
    return t;
}

// line [1,C,G,1] sampled at grid (0, w): x index is exactly 0 (width 1), the x+1 tap is outside.
__device__ __forceinline__ Tap1 make_tap1(float v, int G) {
    float iy = ((v + 1.f) * 0.5f) * (float)(G - 1);
    float fy = floorf(iy);
    float n = iy - fy, s = 1.f - n;
    int y0 = (int)fy;
    Tap1 t;
    t.idx[0] = (y0 >= 0 && y0 < G) ? y0 : -1;          t.w[0] = s;
    t.idx[1] = (y0 + 1 >= 0 && y0 + 1 < G) ? y0 + 1 : -1;  t.w[1] = n;
    return t;
}

__device__ __forceinline__ void normalized(const nmf_vm_params& p, const float4 x, float (&xn)[3]) {
    // fields/tensor_base.py:67
    xn[0] = (x.x - p.aabb_min[0]) * p.inv_size[0] - 1.f;
    xn[1] = (x.y - p.aabb_min[1]) * p.inv_size[1] - 1.f;
    xn[2] = (x.z - p.aabb_min[2]) * p.inv_size[2] - 1.f;
}

struct Ptrs3 {
    const float* p[3];
};
struct MPtrs3 {
    float* p[3];
};

// ------------------------------------------------------------------------------------------------
// pack: dpk[y][x] = (P, conv_x P, conv_y P), dlk[k] = (L, conv L)
// cross-correlation with zero padding 2 (F.conv2d(input, stencil, padding=2)):
//   DX[y][x] = sum_{i=0..4, j=0..4} kx[i][j] P[y+i-2][x+j-2];  kx rows 0 and 4 are zero, column 2 is zero
//   ky = kx^T
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack_plane(nmf_vm_params p, const float* __restrict__ P,
                                                    float* __restrict__ out) {
    const int G = p.grid;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over G*G*CD
    if (t >= (int64_t)G * G * CD) return;
    const int c = (int)(t % CD);
    const int x = (int)((t / CD) % G);
    const int y = (int)(t / ((int64_t)CD * G));
    auto at = [&](int yy, int xx) -> float {
        return (yy >= 0 && yy < G && xx >= 0 && xx < G) ? P[((int64_t)yy * G + xx) * CD + c] : 0.f;
    };
    float dx = 0.f, dy = 0.f;
#pragma unroll
    for (int i = 1; i <= 3; ++i) {
        const float* row = (i == 2) ? p.stencil : p.stencil_off;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j == 2) continue;
            dx += row[j] * at(y + i - 2, x + j - 2);
            dy += row[j] * at(y + j - 2, x + i - 2);     // transposed stencil
        }
    }
    float* o = out + ((int64_t)y * G + x) * DP;
    o[c] = at(y, x);
    o[CD + c] = dx;
    o[2 * CD + c] = dy;
}

__global__ void k_pack_line(nmf_vm_params p, const float* __restrict__ L, float* __restrict__ out) {
    const int G = p.grid;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * CD) return;
    const int c = t % CD, k = t / CD;
    float d = 0.f;
    // only the centre column of the y-stencil overlaps a width-1 line: ky[i][2] = kx[2][i]
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int kk = k + i - 2;
        if (i != 2 && kk >= 0 && kk < G) d += p.stencil[i] * L[kk * CD + c];
    }
    out[k * DL + c] = L[k * CD + c];
    out[k * DL + CD + c] = d;
}

// transpose of the pack: gP = gdpk.P + corr^T(gdpk.DX) + corr^T(gdpk.DY)
__global__ void __launch_bounds__(256) k_unpack_plane(nmf_vm_params p, const float* __restrict__ g,
                                                      float* __restrict__ gP) {
    const int G = p.grid;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)G * G * CD) return;
    const int c = (int)(t % CD);
    const int x = (int)((t / CD) % G);
    const int y = (int)(t / ((int64_t)CD * G));
    auto at = [&](int yy, int xx, int off) -> float {
        return (yy >= 0 && yy < G && xx >= 0 && xx < G) ? g[((int64_t)yy * G + xx) * DP + off + c] : 0.f;
    };
    // DX[Y][X] += kx[i][j] P[Y+i-2][X+j-2]  =>  gP[y][x] += kx[i][j] gDX[y-i+2][x-j+2]
    float acc = at(y, x, 0);
#pragma unroll
    for (int i = 1; i <= 3; ++i) {
        const float* row = (i == 2) ? p.stencil : p.stencil_off;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j == 2) continue;
            acc += row[j] * at(y - i + 2, x - j + 2, CD);
            acc += row[j] * at(y - j + 2, x - i + 2, 2 * CD);
        }
    }
    gP[((int64_t)y * G + x) * CD + c] = acc;
}

__global__ void k_unpack_line(nmf_vm_params p, const float* __restrict__ g, float* __restrict__ gL) {
    const int G = p.grid;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * CD) return;
    const int c = t % CD, k = t / CD;
    float acc = g[k * DL + c];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int kk = k - i + 2;
        if (i != 2 && kk >= 0 && kk < G) acc += p.stencil[i] * g[kk * DL + CD + c];
    }
    gL[k * CD + c] = acc;
}

// ------------------------------------------------------------------------------------------------
// forward: one lane per sample
// ------------------------------------------------------------------------------------------------
template <int N4>
__device__ __forceinline__ void load_run(const float* base, float (&dst)[N4 * 4]) {
    const float4* q = reinterpret_cast<const float4*>(base);
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        float4 v = q[i];
        dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
    }
}

__global__ void __launch_bounds__(256) k_vm_fwd(nmf_vm_params p, const float4* __restrict__ xyzt, int64_t M,
                                                Ptrs3 dpk, Ptrs3 dlk, Ptrs3 apl, Ptrs3 ali,
                                                const float* __restrict__ basis, float* __restrict__ sigma_feat,
                                                float* __restrict__ sigma, float* __restrict__ grad,
                                                float* __restrict__ normal, float* __restrict__ app,
                                                float* __restrict__ coef_out) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int G = p.grid;
    float xn[3];
    normalized(p, xyzt[m], xn);

    if (dpk.p[0]) {
        float sf = 0.f, g[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const Tap1 tl = make_tap1(xn[VEC[i]], G);
            float Lc[CD], DLc[CD];
#pragma unroll
            for (int c = 0; c < CD; ++c) { Lc[c] = 0.f; DLc[c] = 0.f; }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (tl.idx[t] < 0) continue;
                float run[DL];
                load_run<DL / 4>(dlk.p[i] + (int64_t)tl.idx[t] * DL, run);
#pragma unroll
                for (int c = 0; c < CD; ++c) { Lc[c] += tl.w[t] * run[c]; DLc[c] += tl.w[t] * run[CD + c]; }
            }
            const Tap2 tp = make_tap2(xn[MAT0[i]], xn[MAT1[i]], G);
            float s_pl = 0.f, s_dx = 0.f, s_dy = 0.f, s_pdl = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (tp.idx[t] < 0) continue;
                float run[DP];
                load_run<DP / 4>(dpk.p[i] + (int64_t)tp.idx[t] * DP, run);
                float a = 0.f, b = 0.f, cdy = 0.f, d = 0.f;
#pragma unroll
                for (int c = 0; c < CD; ++c) {
                    a += run[c] * Lc[c];
                    d += run[c] * DLc[c];
                    b += run[CD + c] * Lc[c];
                    cdy += run[2 * CD + c] * Lc[c];
                }
                s_pl += tp.w[t] * a; s_pdl += tp.w[t] * d; s_dx += tp.w[t] * b; s_dy += tp.w[t] * cdy;
            }
            sf += s_pl;
            g[MAT0[i]] += s_dx;
            g[MAT1[i]] += s_dy;
            g[VEC[i]] += s_pdl;
        }
        if (sigma_feat) sigma_feat[m] = sf;
        if (sigma) {
            float x = fminf(fmaxf(sf, -15.f), 1e3f) + p.density_shift;       // tensor_base.py:85
            sigma[m] = x > 20.f ? x : log1pf(expf(x));                       // F.softplus (threshold 20)
        }
        g[0] *= p.inv_size[0]; g[1] *= p.inv_size[1]; g[2] *= p.inv_size[2];
        if (grad) { grad[m * 3] = g[0]; grad[m * 3 + 1] = g[1]; grad[m * 3 + 2] = g[2]; }
        if (normal) {                                                        // tensor_base.py:128, mutils.py:8-12
            float n2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
            float inv = 1.f / sqrtf(fmaxf(n2, 1.1920929e-07f));
            normal[m * 3] = -g[0] * inv; normal[m * 3 + 1] = -g[1] * inv; normal[m * 3 + 2] = -g[2] * inv;
        }
    }

    if (apl.p[0] && (app || coef_out)) {
        float out[AD];
#pragma unroll
        for (int j = 0; j < AD; ++j) out[j] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const Tap1 tl = make_tap1(xn[VEC[i]], G);
            float La[CA];
#pragma unroll
            for (int c = 0; c < CA; ++c) La[c] = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (tl.idx[t] < 0) continue;
                float run[CA];
                load_run<CA / 4>(ali.p[i] + (int64_t)tl.idx[t] * CA, run);
#pragma unroll
                for (int c = 0; c < CA; ++c) La[c] += tl.w[t] * run[c];
            }
            const Tap2 tp = make_tap2(xn[MAT0[i]], xn[MAT1[i]], G);
            float Pa[CA];
#pragma unroll
            for (int c = 0; c < CA; ++c) Pa[c] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (tp.idx[t] < 0) continue;
                float run[CA];
                load_run<CA / 4>(apl.p[i] + (int64_t)tp.idx[t] * CA, run);
#pragma unroll
                for (int c = 0; c < CA; ++c) Pa[c] += tp.w[t] * run[c];
            }
#pragma unroll
            for (int c = 0; c < CA; ++c) Pa[c] *= La[c];                     // coefficient (tensoRF.py:204)
            if (coef_out) {
                float4* q = reinterpret_cast<float4*>(coef_out + m * (3 * CA) + i * CA);
#pragma unroll
                for (int c = 0; c < CA / 4; ++c) q[c] = make_float4(Pa[4 * c], Pa[4 * c + 1], Pa[4 * c + 2], Pa[4 * c + 3]);
            }
            if (app) {
#pragma unroll
                for (int j = 0; j < AD; ++j) {
                    const float* wrow = basis + j * (3 * CA) + i * CA;       // uniform -> scalar loads
                    float a = 0.f;
#pragma unroll
                    for (int c = 0; c < CA; ++c) a += wrow[c] * Pa[c];
                    out[j] += a;
                }
            }
        }
        if (app) {
            float4* q = reinterpret_cast<float4*>(app + m * AD);
#pragma unroll
            for (int j = 0; j < AD / 4; ++j) q[j] = make_float4(out[4 * j], out[4 * j + 1], out[4 * j + 2], out[4 * j + 3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward: brick-binned LDS accumulation.
//
// A per-sample scatter with global atomics (768 density + 432 appearance adds per sample) runs at
// ~14 G atomics/s on MI355X -- device-scope float atomics execute at the memory side -- i.e. ~90 ms
// for the 1.1 M samples of a steady-state step.  Instead the samples are counting-sorted by the
// 8x8x8-voxel brick of their lower corner (k_brick_hist / k_scan_bins / k_brick_scatter); one
// workgroup then owns one brick, accumulates every gradient that brick can touch -- three 9x9
// plane tiles (density: 48 ch, appearance: 24 ch) and three 9-entry line segments -- in 76 KB of
// LDS with ds_add_f32, and flushes the non-zero entries once (~80x fewer global atomics).
// ------------------------------------------------------------------------------------------------
constexpr int BR = 8;             // brick edge in texels
constexpr int TL = BR + 1;        // tile edge incl. the +1 halo of the bilinear footprint

__device__ __forceinline__ int axis_floor(const nmf_vm_params& p, float xn_a) {
    float ix = ((xn_a + 1.f) * 0.5f) * (float)(p.grid - 1);      // identical to make_tap*
    return (int)floorf(ix);
}

__device__ __forceinline__ int brick_of(const nmf_vm_params& p, const float (&xn)[3], int nbx) {
    int b[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int x0 = axis_floor(p, xn[a]);
        x0 = x0 < 0 ? 0 : (x0 > p.grid - 1 ? p.grid - 1 : x0);
        b[a] = x0 / BR;
    }
    return (b[2] * nbx + b[1]) * nbx + b[0];
}

__global__ void __launch_bounds__(256) k_brick_hist(nmf_vm_params p, const float4* __restrict__ xyzt, int64_t M,
                                                    int nbx, int32_t* __restrict__ counts,
                                                    int32_t* __restrict__ brick_id) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float xn[3];
    normalized(p, xyzt[m], xn);
    const int b = brick_of(p, xn, nbx);
    brick_id[m] = b;
    atomicAdd(counts + b, 1);
}

// single-workgroup exclusive scan of n int32 counts -> offsets[n+1]; also copies offsets into cursor[n]
__global__ void __launch_bounds__(1024) k_scan_bins(const int32_t* __restrict__ counts, int n,
                                                    int32_t* __restrict__ offsets, int32_t* __restrict__ cursor) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        int v = i < n ? counts[i] : 0;
        int incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const int carry = carry_s;
        const int excl = carry + woff + incl - v;
        if (i < n) { offsets[i] = excl; cursor[i] = excl; }
        __syncthreads();
        if (tid == 1023) carry_s = excl + v;
        __syncthreads();
    }
    if (tid == 0) offsets[n] = carry_s;
}

__global__ void __launch_bounds__(256) k_brick_scatter(const int32_t* __restrict__ brick_id, int64_t M,
                                                       int32_t* __restrict__ cursor, int32_t* __restrict__ perm) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int pos = atomicAdd(cursor + brick_id[m], 1);
    perm[pos] = (int32_t)m;
}

// uniform (per-sample) footprint: global texel index or -1, tile-local cell, weight
struct LTap2 {
    int g[4];
    int l[4];
    float w[4];
};

__device__ __forceinline__ LTap2 make_ltap2(float u, float v, int G, int ox, int oy) {
    float ix = ((u + 1.f) * 0.5f) * (float)(G - 1);
    float iy = ((v + 1.f) * 0.5f) * (float)(G - 1);
    float fx = floorf(ix), fy = floorf(iy);
    float w = ix - fx, e = 1.f - w, n = iy - fy, s = 1.f - n;
    int x0 = (int)fx, y0 = (int)fy;
    LTap2 t;
    t.w[0] = e * s; t.w[1] = w * s; t.w[2] = e * n; t.w[3] = w * n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int X = x0 + (k & 1), Y = y0 + (k >> 1);
        bool in = X >= 0 && X < G && Y >= 0 && Y < G;
        t.g[k] = in ? Y * G + X : -1;
        t.l[k] = (Y - oy) * TL + (X - ox);
    }
    return t;
}

constexpr int LDS_DP = 3 * TL * TL * DP;   // 11664 floats
constexpr int LDS_DL = 3 * TL * DL;        //   864
constexpr int LDS_AP = 3 * TL * TL * CA;   //  5832
constexpr int LDS_AL = 3 * TL * CA;        //   648
constexpr int LDS_TOTAL = LDS_DP + LDS_DL + LDS_AP + LDS_AL;   // 19008 floats = 76 KB
constexpr int BWD_THREADS = 512;
constexpr int BWD_WAVES = BWD_THREADS / 64;

// One workgroup per brick; each WAVE walks samples of the brick one at a time and its LANES are the
// channels of a tap (density: 48 = P|DX|DY, appearance: 24).  Consequences:
//   * every table read is one coalesced 96..192-byte run per tap,
//   * the LDS accumulators are [cell][channel]: the 24..48 active lanes of a ds_add_f32 hit
//     consecutive banks, so the atomic is conflict-free (a lane-per-sample mapping serialised
//     up to 64 ways on hot texels and ran at 0.1 LDS atomics/clk/CU -- measured 19.7 ms / 1 M samples),
//   * per-sample scalars (taps, adjoints) are wave-uniform.
__global__ void __launch_bounds__(BWD_THREADS) k_vm_bwd_brick(
    nmf_vm_params p, const float4* __restrict__ xyzt, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ bin_off, int nbx, Ptrs3 dpk, Ptrs3 dlk, Ptrs3 apl, Ptrs3 ali,
    const float* __restrict__ basis, const float* __restrict__ sigma_feat, const float* __restrict__ grad,
    const float* __restrict__ d_sigma, const float* __restrict__ d_sigma_feat, const float* __restrict__ d_normal,
    const float* __restrict__ d_app, MPtrs3 g_dpk, MPtrs3 g_dlk, MPtrs3 g_apl, MPtrs3 g_ali) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int brick = blockIdx.x;
    const int s = bin_off[brick], e = bin_off[brick + 1];
    if (s == e) return;
    const int G = p.grid;
    const int org[3] = {(brick % nbx) * BR, ((brick / nbx) % nbx) * BR, (brick / (nbx * nbx)) * BR};
    const bool has_density = dpk.p[0] && (d_sigma || d_sigma_feat || d_normal);
    const bool has_app = apl.p[0] && d_app;
    float* l_dp = lds;
    float* l_dl = l_dp + LDS_DP;
    float* l_ap = l_dl + LDS_DL;
    float* l_al = l_ap + LDS_AP;
    for (int i = threadIdx.x; i < LDS_TOTAL; i += BWD_THREADS) lds[i] = 0.f;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // appearance: lane c (<24) keeps column (i*24+c) of basis_mat for the three planes
    float Wc[3][AD];
    if (has_app) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < AD; ++j) Wc[i][j] = lane < CA ? basis[j * (3 * CA) + i * CA + lane] : 0.f;
    }
    __syncthreads();

    const int part = lane >> 4, ch = lane & 15;      // density lanes: part 0 = P, 1 = DX, 2 = DY
    const int lane_dl = lane < DL ? lane : 0, lane_dp = lane < DP ? lane : 0, lane_a = lane < CA ? lane : 0;
    for (int idx = s + wave; idx < e; idx += BWD_WAVES) {
        const int m = __builtin_amdgcn_readfirstlane(perm[idx]);
        float xn[3];
        normalized(p, xyzt[m], xn);
        // ---- footprints (wave-uniform).  Out-of-range taps are redirected to texel 0 / cell 0 with
        // weight 0, so the whole sample is branch-free and all of its table reads are in flight at once
        // (a per-tap branch + wait made this kernel latency-bound: ~40 serial round trips per sample).
        Tap1 tl[3];
        LTap2 tp[3];
        int lcell[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            tl[i] = make_tap1(xn[VEC[i]], G);
            tp[i] = make_ltap2(xn[MAT0[i]], xn[MAT1[i]], G, org[MAT0[i]], org[MAT1[i]]);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bool in = tl[i].idx[t] >= 0;
                lcell[i][t] = in ? tl[i].idx[t] - org[VEC[i]] : 0;
                tl[i].w[t] = in ? tl[i].w[t] : 0.f;
                tl[i].idx[t] = in ? tl[i].idx[t] : 0;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool in = tp[i].g[t] >= 0;
                tp[i].w[t] = in ? tp[i].w[t] : 0.f;
                tp[i].l[t] = in ? tp[i].l[t] : 0;
                tp[i].g[t] = in ? tp[i].g[t] : 0;
            }
        }
        // ---- issue every table read of this sample
        float r_dl[3][2], r_dp[3][4], r_al[3][2], r_ap[3][4];
        if (has_density) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int t = 0; t < 2; ++t) r_dl[i][t] = dlk.p[i][(int64_t)tl[i].idx[t] * DL + lane_dl];
#pragma unroll
                for (int t = 0; t < 4; ++t) r_dp[i][t] = dpk.p[i][(int64_t)tp[i].g[t] * DP + lane_dp];
            }
        }
        float da[AD];
        if (has_app) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int t = 0; t < 2; ++t) r_al[i][t] = ali.p[i][(int64_t)tl[i].idx[t] * CA + lane_a];
#pragma unroll
                for (int t = 0; t < 4; ++t) r_ap[i][t] = apl.p[i][(int64_t)tp[i].g[t] * CA + lane_a];
            }
#pragma unroll
            for (int j = 0; j < AD; ++j) da[j] = d_app[(int64_t)m * AD + j];     // uniform address
        }
        if (has_density) {
            float dsf = d_sigma_feat ? d_sigma_feat[m] : 0.f;
            if (d_sigma) {
                float f = sigma_feat[m];
                float x = fminf(fmaxf(f, -15.f), 1e3f) + p.density_shift;
                float ds = x > 20.f ? 1.f : 1.f / (1.f + expf(-x));             // softplus'
                if (f < -15.f || f > 1e3f) ds = 0.f;                            // clamp'
                dsf += d_sigma[m] * ds;
            }
            float dg[3] = {0.f, 0.f, 0.f};
            if (d_normal) {   // through n = -g / sqrt(max(|g|^2, eps))
                float g0 = grad[m * 3], g1 = grad[m * 3 + 1], g2 = grad[m * 3 + 2];
                float dn0 = d_normal[m * 3], dn1 = d_normal[m * 3 + 1], dn2 = d_normal[m * 3 + 2];
                float n2 = g0 * g0 + g1 * g1 + g2 * g2;
                const float eps = 1.1920929e-07f;
                float inv = 1.f / sqrtf(fmaxf(n2, eps));
                float dot = dn0 * g0 + dn1 * g1 + dn2 * g2;
                float k = n2 > eps ? dot * inv * inv * inv : 0.f;
                dg[0] = (-dn0 * inv + k * g0) * p.inv_size[0];
                dg[1] = (-dn1 * inv + k * g1) * p.inv_size[1];
                dg[2] = (-dn2 * inv + k * g2) * p.inv_size[2];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float dga = dg[MAT0[i]], dgb = dg[MAT1[i]], dgw = dg[VEC[i]];
                // line: lanes 0..31 = (L | DL)
                const float lv = tl[i].w[0] * r_dl[i][0] + tl[i].w[1] * r_dl[i][1];
                const float Lc = __shfl(lv, ch, 64), DLc = __shfl(lv, CD + ch, 64);
                // plane: lanes 0..47 = (P | DX | DY); adjoint of the table entry this lane owns
                const float coefL = part == 0 ? (dsf * Lc + dgw * DLc) : (part == 1 ? dga * Lc : dgb * Lc);
                float q = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) q += tp[i].w[t] * r_dp[i][t];
                if (lane < DP) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        atomicAdd(l_dp + (i * TL * TL + tp[i].l[t]) * DP + lane, tp[i].w[t] * coefL);
                }
                // line adjoints: aL_c = dsf P_c + dga DX_c + dgb DY_c (lanes 0..15), aDL_c = dgw P_c (lanes 16..31)
                const float q1 = __shfl(q, (lane + CD) & 63, 64), q2 = __shfl(q, (lane + 2 * CD) & 63, 64);
                const float qm = __shfl(q, (lane - CD) & 63, 64);
                const float la = lane < CD ? (dsf * q + dga * q1 + dgb * q2) : dgw * qm;
                if (lane < DL) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        atomicAdd(l_dl + (i * TL + lcell[i][t]) * DL + lane, tl[i].w[t] * la);
                }
            }
        }
        if (has_app) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float dcoef = 0.f;
#pragma unroll
                for (int j = 0; j < AD; ++j) dcoef += Wc[i][j] * da[j];
                const float La = tl[i].w[0] * r_al[i][0] + tl[i].w[1] * r_al[i][1];
                float pa = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) pa += tp[i].w[t] * r_ap[i][t];
                if (lane < CA) {
                    const float ap_adj = dcoef * La, al_adj = pa * dcoef;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        atomicAdd(l_ap + (i * TL * TL + tp[i].l[t]) * CA + lane, tp[i].w[t] * ap_adj);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        atomicAdd(l_al + (i * TL + lcell[i][t]) * CA + lane, tl[i].w[t] * al_adj);
                }
            }
        }
    }
    __syncthreads();

    // flush the non-zero entries (tiles of neighbouring bricks overlap on their halo -> atomics)
    if (has_density) {
        for (int k = threadIdx.x; k < LDS_DP; k += BWD_THREADS) {
            const float v = l_dp[k];
            if (v == 0.f) continue;
            const int c = k % DP, cell = (k / DP) % (TL * TL), i = k / (DP * TL * TL);
            const int X = org[MAT0[i]] + cell % TL, Y = org[MAT1[i]] + cell / TL;
            if (X < G && Y < G) atomicAdd(g_dpk.p[i] + ((int64_t)Y * G + X) * DP + c, v);
        }
        for (int k = threadIdx.x; k < LDS_DL; k += BWD_THREADS) {
            const float v = l_dl[k];
            if (v == 0.f) continue;
            const int c = k % DL, cell = (k / DL) % TL, i = k / (DL * TL);
            const int Z = org[VEC[i]] + cell;
            if (Z < G) atomicAdd(g_dlk.p[i] + (int64_t)Z * DL + c, v);
        }
    }
    if (has_app) {
        for (int k = threadIdx.x; k < LDS_AP; k += BWD_THREADS) {
            const float v = l_ap[k];
            if (v == 0.f) continue;
            const int c = k % CA, cell = (k / CA) % (TL * TL), i = k / (CA * TL * TL);
            const int X = org[MAT0[i]] + cell % TL, Y = org[MAT1[i]] + cell / TL;
            if (X < G && Y < G) atomicAdd(g_apl.p[i] + ((int64_t)Y * G + X) * CA + c, v);
        }
        for (int k = threadIdx.x; k < LDS_AL; k += BWD_THREADS) {
            const float v = l_al[k];
            if (v == 0.f) continue;
            const int c = k % CA, cell = (k / CA) % TL, i = k / (CA * TL);
            const int Z = org[VEC[i]] + cell;
            if (Z < G) atomicAdd(g_ali.p[i] + (int64_t)Z * CA + c, v);
        }
    }
}

Ptrs3 mk(const float* const a[3]) {
    Ptrs3 r;
    for (int i = 0; i < 3; ++i) r.p[i] = a ? a[i] : nullptr;
    return r;
}
MPtrs3 mkm(float* const a[3]) {
    MPtrs3 r;
    for (int i = 0; i < 3; ++i) r.p[i] = a ? a[i] : nullptr;
    return r;
}
bool all3(const float* const a[3]) { return a && a[0] && a[1] && a[2]; }
bool all3m(float* const a[3]) { return a && a[0] && a[1] && a[2]; }

}  // namespace

extern "C" int nmf_vm_pack_density(const nmf_vm_params* p, const float* const planes[3], const float* const lines[3],
                                   float* const dpk[3], float* const dlk[3], void* stream) {
    NMF_REQUIRE(p && all3(planes) && all3(lines) && all3m(dpk) && all3m(dlk), NMF_EINVAL, "nmf_vm_pack_density: null");
    NMF_REQUIRE(p->grid >= 2 && p->grid <= 4096, NMF_ERANGE, "nmf_vm_pack_density: grid");
    const int G = p->grid;
    const int64_t n = (int64_t)G * G * CD;
    for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL(k_pack_plane, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, *p, planes[i],
                           dpk[i]);
        hipLaunchKernelGGL(k_pack_line, dim3((unsigned)cdiv(G * CD, 256)), dim3(256), 0, (hipStream_t)stream, *p,
                           lines[i], dlk[i]);
    }
    NMF_CHECK_LAUNCH("nmf_vm_pack_density");
    return NMF_OK;
}

extern "C" int nmf_vm_unpack_density_grad(const nmf_vm_params* p, const float* const g_dpk[3],
                                          const float* const g_dlk[3], float* const g_planes[3],
                                          float* const g_lines[3], void* stream) {
    NMF_REQUIRE(p && all3(g_dpk) && all3(g_dlk) && all3m(g_planes) && all3m(g_lines), NMF_EINVAL,
                "nmf_vm_unpack_density_grad: null");
    const int G = p->grid;
    const int64_t n = (int64_t)G * G * CD;
    for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL(k_unpack_plane, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, *p, g_dpk[i],
                           g_planes[i]);
        hipLaunchKernelGGL(k_unpack_line, dim3((unsigned)cdiv(G * CD, 256)), dim3(256), 0, (hipStream_t)stream, *p,
                           g_dlk[i], g_lines[i]);
    }
    NMF_CHECK_LAUNCH("nmf_vm_unpack_density_grad");
    return NMF_OK;
}

extern "C" int nmf_vm_query_fwd(const nmf_vm_params* p, const float* xyzt, int64_t M, const float* const dpk[3],
                                const float* const dlk[3], const float* const app_planes[3],
                                const float* const app_lines[3], const float* basis, float* sigma_feat, float* sigma,
                                float* grad, float* normal, float* app, float* coef, void* stream) {
    NMF_REQUIRE(p && M >= 0, NMF_EINVAL, "nmf_vm_query_fwd: params");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(xyzt, NMF_EINVAL, "nmf_vm_query_fwd: xyzt null");
    const bool want_d = sigma_feat || sigma || grad || normal;
    const bool want_a = app || coef;
    NMF_REQUIRE(!want_d || (all3(dpk) && all3(dlk)), NMF_EINVAL, "nmf_vm_query_fwd: density tables missing");
    NMF_REQUIRE(!want_a || (all3(app_planes) && all3(app_lines) && (!app || basis)), NMF_EINVAL,
                "nmf_vm_query_fwd: appearance tables missing");
    hipLaunchKernelGGL(k_vm_fwd, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, *p,
                       (const float4*)xyzt, M, want_d ? mk(dpk) : mk(nullptr), want_d ? mk(dlk) : mk(nullptr),
                       want_a ? mk(app_planes) : mk(nullptr), want_a ? mk(app_lines) : mk(nullptr), basis, sigma_feat,
                       sigma, grad, normal, app, coef);
    NMF_CHECK_LAUNCH("nmf_vm_query_fwd");
    return NMF_OK;
}

extern "C" int64_t nmf_vm_bwd_workspace_bytes(int64_t M, int32_t grid) {
    const int64_t nbx = (grid + BR - 1) / BR;
    const int64_t nb = nbx * nbx * nbx;
    return (2 * M + 3 * (nb + 1)) * (int64_t)sizeof(int32_t);
}

extern "C" int nmf_vm_query_bwd(const nmf_vm_params* p, const float* xyzt, int64_t M, const float* const dpk[3],
                                const float* const dlk[3], const float* const app_planes[3],
                                const float* const app_lines[3], const float* basis, const float* sigma_feat,
                                const float* grad, const float* d_sigma, const float* d_sigma_feat,
                                const float* d_normal, const float* d_app, float* const g_dpk[3],
                                float* const g_dlk[3], float* const g_app_planes[3], float* const g_app_lines[3],
                                void* workspace, int64_t workspace_bytes, void* stream) {
    NMF_REQUIRE(p && M >= 0, NMF_EINVAL, "nmf_vm_query_bwd: params");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(xyzt, NMF_EINVAL, "nmf_vm_query_bwd: xyzt null");
    NMF_REQUIRE(M < (1ll << 31), NMF_ERANGE, "nmf_vm_query_bwd: M >= 2^31");
    const bool want_d = d_sigma || d_sigma_feat || d_normal;
    const bool want_a = d_app != nullptr;
    NMF_REQUIRE(!want_d || (all3(dpk) && all3(dlk) && all3m(g_dpk) && all3m(g_dlk)), NMF_EINVAL,
                "nmf_vm_query_bwd: density tables missing");
    NMF_REQUIRE(!d_sigma || sigma_feat, NMF_EINVAL, "nmf_vm_query_bwd: d_sigma needs saved sigma_feat");
    NMF_REQUIRE(!d_normal || grad, NMF_EINVAL, "nmf_vm_query_bwd: d_normal needs saved grad");
    NMF_REQUIRE(!want_a || (all3(app_planes) && all3(app_lines) && basis && all3m(g_app_planes) && all3m(g_app_lines)),
                NMF_EINVAL, "nmf_vm_query_bwd: appearance tables missing");
    NMF_REQUIRE(workspace && workspace_bytes >= nmf_vm_bwd_workspace_bytes(M, p->grid), NMF_EINVAL,
                "nmf_vm_query_bwd: workspace too small (see nmf_vm_bwd_workspace_bytes)");
    if (!want_d && !want_a) return NMF_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nbx = (p->grid + BR - 1) / BR;
    const int nb = nbx * nbx * nbx;
    int32_t* ws = (int32_t*)workspace;
    int32_t* brick_id = ws;            // [M]
    int32_t* perm = ws + M;            // [M]
    int32_t* counts = ws + 2 * M;      // [nb+1]
    int32_t* offsets = counts + nb + 1;   // [nb+1]
    int32_t* cursor = offsets + nb + 1;   // [nb+1]
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * (nb + 1), st);
    if (e != hipSuccess) return nmf_fail((int)e, "nmf_vm_query_bwd: memset");
    hipLaunchKernelGGL(k_brick_hist, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, st, *p, (const float4*)xyzt, M, nbx,
                       counts, brick_id);
    hipLaunchKernelGGL(k_scan_bins, dim3(1), dim3(1024), 0, st, counts, nb, offsets, cursor);
    hipLaunchKernelGGL(k_brick_scatter, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, st, brick_id, M, cursor, perm);
    // 76 KB of dynamic LDS per workgroup (> the 64 KB default cap)
    e = hipFuncSetAttribute((const void*)k_vm_bwd_brick, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(LDS_TOTAL * sizeof(float)));
    if (e != hipSuccess) return nmf_fail((int)e, "nmf_vm_query_bwd: hipFuncSetAttribute");
    hipLaunchKernelGGL(k_vm_bwd_brick, dim3((unsigned)nb), dim3(BWD_THREADS), LDS_TOTAL * sizeof(float), st, *p,
                       (const float4*)xyzt, perm, offsets, nbx, want_d ? mk(dpk) : mk(nullptr),
                       want_d ? mk(dlk) : mk(nullptr), want_a ? mk(app_planes) : mk(nullptr),
                       want_a ? mk(app_lines) : mk(nullptr), basis, sigma_feat, grad, d_sigma, d_sigma_feat, d_normal,
                       d_app, want_d ? mkm(g_dpk) : mkm(nullptr), want_d ? mkm(g_dlk) : mkm(nullptr),
                       want_a ? mkm(g_app_planes) : mkm(nullptr), want_a ? mkm(g_app_lines) : mkm(nullptr));
    NMF_CHECK_LAUNCH("nmf_vm_query_bwd");
    return NMF_OK;
}
