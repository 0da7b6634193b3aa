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
#include <stdlib.h>

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
// contract(off): `ix - floor(ix)` must see the ROUNDED product in every kernel that inlines this -- left to the optimiser
// it is fused into an fma in some kernels and not in others (identical at G - 1 = 2^k, one ulp apart otherwise), and
// k_vm_sigma / k_vm_rows_dn / k_vm_app_rows promise the bits of k_vm_fwd
__device__ __forceinline__ Tap2 make_tap2(float u, float v, int G) {
#pragma clang fp contract(off)
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
#pragma clang fp contract(off)
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
#pragma clang fp contract(off)
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
// uniform select instead of p[i]: indexing a by-value kernel argument dynamically would spill it to scratch
__device__ __forceinline__ const float* pick3(const Ptrs3& a, int i) { return i == 0 ? a.p[0] : (i == 1 ? a.p[1] : a.p[2]); }
__device__ __forceinline__ float* pick3(const MPtrs3& a, int i) { return i == 0 ? a.p[0] : (i == 1 ? a.p[1] : a.p[2]); }

// ------------------------------------------------------------------------------------------------
// pack: dpk[y][x] = (P, conv_x P, conv_y P), dlk[k] = (L, conv L)
// cross-correlation with zero padding 2 (F.conv2d(input, stencil, padding=2)):
//   DX[y][x] = sum_{i=0..4, j=0..4} kx[i][j] P[y+i-2][x+j-2];  kx rows 0 and 4 are zero, column 2 is zero
//   ky = kx^T
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_plane(const nmf_vm_params& p, const float* __restrict__ P,
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

__device__ __forceinline__ void pack_line(const nmf_vm_params& p, const float* __restrict__ L, float* __restrict__ out) {
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

// all six tables of the field in ONE launch: blockIdx.y = 0..2 plane i, 3..5 line i (the line blocks beyond G*CD/256 exit)
struct Pack6 {
    const float* src[6];
    float* dst[6];
};
__global__ void __launch_bounds__(256) k_pack_tables(nmf_vm_params p, Pack6 a) {
    const int y = blockIdx.y;
    const float* src = y == 0 ? a.src[0] : y == 1 ? a.src[1] : y == 2 ? a.src[2] : y == 3 ? a.src[3] : y == 4 ? a.src[4] : a.src[5];
    float* dst = y == 0 ? a.dst[0] : y == 1 ? a.dst[1] : y == 2 ? a.dst[2] : y == 3 ? a.dst[3] : y == 4 ? a.dst[4] : a.dst[5];
    if (y < 3) pack_plane(p, src, dst);
    else pack_line(p, src, dst);
}

// transpose of the pack: gP = gdpk.P + corr^T(gdpk.DX) + corr^T(gdpk.DY)
// x (optional, with l1): the parameter itself in the same storage order -- the gradient of l1[0] * mean |x| is added in the same
// pass (the training step's density_L1 term: fields/tensoRF.py:332-340; one launch less on the serial tail of a step)
__device__ __forceinline__ float l1_term(const float* __restrict__ x, const float* __restrict__ l1, int64_t idx, int64_t n) {
    if (!x) return 0.f;
    const float s = l1[0] / (float)n, v = x[idx];
    return v > 0.f ? s : (v < 0.f ? -s : 0.f);
}
__device__ __forceinline__ void unpack_plane(const nmf_vm_params& p, const float* __restrict__ g,
                                             float* __restrict__ gP, const float* __restrict__ xp, const float* __restrict__ l1) {
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
    const int64_t idx = ((int64_t)y * G + x) * CD + c;
    const float gv = l1_term(xp, l1, idx, (int64_t)G * G * CD);
    gP[idx] = xp ? acc + gv : acc;
}

__device__ __forceinline__ void unpack_line(const nmf_vm_params& p, const float* __restrict__ g, float* __restrict__ gL,
                                            const float* __restrict__ xp, const float* __restrict__ l1) {
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
    const float gv = l1_term(xp, l1, (int64_t)k * CD + c, (int64_t)G * CD);
    gL[k * CD + c] = xp ? acc + gv : acc;
}

struct L1Six {
    const float* x[6];      // the six density parameters (planes 0-2, lines 3-5), or all NULL
    const float* l1;
};
__global__ void __launch_bounds__(256) k_unpack_tables(nmf_vm_params p, Pack6 a, L1Six q) {
    const int y = blockIdx.y;
    const float* src = y == 0 ? a.src[0] : y == 1 ? a.src[1] : y == 2 ? a.src[2] : y == 3 ? a.src[3] : y == 4 ? a.src[4] : a.src[5];
    float* dst = y == 0 ? a.dst[0] : y == 1 ? a.dst[1] : y == 2 ? a.dst[2] : y == 3 ? a.dst[3] : y == 4 ? a.dst[4] : a.dst[5];
    const float* xp = y == 0 ? q.x[0] : y == 1 ? q.x[1] : y == 2 ? q.x[2] : y == 3 ? q.x[3] : y == 4 ? q.x[4] : q.x[5];
    if (y < 3) unpack_plane(p, src, dst, xp, q.l1);
    else unpack_line(p, src, dst, xp, q.l1);
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
// bf16 tables (BASELINE configs[1]): the same runs at half the bytes; a bf16 is the upper half of the fp32 bit pattern,
// all arithmetic stays fp32.  Runs are 16-byte aligned for every table (16 / 24 / 32 / 48 channels).
template <int N4>
__device__ __forceinline__ void load_run(const uint16_t* base, float (&dst)[N4 * 4]) {
    const uint2* q = reinterpret_cast<const uint2*>(base);
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        const uint2 v = q[i];
        dst[4 * i] = __uint_as_float(v.x << 16); dst[4 * i + 1] = __uint_as_float(v.x & 0xffff0000u);
        dst[4 * i + 2] = __uint_as_float(v.y << 16); dst[4 * i + 3] = __uint_as_float(v.y & 0xffff0000u);
    }
}
template <class TT> struct PtrsT3 { const TT* p[3]; };

template <class TT>
__global__ void __launch_bounds__(256) k_vm_fwd(nmf_vm_params p, const float4* __restrict__ xyzt, int64_t M,
                                                PtrsT3<TT> dpk, PtrsT3<TT> dlk, PtrsT3<TT> apl, PtrsT3<TT> ali,
                                                const float* __restrict__ basis, float* __restrict__ sigma_feat,
                                                float* __restrict__ sigma, float* __restrict__ grad,
                                                float* __restrict__ normal, float* __restrict__ app,
                                                float* __restrict__ coef_out, const int64_t* __restrict__ M_live) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // basis_mat through LDS (broadcast reads) when appearance vectors are formed: as uniform global addresses its 1728 entries
    // became 280 scalar loads with 96 waits on the scalar cache per thread (R4, found in the ISA: ~8 us of latency per launch)
    __shared__ float s_basis[AD * 3 * CA];
    const bool use_basis = apl.p[0] && app;
    if (use_basis) {
        for (int i = threadIdx.x; i < AD * 3 * CA; i += 256) s_basis[i] = basis[i];
        __syncthreads();
    }
    // M_live: the sample count still lives on the device (the launch was sized by a bound: nmf_vm_query_fwd_live)
    if (m >= M || (M_live && m >= *M_live)) return;
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
                for (int c = 0; c < CD; ++c) { Lc[c] = fmaf(tl.w[t], run[c], Lc[c]); DLc[c] = fmaf(tl.w[t], run[CD + c], DLc[c]); }   // explicit fma throughout: k_vm_sigma / k_vm_rows_dn repeat these sums bit for bit
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
                    a = fmaf(run[c], Lc[c], a);
                    d = fmaf(run[c], DLc[c], d);
                    b = fmaf(run[CD + c], Lc[c], b);
                    cdy = fmaf(run[2 * CD + c], Lc[c], cdy);
                }
                s_pl = fmaf(tp.w[t], a, s_pl); s_pdl = fmaf(tp.w[t], d, s_pdl); s_dx = fmaf(tp.w[t], b, s_dx);
                s_dy = fmaf(tp.w[t], cdy, s_dy);
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
                    const float* wrow = s_basis + j * (3 * CA) + i * CA;
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

// ---- brick binning helpers (the backward walks sort their samples by brick) --------------------------------------------------------
constexpr int BASIS_COPIES = 16;   // scratch copies of the basis_mat gradient (power of two), see vm_bwd_app2
constexpr int BR = 4;             // brick edge in texels (R2: 4 -> 5x5 = 25 tile cells = 2 MFMA row blocks instead of 6)
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

// Samples arrive in (ray, step) order, so consecutive lanes mostly fall into the same brick: one atomic per RUN of equal
// brick ids inside a wave instead of one per sample (~10x fewer contended atomics on the hot surface bricks).
struct RunInfo {
    bool head;
    int len, off;   // run length (valid on the head lane) and this lane's offset inside its run
};
__device__ __forceinline__ RunInfo wave_runs(int key, bool active) {
    const int lane = lane_id();
    const int prev = __shfl_up(key, 1, 64);
    const bool head = active && (lane == 0 || prev != key || !__shfl_up((int)active, 1, 64));
    const uint64_t H = __ballot(head);
    const uint64_t A = __ballot(active);
    RunInfo r;
    r.head = head;
    const uint64_t below = H & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int my_head = below ? 63 - __clzll(below) : lane;
    const uint64_t above = (lane == 63) ? 0ull : (H >> (lane + 1));
    int next_head = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
    // a run ends at the next head or at the first inactive lane
    const uint64_t inact_above = (lane == 63) ? 0ull : ((~A) >> (lane + 1));
    const int next_inact = inact_above ? lane + 1 + (__ffsll((long long)inact_above) - 1) : 64;
    next_head = next_head < next_inact ? next_head : next_inact;
    r.len = next_head - my_head;
    r.off = lane - my_head;
    return r;
}

// Counter copies: the hot bricks (the visible surface) are few and neighbours share 128-byte lines, and L2 executes the
// atomics of one line one after the other -- 0.24 M runs of the 0.88 M secondary-ray samples on ~100 hot lines cost 84 us
// in the histogram and 90 us in the scatter.  Every brick therefore has KC counters (wave w of the launch uses copy
// w % KC, the same wave in both kernels); the scan runs over the flat [brick][copy] array, so copy k of a brick owns the
// slice of the brick's segment that follows copies < k.
__device__ __forceinline__ int bin_copy(int kc) { return (int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (kc - 1)); }


// ------------------------------------------------------------------------------------------------
// forward, density value only (no gradient / normal): the samples of the re-traced rays need normals on their bounce rows
// alone (nmf_amd/fast_step.py, "sparse normals"), so the 0.9 M-sample query of a training level reads the value third of
// every texel (16 of 48 floats) and of every line entry (16 of 32).  Same taps, same order of the sums as k_vm_fwd, the
// value path of both written with explicit fma: sigma_feat and sigma are identical bits.
// ------------------------------------------------------------------------------------------------
template <class TT>
__global__ void __launch_bounds__(256) k_vm_sigma(nmf_vm_params p, const float4* __restrict__ xyzt, int64_t M,
                                                  PtrsT3<TT> dpk, PtrsT3<TT> dlk, int plane_stride, int line_stride,
                                                  float* __restrict__ sigma_feat, float* __restrict__ sigma) {
    // plane_stride / line_stride: elements per texel / line entry of the tables handed in -- DP / DL for the packed value +
    // derivative tables, CD for the density factors themselves (nmf_vm_query_sigma: a third of the cache lines)
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int G = p.grid;
    float xn[3];
    normalized(p, xyzt[m], xn);
    float sf = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const Tap1 tl = make_tap1(xn[VEC[i]], G);
        float Lc[CD];
#pragma unroll
        for (int c = 0; c < CD; ++c) Lc[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int id_ = max(tl.idx[t], 0);                 // no branch around the loads: a tap outside the table reads
            const float w_ = tl.idx[t] < 0 ? 0.f : tl.w[t];      // entry 0 with weight 0 (the sums keep their bits)
            float run[CD];
            load_run<CD / 4>(dlk.p[i] + (int64_t)id_ * line_stride, run);
#pragma unroll
            for (int c = 0; c < CD; ++c) Lc[c] = fmaf(w_, run[c], Lc[c]);
        }
        const Tap2 tp = make_tap2(xn[MAT0[i]], xn[MAT1[i]], G);
        float s_pl = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int id_ = max(tp.idx[t], 0);                 // no branch around the loads: a tap outside the table reads
            const float w_ = tp.idx[t] < 0 ? 0.f : tp.w[t];      // entry 0 with weight 0 (the sums keep their bits)
            float run[CD];
            load_run<CD / 4>(dpk.p[i] + (int64_t)id_ * plane_stride, run);
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < CD; ++c) a = fmaf(run[c], Lc[c], a);
            s_pl = fmaf(w_, a, s_pl);
        }
        sf += s_pl;
    }
    if (sigma_feat) sigma_feat[m] = sf;
    if (sigma) {
        float x = fminf(fmaxf(sf, -15.f), 1e3f) + p.density_shift;       // tensor_base.py:85
        sigma[m] = x > 20.f ? x : log1pf(expf(x));                       // F.softplus (threshold 20)
    }
}

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const uint16_t* p) { return __uint_as_float((uint32_t)(*p) << 16); }

// ------------------------------------------------------------------------------------------------
// forward, value + gradient + normal of a FEW rows (the bounce rows of a re-traced level: ~20 k): SIXTEEN lanes per row,
// lane 4 i + t owns tap t of plane i (lanes 12..15 idle): it forms the plane's line factors and its tap's four channel
// sums exactly as k_vm_fwd does, the four taps of a plane and then the three planes are combined IN k_vm_fwd's ORDER
// through shuffles -- identical bits, three dependent load batches per lane instead of eighteen.  A lane per row
// (k_vm_fwd) leaves these launches at 70 workgroups and 39 us.
// ------------------------------------------------------------------------------------------------
template <class TT>
__global__ void __launch_bounds__(256) k_vm_rows_dn(nmf_vm_params p, const float4* __restrict__ xyzt, int64_t M,
                                                    PtrsT3<TT> dpk, PtrsT3<TT> dlk, float* __restrict__ sigma_feat,
                                                    float* __restrict__ sigma, float* __restrict__ grad,
                                                    float* __restrict__ normal) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15, i = sub >> 2, t = sub & 3;
    const bool ok = row < M && i < 3;
    const int G = p.grid;
    float xn[3] = {0.f, 0.f, 0.f};
    if (row < M) normalized(p, xyzt[row], xn);
    float a = 0.f, b = 0.f, cdy = 0.f, d = 0.f, w = 0.f;
    bool valid = false;
    if (ok) {
        const int vec = i == 0 ? 2 : (i == 1 ? 1 : 0), m0 = i == 2 ? 1 : 0, m1 = i == 0 ? 1 : 2;      // VEC / MAT0 / MAT1
        const TT* lines = i == 0 ? dlk.p[0] : (i == 1 ? dlk.p[1] : dlk.p[2]);
        const TT* planes = i == 0 ? dpk.p[0] : (i == 1 ? dpk.p[1] : dpk.p[2]);
        const Tap1 tl = make_tap1(xn[vec], G);
        float Lc[CD], DLc[CD];
#pragma unroll
        for (int c = 0; c < CD; ++c) { Lc[c] = 0.f; DLc[c] = 0.f; }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (tl.idx[q] < 0) continue;
            float run[DL];
            load_run<DL / 4>(lines + (int64_t)tl.idx[q] * DL, run);
#pragma unroll
            for (int c = 0; c < CD; ++c) { Lc[c] = fmaf(tl.w[q], run[c], Lc[c]); DLc[c] = fmaf(tl.w[q], run[CD + c], DLc[c]); }
        }
        const Tap2 tp = make_tap2(xn[m0], xn[m1], G);
        const int idx = t == 0 ? tp.idx[0] : (t == 1 ? tp.idx[1] : (t == 2 ? tp.idx[2] : tp.idx[3]));
        w = t == 0 ? tp.w[0] : (t == 1 ? tp.w[1] : (t == 2 ? tp.w[2] : tp.w[3]));
        valid = idx >= 0;
        if (valid) {
            float run[DP];
            load_run<DP / 4>(planes + (int64_t)idx * DP, run);
#pragma unroll
            for (int c = 0; c < CD; ++c) {
                a = fmaf(run[c], Lc[c], a);
                d = fmaf(run[c], DLc[c], d);
                b = fmaf(run[CD + c], Lc[c], b);
                cdy = fmaf(run[2 * CD + c], Lc[c], cdy);
            }
        }
    }
    // the four taps of this lane's plane, in tap order
    const int base = (threadIdx.x & 63) & ~3;
    float s_pl = 0.f, s_dx = 0.f, s_dy = 0.f, s_pdl = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float wq = __shfl(w, base + q, 64), aq = __shfl(a, base + q, 64), dq = __shfl(d, base + q, 64);
        const float bq = __shfl(b, base + q, 64), cq = __shfl(cdy, base + q, 64);
        const bool vq = __shfl((int)valid, base + q, 64) != 0;
        if (vq) {
            s_pl = fmaf(wq, aq, s_pl); s_pdl = fmaf(wq, dq, s_pdl); s_dx = fmaf(wq, bq, s_dx); s_dy = fmaf(wq, cq, s_dy);
        }
    }
    // the three planes, in plane order, on lane 0 of the row
    const int r0 = (threadIdx.x & 63) & ~15;
    float sf = 0.f, g[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const float spl = __shfl(s_pl, r0 + 4 * pl, 64), sdx = __shfl(s_dx, r0 + 4 * pl, 64);
        const float sdy = __shfl(s_dy, r0 + 4 * pl, 64), spd = __shfl(s_pdl, r0 + 4 * pl, 64);
        sf += spl;
        g[MAT0[pl]] += sdx;
        g[MAT1[pl]] += sdy;
        g[VEC[pl]] += spd;
    }
    if (row >= M || sub != 0) return;
    if (sigma_feat) sigma_feat[row] = sf;
    if (sigma) {
        float x = fminf(fmaxf(sf, -15.f), 1e3f) + p.density_shift;       // tensor_base.py:85
        sigma[row] = x > 20.f ? x : log1pf(expf(x));                     // F.softplus (threshold 20)
    }
    g[0] *= p.inv_size[0]; g[1] *= p.inv_size[1]; g[2] *= p.inv_size[2];
    if (grad) { grad[row * 3] = g[0]; grad[row * 3 + 1] = g[1]; grad[row * 3 + 2] = g[2]; }
    if (normal) {                                                        // tensor_base.py:128, mutils.py:8-12
        float n2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
        float inv = 1.f / sqrtf(fmaxf(n2, 1.1920929e-07f));
        normal[row * 3] = -g[0] * inv; normal[row * 3 + 1] = -g[1] * inv; normal[row * 3 + 2] = -g[2] * inv;
    }
}

// ------------------------------------------------------------------------------------------------
// forward, appearance only: EIGHT lanes per sample.  The sparse-appearance path queries the appearance features of the
// bounce rows alone (8 k - 20 k rows per level): with a lane per sample that is 30 - 70 workgroups on 256 CUs, each lane
// running 108 loads, 430 tap FMAs and the 72 -> 24 basis product one after the other (27 us per launch whatever the row
// count, profiles/r02_c).  Here lane q of a row owns the channels 3q .. 3q+2 of every tap (a 96-byte run per tap and
// row), parks its 9 coefficients in LDS, and then computes the outputs 3q .. 3q+2 from the row's 72 coefficients with the
// basis matrix staged in LDS.  Tap order and the c-order of the basis product are those of k_vm_fwd: identical bits.
// ------------------------------------------------------------------------------------------------

constexpr int APP_ROWS = 32;              // rows per 256-thread workgroup
constexpr int APP_PITCH = 3 * CA + 4;     // LDS pitch of a coefficient row (76: rows of a wave start on different banks)
template <class TT>
__global__ void __launch_bounds__(256) k_vm_app_rows(nmf_vm_params p, const float4* __restrict__ xyzt, int64_t M,
                                                     PtrsT3<TT> apl, PtrsT3<TT> ali, const float* __restrict__ basis,
                                                     float* __restrict__ app) {
    __shared__ float s_basis[AD * 3 * CA];
    __shared__ float s_coef[APP_ROWS * APP_PITCH];
    for (int i = threadIdx.x; i < AD * 3 * CA; i += 256) s_basis[i] = basis[i];
    const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
    const int64_t m = (int64_t)blockIdx.x * APP_ROWS + r;
    const int G = p.grid;
    if (m < M) {
        float xn[3];
        normalized(p, xyzt[m], xn);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const Tap1 tl = make_tap1(xn[VEC[i]], G);
            float La[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int id_ = max(tl.idx[t], 0);                 // no branch around the loads: a tap outside the table reads
                const float w_ = tl.idx[t] < 0 ? 0.f : tl.w[t];      // entry 0 with weight 0 (the sums keep their bits)
                const TT* run = ali.p[i] + (int64_t)id_ * CA + 3 * q;
#pragma unroll
                for (int k = 0; k < 3; ++k) La[k] += w_ * ld1(run + k);
            }
            const Tap2 tp = make_tap2(xn[MAT0[i]], xn[MAT1[i]], G);
            float Pa[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int id_ = max(tp.idx[t], 0);                 // no branch around the loads: a tap outside the table reads
                const float w_ = tp.idx[t] < 0 ? 0.f : tp.w[t];      // entry 0 with weight 0 (the sums keep their bits)
                const TT* run = apl.p[i] + (int64_t)id_ * CA + 3 * q;
#pragma unroll
                for (int k = 0; k < 3; ++k) Pa[k] += w_ * ld1(run + k);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) s_coef[r * APP_PITCH + i * CA + 3 * q + k] = Pa[k] * La[k];     // tensoRF.py:204
        }
    }
    __syncthreads();
    if (m >= M) return;
    float out[3] = {0.f, 0.f, 0.f};
    const float* cf = s_coef + r * APP_PITCH;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CA; ++c) {
            const float v = cf[i * CA + c];
#pragma unroll
            for (int k = 0; k < 3; ++k) a[k] += s_basis[(3 * q + k) * (3 * CA) + i * CA + c] * v;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] += a[k];
    }
    float* o = app + m * AD + 3 * q;
    o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
}

// ------------------------------------------------------------------------------------------------
// backward: brick-binned LDS accumulation.
//
// A per-sample scatter with global atomics (768 density + 432 appearance adds per sample) runs at
// ~14 G atomics/s on MI355X -- device-scope float atomics execute at the memory side -- i.e. ~90 ms
// for the 1.1 M samples of a steady-state step.  Instead the samples are counting-sorted by the
// BR^3-voxel brick of their lower corner (k_plan_hist / k_bins_partial+k_bins_final / k_plan_place, then k_brick_records);
// a wave then owns a slice of one brick and accumulates every gradient that brick can touch -- three
// TLxTL plane tiles (density: 48 ch, appearance: 24 ch) and three TL-entry line segments -- ON THE
// MATRIX CORES (see k_vm_bwd_brick), flushing the non-zero entries once.  BR = 4 (TL = 5): a 25-cell
// tile is 2 MFMA row blocks of 16, against 6 for the 81 cells of an 8^3 brick, so the walk issues a
// third of the matrix instructions per sample for ~2x the flush atomics (measured 498 -> 318 us).
// ------------------------------------------------------------------------------------------------
// The samples of one walk may come from up to four caller arrays ("segments": the sample sets of the primary and of the
// re-traced rays of one training pass are walked together, so that bricks both touch are flushed once).  Sample m of
// the concatenation lives in segment sg at row m - start[sg].
constexpr int MAX_SEG = NMF_VM_MAX_SEGMENTS;
struct Segs {
    const float* xyzt[MAX_SEG];
    const float* sigma_feat[MAX_SEG];
    const float* grad[MAX_SEG];
    const float* d_sigma[MAX_SEG];
    const float* d_sigma_feat[MAX_SEG];
    const float* d_normal[MAX_SEG];
    const float* d_app[MAX_SEG];
    int64_t start[MAX_SEG + 1];
};
template <class T>
__device__ __forceinline__ T pick4(const T (&a)[MAX_SEG], int k) {
    return k == 0 ? a[0] : (k == 1 ? a[1] : (k == 2 ? a[2] : a[3]));
}
__device__ __forceinline__ int seg_of(const Segs& sg, int64_t m, int64_t& local) {
    const int k = (m >= sg.start[1] ? 1 : 0) + (m >= sg.start[2] ? 1 : 0) + (m >= sg.start[3] ? 1 : 0);
    local = m - (k == 0 ? sg.start[0] : (k == 1 ? sg.start[1] : (k == 2 ? sg.start[2] : sg.start[3])));
    return k;
}

// ---- the sort as a PLAN (R4): it depends on the sample positions alone, which the forward knows ----------------------
// The three walks of a training step used to redo histogram + scan + atomic scatter inside the backward (300 us of kernel
// time per step, 80 of them in the serial tail of the step).  nmf_vm_bin_plan runs these once, as soon as the positions
// exist (on a side stream under the forward), with ONE atomic pass: the counter add of the histogram already returns the
// sample's rank inside its (brick, counter copy), so slot[m] = start(brick, copy) + rank once the scan has run -- no
// second pass of cursor atomics.  What is left for the backward is k_brick_records: one permuted 16-byte store per sample.
__global__ void __launch_bounds__(256) k_plan_hist(nmf_vm_params p, Segs sg, int64_t M, int nbx, int kc,
                                                   int32_t* __restrict__ counts, int2* __restrict__ keyrank) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = m < M;
    int b = -1;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        int64_t l;
        const int k = seg_of(sg, m, l);
        float xn[3];
        x = reinterpret_cast<const float4*>(pick4(sg.xyzt, k))[l];
        normalized(p, x, xn);
        b = brick_of(p, xn, nbx);
    }
    const RunInfo r = wave_runs(b, active);
    const int key = b * kc + bin_copy(kc);
    int base = 0;
    if (r.head) base = atomicAdd(counts + key, r.len);
    base = __shfl(base, lane_id() - r.off, 64);
    if (active) keyrank[m] = make_int2(key, base + r.off);
}

__global__ void __launch_bounds__(256) k_plan_place(Segs sg, int64_t M, const int2* __restrict__ keyrank,
                                                    const int32_t* __restrict__ cursor, int32_t* __restrict__ slot,
                                                    float4* __restrict__ rec0) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int2 kr = keyrank[m];
    const int pos = cursor[kr.x] + kr.y;
    slot[m] = pos;
    int64_t l;
    const int k = seg_of(sg, m, l);
    rec0[pos] = reinterpret_cast<const float4*>(pick4(sg.xyzt, k))[l];
}

// Exclusive scan of the n brick counts -> offsets[n+1], the scatter cursors, and the work-item list of the backward walk:
// brick b with c samples becomes ceil(c / item) items (b, t) covering samples [offsets[b] + t*item, +item).  Only
// non-empty bricks produce items, items are equally sized, and consecutive items land on consecutive XCDs, so the hot
// surface bricks are spread over the whole chip instead of following the brick index -> XCD round-robin.
// Two launches over chunks of 4096 bricks (R2: with 4^3 bricks there are 32 k bricks at 128^3 and 422 k at 300^3; the
// single-workgroup scan of round 1 took 57 us / 372 us there): k_bins_partial sums each chunk (samples | items packed in
// one int64), k_bins_final lets every chunk add up the totals before it and scan itself.
constexpr int SB_THREADS = 1024, SB_PER = 4, SB_CHUNK = SB_THREADS * SB_PER;

__device__ __forceinline__ int brick_count(const int32_t* __restrict__ counts, int i, int kc, int (&ck)[8]) {
    int c = 0;
    if (kc == 4) {
        const int4 v4 = reinterpret_cast<const int4*>(counts)[i];
        ck[0] = v4.x; ck[1] = v4.y; ck[2] = v4.z; ck[3] = v4.w;
        c = v4.x + v4.y + v4.z + v4.w;
    } else {
        for (int k = 0; k < kc; ++k) { ck[k] = counts[i * kc + k]; c += ck[k]; }
    }
    return c;
}

__device__ __forceinline__ int64_t block_sum_i64(int64_t v, int64_t* ws) {       // all threads get the block total
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    if (lane == 0) ws[wid] = v;
    __syncthreads();
    int64_t t = 0;
    for (int w = 0; w < SB_THREADS / 64; ++w) t += ws[w];
    __syncthreads();
    return t;
}

__global__ void __launch_bounds__(SB_THREADS) k_bins_partial(const int32_t* __restrict__ counts, int n, int kc, int item,
                                                             int64_t* __restrict__ chunk_tot) {
    __shared__ int64_t ws[SB_THREADS / 64];
    const int i0 = blockIdx.x * SB_CHUNK + threadIdx.x * SB_PER;
    int64_t v = 0;
    int ck[8];
#pragma unroll
    for (int q = 0; q < SB_PER; ++q) {
        if (i0 + q < n) {
            const int c = brick_count(counts, i0 + q, kc, ck);
            v += (int64_t)c | ((int64_t)((c + item - 1) / item) << 32);
        }
    }
    const int64_t t = block_sum_i64(v, ws);
    if (threadIdx.x == 0) chunk_tot[blockIdx.x] = t;
}

__global__ void __launch_bounds__(SB_THREADS) k_bins_final(const int32_t* __restrict__ counts, int n, int kc,
                                                           const int64_t* __restrict__ chunk_tot,
                                                           int32_t* __restrict__ offsets, int32_t* __restrict__ cursor,
                                                           int item, int2* __restrict__ items,
                                                           int32_t* __restrict__ n_items) {
    __shared__ int64_t ws[SB_THREADS / 64];
    __shared__ int64_t wsum[SB_THREADS / 64];
    __shared__ int32_t s_io[SB_CHUNK + 1];       // first work item of every brick of this chunk (R4: the item list is written in parallel)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // totals of the chunks before this one (at most a few hundred values)
    int64_t before = 0;
    for (int c = tid; c < (int)blockIdx.x; c += SB_THREADS) before += chunk_tot[c];
    const int64_t carry = block_sum_i64(before, ws);
    const int i0 = blockIdx.x * SB_CHUNK + tid * SB_PER;
    int c[SB_PER], ni[SB_PER], ck[SB_PER][8];
    int64_t v = 0;
#pragma unroll
    for (int q = 0; q < SB_PER; ++q) {
        c[q] = 0;
        if (i0 + q < n) c[q] = brick_count(counts, i0 + q, kc, ck[q]);
        ni[q] = (c[q] + item - 1) / item;
        v += (int64_t)c[q] | ((int64_t)ni[q] << 32);
    }
    int64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        int64_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    int64_t run = carry + woff + incl - v;
#pragma unroll
    for (int q = 0; q < SB_PER; ++q) {
        if (i0 + q < n) {
            const int so = (int)(run & 0xffffffffll), io = (int)(run >> 32);
            offsets[i0 + q] = so;
            int pos = so;
            for (int k = 0; k < kc; ++k) { cursor[(i0 + q) * kc + k] = pos; pos += ck[q][k]; }
            s_io[tid * SB_PER + q] = io;
        } else {
            s_io[tid * SB_PER + q] = (int)(run >> 32);
        }
        run += (int64_t)c[q] | ((int64_t)ni[q] << 32);
    }
    if (tid == SB_THREADS - 1) s_io[SB_CHUNK] = (int)(run >> 32);
    __syncthreads();
    // the work items of this chunk's bricks, a lane per item: a thread per BRICK wrote its items one after the other, and the few
    // surface bricks that hold thousands of samples made the launch 15 us long (32 k counters: a 3 us job)
    {
        const int first = s_io[0], last = s_io[SB_CHUNK];
        for (int j = first + tid; j < last; j += SB_THREADS) {
            int lo = 0, hi = SB_CHUNK;                 // the LAST brick b of the chunk with s_io[b] <= j (empty bricks share a value)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_io[mid] <= j) lo = mid; else hi = mid;
            }
            items[j] = make_int2(blockIdx.x * SB_CHUNK + lo, j - s_io[lo]);
        }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == SB_THREADS - 1) {       // the last thread of the last chunk holds the grand total
        offsets[n] = (int)(run & 0xffffffffll);
        *n_items = (int)(run >> 32);
    }
}

// The same scan in ONE launch (R4): chunks of 1024 bricks on 256-thread workgroups, the totals of the chunks before a chunk
// obtained by look-back -- a workgroup takes its chunk number from a ticket (so every chunk before it belongs to a workgroup that
// has started), publishes its own total (samples | items << 32 | ready bit, one relaxed device-scope 64-bit store: the word is flag
// and payload at once, no fence) and wave 0 then reads the words of its predecessors, spinning on the ready bit.  Against
// k_bins_partial + k_bins_final (8 workgroups of 1024 threads at 128^3, 4.7 + 13.4 us with nothing to sort): one launch less per
// walk and 32 workgroups instead of 8.  state[0] = ticket, state[1 + c] = word of chunk c (zeroed with the counters).
constexpr int SC_THREADS = 256, SC_PER = 4, SC_CHUNK = SC_THREADS * SC_PER;
constexpr unsigned long long SC_READY = 1ull << 63;

// clean: the counters are handed back ZERO (every workgroup clears the counters of its own chunk once it has read them; the chunk
// words are cleared by the launch that follows, see k_place_records) -- a caller that keeps the scratch between calls then needs no
// memset launch in front of the next histogram (nmf_vm_query_bwd_segments_clean).
__global__ void __launch_bounds__(SC_THREADS) k_bins_scan(int32_t* counts, int n, int kc,
                                                          unsigned long long* __restrict__ state, int32_t* __restrict__ offsets,
                                                          int32_t* __restrict__ cursor, int item, int2* __restrict__ items,
                                                          int32_t* __restrict__ n_items, int clean) {
    __shared__ int64_t wsum[SC_THREADS / 64];
    __shared__ int64_t s_carry;
    __shared__ int s_chunk;
    __shared__ int32_t s_io[SC_CHUNK + 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) s_chunk = (int)atomicAdd(&state[0], 1ull);
    __syncthreads();
    const int chunk = s_chunk;
    const int i0 = chunk * SC_CHUNK + tid * SC_PER;
    int c[SC_PER], ni[SC_PER], ck[SC_PER][8];
    int64_t v = 0;
#pragma unroll
    for (int q = 0; q < SC_PER; ++q) {
        c[q] = 0;
        if (i0 + q < n) c[q] = brick_count(counts, i0 + q, kc, ck[q]);
        ni[q] = (c[q] + item - 1) / item;
        v += (int64_t)c[q] | ((int64_t)ni[q] << 32);
    }
    if (clean) {
#pragma unroll
        for (int q = 0; q < SC_PER; ++q) {
            if (i0 + q < n) {
                if (kc == 4) reinterpret_cast<int4*>(counts)[i0 + q] = make_int4(0, 0, 0, 0);
                else for (int k = 0; k < kc; ++k) counts[(i0 + q) * kc + k] = 0;
            }
        }
    }
    int64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int64_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SC_THREADS / 64; ++w) {
        if (w < wid) woff += wsum[w];
        total += wsum[w];
    }
    if (wid == 0) {
        if (lane == 0)
            __hip_atomic_store(&state[1 + chunk], (unsigned long long)total | SC_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int64_t carry = 0;
        for (int base = chunk - 1; base >= 0; base -= 64) {
            const int cb = base - lane;
            unsigned long long w = SC_READY;
            if (cb >= 0) {
                do {
                    w = __hip_atomic_load(&state[1 + cb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while (!(w & SC_READY));
            }
            int64_t t = (int64_t)(w & ~SC_READY);
            for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d, 64);
            carry += __shfl(t, 0, 64);
        }
        if (lane == 0) s_carry = carry;
    }
    __syncthreads();
    int64_t run = s_carry + woff + incl - v;
    int so[SC_PER];
#pragma unroll
    for (int q = 0; q < SC_PER; ++q) {
        so[q] = (int)(run & 0xffffffffll);
        s_io[tid * SC_PER + q] = (int)(run >> 32);
        run += (int64_t)c[q] | ((int64_t)ni[q] << 32);
    }
    if (i0 + SC_PER <= n) {        // (i0 is a multiple of 4 and the arrays are 16-byte aligned)
        *reinterpret_cast<int4*>(offsets + i0) = make_int4(so[0], so[1], so[2], so[3]);
        if (kc == 4) {
#pragma unroll
            for (int q = 0; q < SC_PER; ++q) {
                const int a = so[q], b = a + ck[q][0], cc = b + ck[q][1], d = cc + ck[q][2];
                reinterpret_cast<int4*>(cursor)[i0 + q] = make_int4(a, b, cc, d);
            }
        } else {
#pragma unroll
            for (int q = 0; q < SC_PER; ++q) {
                int pos = so[q];
                for (int k = 0; k < kc; ++k) { cursor[(i0 + q) * kc + k] = pos; pos += ck[q][k]; }
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < SC_PER; ++q) {
            if (i0 + q < n) {
                offsets[i0 + q] = so[q];
                int pos = so[q];
                for (int k = 0; k < kc; ++k) { cursor[(i0 + q) * kc + k] = pos; pos += ck[q][k]; }
            }
        }
    }
    if (tid == SC_THREADS - 1) s_io[SC_CHUNK] = (int)(run >> 32);
    __syncthreads();
    {   // the work items of this chunk's bricks, a lane per item (see k_bins_final)
        const int first = s_io[0], last = s_io[SC_CHUNK];
        for (int j = first + tid; j < last; j += SC_THREADS) {
            int lo = 0, hi = SC_CHUNK;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_io[mid] <= j) lo = mid; else hi = mid;
            }
            items[j] = make_int2(chunk * SC_CHUNK + lo, j - s_io[lo]);
        }
    }
    if (chunk == (int)gridDim.x - 1 && tid == SC_THREADS - 1) {
        offsets[n] = (int)(run & 0xffffffffll);
        *n_items = (int)(run >> 32);
    }
}

// adjoint of the raw density feature and of the raw density gradient (normalised-coordinate units) of sample l of segment k:
// the softplus / normalize backward, once per sample
// One sample's adjoint inputs, and what is made of them: two functions so that a caller can put its own loads between the two
// (with each load next to its use the placing kernel waited four times in a row).
struct AdjIn { float dsf, f, dsv, g0, g1, g2, dn0, dn1, dn2; };
__device__ __forceinline__ AdjIn sample_adjoint_load(const Segs& sg, int k, int64_t l) {
    AdjIn a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* d_sigma_feat = pick4(sg.d_sigma_feat, k);
    if (d_sigma_feat) a.dsf = d_sigma_feat[l];
    if (sg.d_sigma[0]) {
        a.f = pick4(sg.sigma_feat, k)[l];
        a.dsv = pick4(sg.d_sigma, k)[l];
    }
    if (sg.d_normal[0]) {
        const float* grad = pick4(sg.grad, k);
        const float* d_normal = pick4(sg.d_normal, k);
        a.g0 = grad[l * 3], a.g1 = grad[l * 3 + 1], a.g2 = grad[l * 3 + 2];
        a.dn0 = d_normal[l * 3], a.dn1 = d_normal[l * 3 + 1], a.dn2 = d_normal[l * 3 + 2];
    }
    return a;
}
__device__ __forceinline__ float4 sample_adjoint_finish(const nmf_vm_params& p, const Segs& sg, const AdjIn& a) {
    float dsf = a.dsf;
    if (sg.d_sigma[0]) {
        const float f = a.f;
        const float x = fminf(fmaxf(f, -15.f), 1e3f) + p.density_shift;
        float ds = x > 20.f ? 1.f : 1.f / (1.f + expf(-x));          // softplus'
        if (f < -15.f || f > 1e3f) ds = 0.f;                         // clamp'
        dsf += a.dsv * ds;
    }
    float dg0 = 0.f, dg1 = 0.f, dg2 = 0.f;
    if (sg.d_normal[0]) {   // through n = -g / sqrt(max(|g|^2, eps))
        const float g0 = a.g0, g1 = a.g1, g2 = a.g2, dn0 = a.dn0, dn1 = a.dn1, dn2 = a.dn2;
        const float n2 = g0 * g0 + g1 * g1 + g2 * g2;
        const float eps = 1.1920929e-07f;
        const float inv = 1.f / sqrtf(fmaxf(n2, eps));
        const float dot = dn0 * g0 + dn1 * g1 + dn2 * g2;
        const float kk = n2 > eps ? dot * inv * inv * inv : 0.f;
        dg0 = (-dn0 * inv + kk * g0) * p.inv_size[0];
        dg1 = (-dn1 * inv + kk * g1) * p.inv_size[1];
        dg2 = (-dn2 * inv + kk * g2) * p.inv_size[2];
    }
    return make_float4(dsf, dg0, dg1, dg2);
}
__device__ __forceinline__ float4 sample_adjoint(const nmf_vm_params& p, const Segs& sg, int k, int64_t l) {
    return sample_adjoint_finish(p, sg, sample_adjoint_load(sg, k, l));
}

// The per-sample inputs of the backward walk, written in SORTED order so that the brick kernels stream them without an
// indirection: rec1 = (adjoint of the raw density feature, adjoint of the raw density gradient); APP: the d_app row (its
// 72 coefficient adjoints follow in k_dcoef).  slot[] and the sorted positions rec0 come from the plan.
template <bool APP>
__global__ void __launch_bounds__(256) k_brick_records(nmf_vm_params p, Segs sg, const int32_t* __restrict__ slot, int64_t M,
                                                       float4* __restrict__ rec1, float* __restrict__ d_app_sorted) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int pos = slot[m];
    int64_t l;
    const int k = seg_of(sg, m, l);
    rec1[pos] = sample_adjoint(p, sg, k, l);
    if constexpr (APP) {
        float da[AD];
        load_run<AD / 4>(pick4(sg.d_app, k) + l * AD, da);
        float4* srt = reinterpret_cast<float4*>(d_app_sorted + (int64_t)pos * AD);
#pragma unroll
        for (int q = 0; q < AD / 4; ++q) srt[q] = make_float4(da[4 * q], da[4 * q + 1], da[4 * q + 2], da[4 * q + 3]);
    }
}

// place + records in one pass, for a walk that sorts inside its own call (no plan from the forward): the slot of a sample is used
// where it is computed, so slot[] is neither written nor read and the launch between the two is gone
template <bool APP>
__global__ void __launch_bounds__(256) k_place_records(nmf_vm_params p, Segs sg, int64_t M, const int2* __restrict__ keyrank,
                                                       const int32_t* __restrict__ cursor, float4* __restrict__ rec0,
                                                       float4* __restrict__ rec1, float* __restrict__ d_app_sorted,
                                                       unsigned long long* __restrict__ scan_state, int n_state) {
    // (scan_state: the ticket and chunk words of the scan that ran in front of this launch, cleared for the next walk on this scratch)
    if (scan_state && blockIdx.x == 0)
        for (int i = threadIdx.x; i < n_state; i += 256) scan_state[i] = 0ull;
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    // what does not depend on the slot is requested first: keyrank -> cursor is a chain of two round trips, the sample's position and
    // adjoints one more next to it (they used to follow it: four)
    const int2 kr = keyrank[m];
    int64_t l;
    const int k = seg_of(sg, m, l);
    const float4 x = reinterpret_cast<const float4*>(pick4(sg.xyzt, k))[l];
    const AdjIn ain = sample_adjoint_load(sg, k, l);
    float da[APP ? AD : 1];
    if constexpr (APP) load_run<AD / 4>(pick4(sg.d_app, k) + l * AD, da);
    const int pos = cursor[kr.x] + kr.y;
    const float4 adj = sample_adjoint_finish(p, sg, ain);
    rec0[pos] = x;
    rec1[pos] = adj;
    if constexpr (APP) {
        float4* srt = reinterpret_cast<float4*>(d_app_sorted + (int64_t)pos * AD);
#pragma unroll
        for (int q = 0; q < AD / 4; ++q) srt[q] = make_float4(da[4 * q], da[4 * q + 1], da[4 * q + 2], da[4 * q + 3]);
    }
}

// adjoint of the 72 plane*line coefficients: dcoef[pos] = d_app[pos] x basis_mat, one thread per (row, 4 coefficients) --
// the appearance walk has few rows (the bounce points), a thread per row left the chip idle for 30 us
__global__ void __launch_bounds__(256) k_dcoef(const float* __restrict__ d_app_sorted, const float* __restrict__ basis,
                                               int64_t M, float* __restrict__ dcoef) {
    constexpr int Q = 3 * CA / 4;            // 18 float4 groups per row
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * Q) return;
    const int64_t pos = t / Q;
    const int c4 = (int)(t - pos * Q);
    const float* da = d_app_sorted + pos * AD;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < AD; ++q) {
        const float a = da[q];
        const float4 b = *reinterpret_cast<const float4*>(basis + q * (3 * CA) + 4 * c4);
        v[0] += b.x * a; v[1] += b.y * a; v[2] += b.z * a; v[3] += b.w * a;
    }
    reinterpret_cast<float4*>(dcoef + pos * (3 * CA))[c4] = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ void __launch_bounds__(256) k_basis_reduce(float* copies, float* __restrict__ g_basis, int clean) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= AD * 3 * CA) return;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < BASIS_COPIES; ++c) a += copies[c * (AD * 3 * CA) + t];
    if (clean) {
#pragma unroll
        for (int c = 0; c < BASIS_COPIES; ++c) copies[c * (AD * 3 * CA) + t] = 0.f;
    }
    if (a != 0.f) g_basis[t] += a;          // the caller's accumulator: this launch is the only writer in stream order
}

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int BWD_THREADS = 64;               // one single-wave workgroup per (brick, part, plane/line pair)
constexpr int NRB = (TL * TL + 15) / 16;     // tile cells -> row blocks of 16 (BR 8: 81 -> 6, BR 4: 25 -> 2)
constexpr int BWD_ITEM = 256;                // samples per work item (brick slice); 128 below 400 k samples
constexpr int BWD_ITEM_MIN = 64;             // smallest item size the workspace is sized for (tuning knob)

// Scatter-add on the matrix cores.
//
// Inside one BR^3 brick every gradient tile is small and dense (a TLxTL plane tile x 48 / 24 channels, a TL-entry line
// segment x 32 / 24 channels), and the update   G[cell][ch] += w(sample, cell) * adj(sample, ch)   is the product of a
// sparse [cells x samples] weight matrix (4 non-zeros per column) with a dense [samples x channels] adjoint matrix.
// LDS float atomics run at ~0.2 lane-ops/clk/CU on gfx950 (measured: 17 ms per 1 M samples, 36 ds_add_f32 per sample),
// so the accumulation is done with v_mfma_f32_16x16x4_f32 instead: exact fp32, K = 4 samples per instruction, the
// NRB x 3 (+2) accumulator tiles of a plane live in AGPRs for the whole work item, no atomics until the final flush.
// The 95 % zero products are free: the matrix pipe is otherwise idle here.
//
// Lane mapping of a wave: k = lane >> 4 is the sample of the current group of 4, j = lane & 15 is a channel (B operand)
// and, for the A operand, a tile cell.  One single-wave workgroup owns one (work item, plane/line pair, density or
// appearance) and walks the item's samples in sorted order.
// C layout of 16x16x4: column (channel) = lane & 15, row (cell) = 4 * (lane >> 4) + reg
__device__ __forceinline__ void flush_plane_tile(const floatx4 (&acc)[NRB], float* __restrict__ g, int nch, int ch,
                                                 bool ch_ok, int ox, int oy, int G, int lane) {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cell = 16 * rb + 4 * (lane >> 4) + r;
            const float v = acc[rb][r];
            if (cell < TL * TL && ch_ok && v != 0.f) {
                const int X = ox + cell % TL, Y = oy + cell / TL;
                if (X < G && Y < G) atomicAdd(g + ((int64_t)Y * G + X) * nch + ch, v);
            }
        }
    }
}

// ---- second-generation walk: footprints computed ONCE per sample ------------------------------------------------
// In the walk above the 16 channel lanes of a sample all recompute its footprint (~330 VALU instructions per group of 4
// samples against 19 MFMA: the kernel was VALU-bound, PMC in profiles/README.md).  Here a wave first takes 64 samples,
// ONE PER LANE (coalesced record loads), computes each footprint once -- bilinear weights, tile-local cell of the
// top-left tap, clamped table offsets, line taps, adjoints -- and parks 16 floats per sample in LDS; the 16 groups of 4
// samples then fetch their sample's record with four broadcast ds_read_b128 and only build the MFMA operands.
struct FpRec {          // 4 x float4 per sample in LDS
    float4 w;           // bilinear weights nw, ne, sw, se (0 for taps outside the grid)
    float4 q;           // bits: c00 (tile cell of the nw tap) | g00 (clamped nw texel) | flags (1: ne step, 2: sw step, 4: valid) | lcell
    float4 l;           // l0, l1 | bits: z0c, z1c (clamped line taps)
    float4 a;           // density: dsf, dga, dgb, dgw (0 when invalid)   appearance: bits m (sample id), -, -, -
};

__device__ __forceinline__ void fp_prologue(const nmf_vm_params& p, const float4& x, bool valid, int i, int ox, int oy,
                                            int oz, float4& W, float4& Q, float4& L) {
    float xn[3];
    normalized(p, x, xn);
    const int G = p.grid;
    const float u = i == 2 ? xn[1] : xn[0], v = i == 0 ? xn[1] : xn[2];
    const float w = i == 0 ? xn[2] : (i == 1 ? xn[1] : xn[0]);
    const float gm = (float)(G - 1);
    const float ix = ((u + 1.f) * 0.5f) * gm, iy = ((v + 1.f) * 0.5f) * gm, iz = ((w + 1.f) * 0.5f) * gm;
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const float wx = ix - fx, ex = 1.f - wx, wy = iy - fy, ey = 1.f - wy, wz = iz - fz, ez = 1.f - wz;
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const bool xi0 = x0 >= 0 && x0 < G, xi1 = x0 + 1 >= 0 && x0 + 1 < G;
    const bool yi0 = y0 >= 0 && y0 < G, yi1 = y0 + 1 >= 0 && y0 + 1 < G;
    W = make_float4((xi0 && yi0) ? ex * ey : 0.f, (xi1 && yi0) ? wx * ey : 0.f, (xi0 && yi1) ? ex * wy : 0.f,
                    (xi1 && yi1) ? wx * wy : 0.f);
    const int xc0 = min(max(x0, 0), G - 1), xc1 = min(max(x0 + 1, 0), G - 1);
    const int yc0 = min(max(y0, 0), G - 1), yc1 = min(max(y0 + 1, 0), G - 1);
    const int c00 = (y0 - oy) * TL + (x0 - ox);
    const int flags = (xc1 != xc0 ? 1 : 0) | (yc1 != yc0 ? 2 : 0) | (valid ? 4 : 0);
    const bool zi0 = z0 >= 0 && z0 < G, zi1 = z0 + 1 >= 0 && z0 + 1 < G;
    Q = make_float4(__int_as_float(c00), __int_as_float(yc0 * G + xc0), __int_as_float(flags), __int_as_float(z0 - oz));
    L = make_float4(zi0 ? ez : 0.f, zi1 ? wz : 0.f, __int_as_float(min(max(z0, 0), G - 1)),
                    __int_as_float(min(max(z0 + 1, 0), G - 1)));
}

// A operand of row block rb for lane j: bilinear weight of the sample on tile cell 16*rb + j
__device__ __forceinline__ float a_weight(const float4& W, int jc /* = j - c00 */, int rb) {
    const int d = jc + 16 * rb;
    float a = d == 0 ? W.x : 0.f;
    a = d == 1 ? W.y : a;
    a = d == TL ? W.z : a;
    a = d == TL + 1 ? W.w : a;
    return a;
}

// one pipeline stage of the density walk: a sample's record (broadcast LDS reads) and its 12 + 4 table taps
template <bool WITH_NORMAL>
struct DGrp {
    float4 W, A;
    int c00, lcell;
    float l0, l1;
    float t[12];   // P, X, Y at the nw, ne, sw, se texels
    float u[4];    // L (2 taps), DL (2 taps)
    // T / TLn: the plane / line table of this walk (uniform), j4 = 4 * channel lane.  Taps are addressed as a uniform base +
    // a 32-bit byte offset (global_load ... s[base] with a VGPR offset): 64-bit per-lane pointers cost ~17 VALU per group
    // in a loop whose ceiling is MFMA + VALU cycles
    __device__ __forceinline__ void load(const float4* lds, int k, int g, const float* __restrict__ T,
                                         const float* __restrict__ TLn, int G, uint32_t j4) {
        const float4* r = lds + (4 * g + k) * 4;
        W = r[0];
        const float4 Q = r[1], L = r[2];
        A = r[3];
        c00 = __float_as_int(Q.x);
        lcell = __float_as_int(Q.w);
        l0 = L.x; l1 = L.y;
        const uint32_t fl = (uint32_t)__float_as_int(Q.z);
        const uint32_t sxo = (fl & 1u) ? (uint32_t)(DP * 4) : 0u, syo = (fl & 2u) ? (uint32_t)G * (DP * 4) : 0u;
        const char* tb = reinterpret_cast<const char*>(T);
        const uint32_t o0 = __umul24((uint32_t)__float_as_int(Q.y), (uint32_t)(DP * 4)) + j4;     // texel index < 2^24
        const uint32_t o1 = o0 + sxo, o2 = o0 + syo, o3 = o2 + sxo;
        // the channel-block offsets are added AFTER the zero extension so that they fold into the instruction's immediate
        auto ld = [](const char* b, uint32_t off, int imm) { return *reinterpret_cast<const float*>(b + (size_t)off + imm); };
        t[0] = ld(tb, o0, 0); t[1] = ld(tb, o1, 0); t[2] = ld(tb, o2, 0); t[3] = ld(tb, o3, 0);
        if (WITH_NORMAL) {
            t[4] = ld(tb, o0, CD * 4); t[5] = ld(tb, o1, CD * 4); t[6] = ld(tb, o2, CD * 4); t[7] = ld(tb, o3, CD * 4);
            t[8] = ld(tb, o0, 2 * CD * 4); t[9] = ld(tb, o1, 2 * CD * 4); t[10] = ld(tb, o2, 2 * CD * 4);
            t[11] = ld(tb, o3, 2 * CD * 4);
        }
        const char* lb = reinterpret_cast<const char*>(TLn);
        const uint32_t a0 = __umul24((uint32_t)__float_as_int(L.z), (uint32_t)(DL * 4)) + j4;
        const uint32_t a1 = __umul24((uint32_t)__float_as_int(L.w), (uint32_t)(DL * 4)) + j4;
        u[0] = ld(lb, a0, 0); u[1] = ld(lb, a1, 0);
        // the derivative line taps only meet dgw, which the value-only walk does not have (k_brick_records writes 0 for it)
        if (WITH_NORMAL) { u[2] = ld(lb, a0, CD * 4); u[3] = ld(lb, a1, CD * 4); }
    }
};

// Tried and dropped (R3, commit d86b52d): the A operands (a_weight: four compares + four selects per row block and group
// step, a third of the loop's VALU instructions -- 110 -> 74 per two group steps of the value-only walk) laid out once per
// sample as rows in LDS and read with one ds_read per row block.  Same bits, but 12 KB of LDS per wave instead of 4: next
// to the BRDF-MLP backward (150 KB of a CU's 160 KB) the walk's waves no longer fit on those CUs, and the step got SLOWER
// (in-process A/B: 1.577 -> 1.611 ms at 128^3, 1.654 -> 1.718 ms at 300^3).  The loop is bound by the latency of its table
// taps and by where its waves can be resident, not by its VALU count.  More resident waves do nothing either: the value-only
// walk forced to 80 / 64 registers (6 / 8 waves per SIMD instead of 5) measures 1.562 / 1.566 / 1.572 ms and 1.638 / 1.627 /
// 1.650 ms (300^3) -- inside the noise; the walk is not what bounds the window of the backward it runs in.
template <bool WITH_NORMAL>
__device__ __forceinline__ void vm_bwd_density2(nmf_vm_params p, const float4* __restrict__ rec0,
                                                const float4* __restrict__ rec1, int brick, int s, int e, int i, int nbx,
                                                Ptrs3 dpk, Ptrs3 dlk, MPtrs3 g_dpk, MPtrs3 g_dlk, float4* __restrict__ lds) {
    const int G = p.grid;
    const int bx = (brick % nbx) * BR, by = ((brick / nbx) % nbx) * BR, bz = (brick / (nbx * nbx)) * BR;
    const int lane = threadIdx.x & 63;
    const int k = lane >> 4, j = lane & 15;
    const int ox = i == 2 ? by : bx, oy = i == 0 ? by : bz, oz = i == 0 ? bz : (i == 1 ? by : bx);
    const float* __restrict__ T = pick3(dpk, i);
    const float* __restrict__ TLn = pick3(dlk, i);
    const uint32_t j4 = 4u * (uint32_t)j;
    floatx4 accP[NRB], accX[NRB], accY[NRB];
    floatx4 accL = {0, 0, 0, 0}, accDL = {0, 0, 0, 0};
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) { accP[rb] = accX[rb] = accY[rb] = floatx4{0, 0, 0, 0}; }

    for (int base = s; base < e; base += 64) {
        {   // one sample per lane
            const int idx = base + lane;
            const bool valid = idx < e;
            const float4 x = rec0[min(idx, e - 1)];
            float4 adj = rec1[min(idx, e - 1)];
            float4 W, Q, L;
            fp_prologue(p, x, valid, i, ox, oy, oz, W, Q, L);
            const float dga = i == 2 ? adj.z : adj.y, dgb = i == 0 ? adj.z : adj.w;
            const float dgw = i == 0 ? adj.w : (i == 1 ? adj.z : adj.y);
            const float vz = valid ? 1.f : 0.f;
            lds[lane * 4 + 0] = W;
            lds[lane * 4 + 1] = Q;
            lds[lane * 4 + 2] = L;
            lds[lane * 4 + 3] = make_float4(vz * adj.x, vz * dga, vz * dgb, vz * dgw);
        }
        __syncthreads();
        const int ng = min(16, (e - base + 3) >> 2);
        // software pipeline (ping-pong stages, scheduling barriers keep the order): the record + table taps of group g+1
        // are in flight while group g feeds the matrix pipe
        DGrp<WITH_NORMAL> st0, st1;
        auto step = [&](DGrp<WITH_NORMAL>& cur, DGrp<WITH_NORMAL>& nxt, int g) {
            nxt.load(lds, k, min(g + 1, 15), T, TLn, G, j4);  // branch-free: slots past the end hold zero adjoints
            __builtin_amdgcn_sched_barrier(0);
            const float4 W = cur.W, A = cur.A;
            const float Pq = W.x * cur.t[0] + W.y * cur.t[1] + W.z * cur.t[2] + W.w * cur.t[3];
            float Xq = 0.f, Yq = 0.f;
            if (WITH_NORMAL) {
                Xq = W.x * cur.t[4] + W.y * cur.t[5] + W.z * cur.t[6] + W.w * cur.t[7];
                Yq = W.x * cur.t[8] + W.y * cur.t[9] + W.z * cur.t[10] + W.w * cur.t[11];
            }
            const float Lc = cur.l0 * cur.u[0] + cur.l1 * cur.u[1];
            const float DLc = WITH_NORMAL ? cur.l0 * cur.u[2] + cur.l1 * cur.u[3] : 0.f;
            const float dsf = A.x, dga = A.y, dgb = A.z, dgw = A.w;
            const float bP = WITH_NORMAL ? dsf * Lc + dgw * DLc : dsf * Lc, bX = dga * Lc, bY = dgb * Lc;
            const float bL = dsf * Pq + dga * Xq + dgb * Yq, bDL = dgw * Pq;
            const int jc = j - cur.c00;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const float a = a_weight(W, jc, rb);
                accP[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bP, accP[rb], 0, 0, 0);
                if (WITH_NORMAL) {
                    accX[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bX, accX[rb], 0, 0, 0);
                    accY[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bY, accY[rb], 0, 0, 0);
                }
            }
            const int dl = j - cur.lcell;
            const float al = dl == 0 ? cur.l0 : (dl == 1 ? cur.l1 : 0.f);
            accL = __builtin_amdgcn_mfma_f32_16x16x4f32(al, bL, accL, 0, 0, 0);
            if (WITH_NORMAL) accDL = __builtin_amdgcn_mfma_f32_16x16x4f32(al, bDL, accDL, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        st0.load(lds, k, 0, T, TLn, G, j4);
        for (int g = 0; g < ng; g += 2) {       // an odd tail stage runs on a slot whose adjoints are zero
            step(st0, st1, g);
            step(st1, st0, g + 1);
        }
        __syncthreads();
    }
    flush_plane_tile(accP, pick3(g_dpk, i), DP, j, true, ox, oy, G, lane);
    if (WITH_NORMAL) {
        flush_plane_tile(accX, pick3(g_dpk, i), DP, CD + j, true, ox, oy, G, lane);
        flush_plane_tile(accY, pick3(g_dpk, i), DP, 2 * CD + j, true, ox, oy, G, lane);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cell = 4 * (lane >> 4) + r;
        if (cell < TL && oz + cell < G) {
            if (accL[r] != 0.f) atomicAdd(pick3(g_dlk, i) + (int64_t)(oz + cell) * DL + j, accL[r]);
            if (WITH_NORMAL && accDL[r] != 0.f) atomicAdd(pick3(g_dlk, i) + (int64_t)(oz + cell) * DL + CD + j, accDL[r]);
        }
    }
}

// one pipeline stage of the appearance walk
struct AGrp {
    float4 W;
    int c00, lcell;
    float l0, l1, dc0, dc1, aq0, aq1;
    float t[8];    // plane taps nw, ne, sw, se for channel j, then for channel jh
    float u[4];    // line taps (2) for channel j, then jh
    __device__ __forceinline__ void load(const float4* lds, const float* ldc, const float* lda, int k, int j, int jh,
                                         int g, const float* __restrict__ T, const float* __restrict__ TLn, int G,
                                         bool want_basis) {
        const int sl = 4 * g + k;                       // sample slot of this lane's group member
        const float4* r = lds + sl * 4;
        W = r[0];
        const float4 Q = r[1], L = r[2];
        c00 = __float_as_int(Q.x);
        lcell = __float_as_int(Q.w);
        l0 = L.x; l1 = L.y;
        dc0 = ldc[sl * CA + j]; dc1 = ldc[sl * CA + jh];            // zero for slots past the end
        aq0 = 0.f; aq1 = 0.f;
        if (want_basis) { aq0 = lda[sl * AD + j]; aq1 = lda[sl * AD + jh]; }
        // uniform base + 32-bit byte offsets, as in DGrp::load
        const uint32_t fl = (uint32_t)__float_as_int(Q.z);
        const uint32_t sxo = (fl & 1u) ? (uint32_t)(CA * 4) : 0u, syo = (fl & 2u) ? (uint32_t)G * (CA * 4) : 0u;
        const char* tb = reinterpret_cast<const char*>(T);
        const uint32_t o = __umul24((uint32_t)__float_as_int(Q.y), (uint32_t)(CA * 4));
        const uint32_t oj = o + 4u * (uint32_t)j, oh = o + 4u * (uint32_t)jh;
        auto ld = [](const char* b, uint32_t off) { return *reinterpret_cast<const float*>(b + (size_t)off); };
        t[0] = ld(tb, oj); t[1] = ld(tb, oj + sxo); t[2] = ld(tb, oj + syo); t[3] = ld(tb, oj + syo + sxo);
        t[4] = ld(tb, oh); t[5] = ld(tb, oh + sxo); t[6] = ld(tb, oh + syo); t[7] = ld(tb, oh + syo + sxo);
        const char* lb = reinterpret_cast<const char*>(TLn);
        const uint32_t a0 = __umul24((uint32_t)__float_as_int(L.z), (uint32_t)(CA * 4));
        const uint32_t a1 = __umul24((uint32_t)__float_as_int(L.w), (uint32_t)(CA * 4));
        u[0] = ld(lb, a0 + 4u * (uint32_t)j); u[1] = ld(lb, a1 + 4u * (uint32_t)j);
        u[2] = ld(lb, a0 + 4u * (uint32_t)jh); u[3] = ld(lb, a1 + 4u * (uint32_t)jh);
    }
};

__device__ __forceinline__ void vm_bwd_app2(nmf_vm_params p, const float4* __restrict__ rec0,
                                            int brick, int s, int e, int i, int nbx,
                                            Ptrs3 apl, Ptrs3 ali, const float* __restrict__ dcoef,
                                            const float* __restrict__ d_app, MPtrs3 g_apl, MPtrs3 g_ali,
                                            float* __restrict__ g_basis, float4* __restrict__ lds) {
    const int G = p.grid;
    const int bx = (brick % nbx) * BR, by = ((brick / nbx) % nbx) * BR, bz = (brick / (nbx * nbx)) * BR;
    const int lane = threadIdx.x & 63;
    const int k = lane >> 4, j = lane & 15;
    const int ox = i == 2 ? by : bx, oy = i == 0 ? by : bz, oz = i == 0 ? bz : (i == 1 ? by : bx);
    const bool hi_ok = j < CA - 16;                   // channel halves: j (0..15) and 16 + j (valid for j < 8)
    const int jh = hi_ok ? 16 + j : j;
    const float* __restrict__ T = pick3(apl, i);
    const float* __restrict__ TLn = pick3(ali, i);
    // LDS: [0,256) footprint records, [256,640) coefficient adjoints (24 per sample), [640,1024) d_app rows (24 per sample)
    float4* lds_dc = lds + 256;
    float4* lds_da = lds + 640;
    const float* ldc = reinterpret_cast<const float*>(lds_dc);
    const float* lda = reinterpret_cast<const float*>(lds_da);
    floatx4 acc0[NRB], acc1[NRB];
    floatx4 accL0 = {0, 0, 0, 0}, accL1 = {0, 0, 0, 0};
    floatx4 accW00 = {0, 0, 0, 0}, accW01 = {0, 0, 0, 0}, accW10 = {0, 0, 0, 0}, accW11 = {0, 0, 0, 0};
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) { acc0[rb] = acc1[rb] = floatx4{0, 0, 0, 0}; }

    for (int base = s; base < e; base += 64) {
        {   // one sample per lane: footprint + its streaming inputs (coefficient adjoints, d_app row) fetched with six
            // 16-byte loads each, all in flight together -- the inner loop then only touches LDS and the cached tables
            const int idx = min(base + lane, e - 1);
            const bool valid = base + lane < e;
            const float4 x = rec0[idx];
            const float4* dc = reinterpret_cast<const float4*>(dcoef + (int64_t)idx * (3 * CA) + i * CA);
            const float4* da = reinterpret_cast<const float4*>(d_app + (int64_t)idx * AD);      // rows in brick order
            float4 c[6], a[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) c[q] = valid ? dc[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (g_basis) {
#pragma unroll
                for (int q = 0; q < 6; ++q) a[q] = valid ? da[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float4 W, Q, L;
            fp_prologue(p, x, valid, i, ox, oy, oz, W, Q, L);
            lds[lane * 4 + 0] = W;
            lds[lane * 4 + 1] = Q;
            lds[lane * 4 + 2] = L;
#pragma unroll
            for (int q = 0; q < 6; ++q) lds_dc[lane * 6 + q] = c[q];
            if (g_basis) {
#pragma unroll
                for (int q = 0; q < 6; ++q) lds_da[lane * 6 + q] = a[q];
            }
        }
        __syncthreads();
        const int ng = min(16, (e - base + 3) >> 2);
        AGrp st0, st1;
        auto step = [&](AGrp& cur, AGrp& nxt, int g) {
            nxt.load(lds, ldc, lda, k, j, jh, min(g + 1, 15), T, TLn, G, g_basis != nullptr);
            __builtin_amdgcn_sched_barrier(0);
            const float4 W = cur.W;
            const float Pa0 = W.x * cur.t[0] + W.y * cur.t[1] + W.z * cur.t[2] + W.w * cur.t[3];
            const float Pa1 = W.x * cur.t[4] + W.y * cur.t[5] + W.z * cur.t[6] + W.w * cur.t[7];
            const float La0 = cur.l0 * cur.u[0] + cur.l1 * cur.u[1], La1 = cur.l0 * cur.u[2] + cur.l1 * cur.u[3];
            const float bP0 = cur.dc0 * La0, bP1 = hi_ok ? cur.dc1 * La1 : 0.f;     // adjoint of the plane entries
            const float bL0 = cur.dc0 * Pa0, bL1 = hi_ok ? cur.dc1 * Pa1 : 0.f;     // adjoint of the line entries
            const int jc = j - cur.c00;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const float a = a_weight(W, jc, rb);
                acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bP0, acc0[rb], 0, 0, 0);
                acc1[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bP1, acc1[rb], 0, 0, 0);
            }
            const int dl = j - cur.lcell;
            const float al = dl == 0 ? cur.l0 : (dl == 1 ? cur.l1 : 0.f);
            accL0 = __builtin_amdgcn_mfma_f32_16x16x4f32(al, bL0, accL0, 0, 0, 0);
            accL1 = __builtin_amdgcn_mfma_f32_16x16x4f32(al, bL1, accL1, 0, 0, 0);
            if (g_basis) {
                const float aq1 = hi_ok ? cur.aq1 : 0.f;
                const float c0 = Pa0 * La0, c1 = hi_ok ? Pa1 * La1 : 0.f;          // coefficient (tensoRF.py:204)
                accW00 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.aq0, c0, accW00, 0, 0, 0);
                accW01 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.aq0, c1, accW01, 0, 0, 0);
                accW10 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq1, c0, accW10, 0, 0, 0);
                accW11 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq1, c1, accW11, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        st0.load(lds, ldc, lda, k, j, jh, 0, T, TLn, G, g_basis != nullptr);
        for (int g = 0; g < ng; g += 2) {       // an odd tail stage runs on a slot whose adjoints are zero
            step(st0, st1, g);
            step(st1, st0, g + 1);
        }
        __syncthreads();
    }
    if (g_basis) {
        // every work item adds into the same 24 x 72 matrix (54 cache lines): 1 200 items x 3 planes serialised in the L2
        // atomic units for ~45 us of a 50 us walk.  The adds go to one of BASIS_COPIES scratch copies (by work item);
        // k_basis_reduce folds them into the caller's matrix.
        float* gb = g_basis + (int64_t)(blockIdx.x & (BASIS_COPIES - 1)) * (AD * 3 * CA);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 4 * (lane >> 4) + r;                 // row of the 16x16 tile, column = j
            float* w0 = gb + (int64_t)q * (3 * CA) + i * CA;
            float* w1 = gb + (int64_t)(16 + q) * (3 * CA) + i * CA;
            if (accW00[r] != 0.f) atomicAdd(w0 + j, accW00[r]);
            if (hi_ok && accW01[r] != 0.f) atomicAdd(w0 + 16 + j, accW01[r]);
            if (q < AD - 16) {
                if (accW10[r] != 0.f) atomicAdd(w1 + j, accW10[r]);
                if (hi_ok && accW11[r] != 0.f) atomicAdd(w1 + 16 + j, accW11[r]);
            }
        }
    }
    flush_plane_tile(acc0, pick3(g_apl, i), CA, j, true, ox, oy, G, lane);
    flush_plane_tile(acc1, pick3(g_apl, i), CA, 16 + j, hi_ok, ox, oy, G, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cell = 4 * (lane >> 4) + r;
        if (cell < TL && oz + cell < G) {
            if (accL0[r] != 0.f) atomicAdd(pick3(g_ali, i) + (int64_t)(oz + cell) * CA + j, accL0[r]);
            if (hi_ok && accL1[r] != 0.f) atomicAdd(pick3(g_ali, i) + (int64_t)(oz + cell) * CA + 16 + j, accL1[r]);
        }
    }
}

// one launch for both halves: blockIdx.x strides over the work items (brick, BWD_ITEM-sample slice) -- a persistent
// loop, the grid is capped at 16 k workgroups so that 300^3 (where most of the 422 k bricks hold a handful of samples)
// does not pay one workgroup launch per item --, blockIdx.y = density planes 0-2 / appearance planes 3-5, so all six
// latency-bound walks of an item overlap; single-wave workgroups
// HALVES: 0 = density planes only, 1 = appearance planes only, 2 = both (blockIdx.y / 3 picks).  The single-half
// instantiations keep the other walk's registers out of the allocation: both walks in one kernel need 172 registers = 2
// waves per SIMD; the density walk alone ran 3 (136 registers) and runs 4 since its tap addresses are 32-bit offsets (128).  The 1 M-sample
// density walk of a training step is latency bound (profiles r02_c: 1.68 waves per SIMD resident, ALU 41 % busy):
// 318 -> 290 us at 128^3, field backward 0.306 -> 0.256 ms per launch at 300^3.
#define NMF_BWD_ARGS                                                                                                   \
    nmf_vm_params p, const float4 *__restrict__ rec0, const float4 *__restrict__ rec1, const int32_t *__restrict__ bin_off, \
        const int2 *__restrict__ items, const int32_t *__restrict__ n_items, int item_size, int nbx, Ptrs3 dpk, Ptrs3 dlk,  \
        Ptrs3 apl, Ptrs3 ali, const float *__restrict__ dcoef, const float *__restrict__ d_app, MPtrs3 g_dpk, MPtrs3 g_dlk, \
        MPtrs3 g_apl, MPtrs3 g_ali, float *__restrict__ g_basis, int z_density, int z_app
template <bool WITH_NORMAL, int HALVES>
__device__ __forceinline__ void walk_items(NMF_BWD_ARGS) {
    // The item count lives on the device; the grid is a fixed number of single-wave workgroups that stride over the list
    // (a grid sized by the host-side upper bound -- non-empty bricks <= all bricks -- would be 100 k-1 M mostly idle
    // workgroups with 4^3 bricks)
    // 16 KB when the appearance halves run (records + coefficient / adjoint rows of 64 samples), 4 KB for a density-only
    // launch
    extern __shared__ float4 lds[];
    const int n = *n_items;
    const int half = (int)blockIdx.y / 3, i = (int)blockIdx.y % 3;
    for (int item = (int)blockIdx.x; item < n; item += (int)gridDim.x) {
        const int2 it = items[item];
        const int brick = it.x;
        const int s = bin_off[brick] + it.y * item_size, e = min(s + item_size, bin_off[brick + 1]);
        if (HALVES == 0 || (HALVES == 2 && half == z_density))
            vm_bwd_density2<WITH_NORMAL>(p, rec0, rec1, brick, s, e, i, nbx, dpk, dlk, g_dpk, g_dlk, lds);
        else if (HALVES == 1 || (HALVES == 2 && half == z_app))
            vm_bwd_app2(p, rec0, brick, s, e, i, nbx, apl, ali, dcoef, d_app, g_apl, g_ali, g_basis, lds);
        __syncthreads();
    }
}
#define NMF_BWD_PASS                                                                                                  \
    p, rec0, rec1, bin_off, items, n_items, item_size, nbx, dpk, dlk, apl, ali, dcoef, d_app, g_dpk, g_dlk, g_apl, g_ali, \
        g_basis, z_density, z_app
template <bool WITH_NORMAL, int HALVES>
__global__ void __launch_bounds__(BWD_THREADS) k_vm_bwd_brick(NMF_BWD_ARGS) {
    walk_items<WITH_NORMAL, HALVES>(NMF_BWD_PASS);
}
template <bool WITH_NORMAL>
__global__ void __launch_bounds__(BWD_THREADS) __attribute__((amdgpu_waves_per_eu(3))) k_vm_bwd_density(NMF_BWD_ARGS) {
    walk_items<WITH_NORMAL, 0>(NMF_BWD_PASS);
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
    Pack6 a;
    for (int i = 0; i < 3; ++i) {
        a.src[i] = planes[i]; a.dst[i] = dpk[i];
        a.src[3 + i] = lines[i]; a.dst[3 + i] = dlk[i];
    }
    NMF_LAUNCH(k_pack_tables, dim3((unsigned)cdiv(n, 256), 6), dim3(256), 0, (hipStream_t)stream, *p, a);
    NMF_CHECK_LAUNCH("nmf_vm_pack_density");
    return NMF_OK;
}

static int unpack_impl(const nmf_vm_params* p, const float* const g_dpk[3], const float* const g_dlk[3], float* const g_planes[3],
                       float* const g_lines[3], const float* const x_planes[3], const float* const x_lines[3], const float* l1,
                       void* stream);
extern "C" int nmf_vm_unpack_density_grad(const nmf_vm_params* p, const float* const g_dpk[3],
                                          const float* const g_dlk[3], float* const g_planes[3],
                                          float* const g_lines[3], void* stream) {
    return unpack_impl(p, g_dpk, g_dlk, g_planes, g_lines, nullptr, nullptr, nullptr, stream);
}
extern "C" int nmf_vm_unpack_density_grad_l1(const nmf_vm_params* p, const float* const g_dpk[3], const float* const g_dlk[3],
                                             float* const g_planes[3], float* const g_lines[3], const float* const x_planes[3],
                                             const float* const x_lines[3], const float* l1_scale_dev, void* stream) {
    NMF_REQUIRE(all3(x_planes) && all3(x_lines) && l1_scale_dev, NMF_EINVAL, "nmf_vm_unpack_density_grad_l1: null");
    return unpack_impl(p, g_dpk, g_dlk, g_planes, g_lines, x_planes, x_lines, l1_scale_dev, stream);
}
static int unpack_impl(const nmf_vm_params* p, const float* const g_dpk[3], const float* const g_dlk[3], float* const g_planes[3],
                       float* const g_lines[3], const float* const x_planes[3], const float* const x_lines[3], const float* l1,
                       void* stream) {
    NMF_REQUIRE(p && all3(g_dpk) && all3(g_dlk) && all3m(g_planes) && all3m(g_lines), NMF_EINVAL,
                "nmf_vm_unpack_density_grad: null");
    const int G = p->grid;
    const int64_t n = (int64_t)G * G * CD;
    Pack6 a;
    for (int i = 0; i < 3; ++i) {
        a.src[i] = g_dpk[i]; a.dst[i] = g_planes[i];
        a.src[3 + i] = g_dlk[i]; a.dst[3 + i] = g_lines[i];
    }
    L1Six q;
    for (int i = 0; i < 3; ++i) { q.x[i] = x_planes ? x_planes[i] : nullptr; q.x[3 + i] = x_lines ? x_lines[i] : nullptr; }
    q.l1 = l1;
    NMF_LAUNCH(k_unpack_tables, dim3((unsigned)cdiv(n, 256), 6), dim3(256), 0, (hipStream_t)stream, *p, a, q);
    NMF_CHECK_LAUNCH("nmf_vm_unpack_density_grad");
    return NMF_OK;
}

template <class TT>
static PtrsT3<TT> mkT(const TT* const a[3], bool on) {
    PtrsT3<TT> r;
    for (int i = 0; i < 3; ++i) r.p[i] = (on && a) ? a[i] : nullptr;
    return r;
}

template <class TT>
static int vm_query_fwd_impl(const char* what, const nmf_vm_params* p, const float* xyzt, int64_t M, const TT* const dpk[3],
                             const TT* const dlk[3], const TT* const app_planes[3], const TT* const app_lines[3],
                             const float* basis, float* sigma_feat, float* sigma, float* grad, float* normal, float* app,
                             float* coef, void* stream, const int64_t* M_live = nullptr) {
    NMF_REQUIRE(p && M >= 0, NMF_EINVAL, "nmf_vm_query_fwd: params");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(xyzt, NMF_EINVAL, "nmf_vm_query_fwd: xyzt null");
    const bool want_d = sigma_feat || sigma || grad || normal;
    const bool want_a = app || coef;
    NMF_REQUIRE(!want_d || (dpk && dlk && dpk[0] && dpk[1] && dpk[2] && dlk[0] && dlk[1] && dlk[2]), NMF_EINVAL,
                "nmf_vm_query_fwd: density tables missing");
    NMF_REQUIRE(!want_a || (app_planes && app_lines && app_planes[0] && app_planes[1] && app_planes[2] && app_lines[0] &&
                            app_lines[1] && app_lines[2] && (!app || basis)),
                NMF_EINVAL, "nmf_vm_query_fwd: appearance tables missing");
    NMF_REQUIRE(!M_live || (want_d && (grad || normal)), NMF_EINVAL,
                "nmf_vm_query_fwd_live: only the general query (density with gradient / normal) takes a device-side count");
    if (want_d && !want_a && !grad && !normal) {      // density value only
        NMF_LAUNCH_NAMED(sizeof(TT) == 4 ? "k_vm_sigma<float>" : "k_vm_sigma<unsigned short>", k_vm_sigma<TT>, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, *p,
                           (const float4*)xyzt, M, mkT<TT>(dpk, true), mkT<TT>(dlk, true), DP, DL, sigma_feat, sigma);
        NMF_CHECK_LAUNCH(what);
        return NMF_OK;
    }
    if (!want_d && app && !coef) {      // appearance of the bounce rows: 8 lanes per row
        NMF_LAUNCH_NAMED(sizeof(TT) == 4 ? "k_vm_app_rows<float>" : "k_vm_app_rows<unsigned short>", k_vm_app_rows<TT>, dim3((unsigned)cdiv(M, APP_ROWS)), dim3(256), 0, (hipStream_t)stream, *p,
                           (const float4*)xyzt, M, mkT<TT>(app_planes, true), mkT<TT>(app_lines, true), basis, app);
        NMF_CHECK_LAUNCH(what);
        return NMF_OK;
    }
    NMF_LAUNCH_NAMED(sizeof(TT) == 4 ? "k_vm_fwd<float>" : "k_vm_fwd<unsigned short>", k_vm_fwd<TT>, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, *p,
                       (const float4*)xyzt, M, mkT<TT>(dpk, want_d), mkT<TT>(dlk, want_d), mkT<TT>(app_planes, want_a),
                       mkT<TT>(app_lines, want_a), basis, sigma_feat, sigma, grad, normal, app, coef, M_live);
    NMF_CHECK_LAUNCH(what);
    return NMF_OK;
}

extern "C" int nmf_vm_query_fwd(const nmf_vm_params* p, const float* xyzt, int64_t M, const float* const dpk[3],
                                const float* const dlk[3], const float* const app_planes[3],
                                const float* const app_lines[3], const float* basis, float* sigma_feat, float* sigma,
                                float* grad, float* normal, float* app, float* coef, void* stream) {
    return vm_query_fwd_impl<float>("nmf_vm_query_fwd", p, xyzt, M, dpk, dlk, app_planes, app_lines, basis, sigma_feat, sigma,
                                    grad, normal, app, coef, stream);
}

extern "C" int nmf_vm_query_fwd_live(const nmf_vm_params* p, const float* xyzt, int64_t M_cap, const int64_t* M_live,
                                     const void* const dpk[3], const void* const dlk[3], const void* const app_planes[3],
                                     const void* const app_lines[3], int32_t tables_bf16, const float* basis, float* sigma_feat,
                                     float* sigma, float* grad, float* normal, float* app, float* coef, void* stream) {
    NMF_REQUIRE(M_live, NMF_EINVAL, "nmf_vm_query_fwd_live: M_live null");
    if (tables_bf16)
        return vm_query_fwd_impl<uint16_t>("nmf_vm_query_fwd_live", p, xyzt, M_cap, (const uint16_t* const*)dpk, (const uint16_t* const*)dlk,
                                           (const uint16_t* const*)app_planes, (const uint16_t* const*)app_lines, basis, sigma_feat,
                                           sigma, grad, normal, app, coef, stream, M_live);
    return vm_query_fwd_impl<float>("nmf_vm_query_fwd_live", p, xyzt, M_cap, (const float* const*)dpk, (const float* const*)dlk,
                                    (const float* const*)app_planes, (const float* const*)app_lines, basis, sigma_feat, sigma, grad,
                                    normal, app, coef, stream, M_live);
}

extern "C" int nmf_vm_query_sigma(const nmf_vm_params* p, const float* xyzt, int64_t M, const void* const planes[3],
                                  const void* const lines[3], int32_t tables_bf16, float* sigma_feat, float* sigma,
                                  void* stream) {
    NMF_REQUIRE(p && M >= 0, NMF_EINVAL, "nmf_vm_query_sigma: params");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(xyzt && planes && lines && planes[0] && planes[1] && planes[2] && lines[0] && lines[1] && lines[2] &&
                    (sigma_feat || sigma),
                NMF_EINVAL, "nmf_vm_query_sigma: null");
    if (tables_bf16) {
        const uint16_t* pl[3] = {(const uint16_t*)planes[0], (const uint16_t*)planes[1], (const uint16_t*)planes[2]};
        const uint16_t* li[3] = {(const uint16_t*)lines[0], (const uint16_t*)lines[1], (const uint16_t*)lines[2]};
        NMF_LAUNCH(k_vm_sigma<uint16_t>, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, *p,
                           (const float4*)xyzt, M, mkT<uint16_t>(pl, true), mkT<uint16_t>(li, true), CD, CD, sigma_feat, sigma);
    } else {
        const float* pl[3] = {(const float*)planes[0], (const float*)planes[1], (const float*)planes[2]};
        const float* li[3] = {(const float*)lines[0], (const float*)lines[1], (const float*)lines[2]};
        NMF_LAUNCH(k_vm_sigma<float>, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, *p,
                           (const float4*)xyzt, M, mkT<float>(pl, true), mkT<float>(li, true), CD, CD, sigma_feat, sigma);
    }
    NMF_CHECK_LAUNCH("nmf_vm_query_sigma");
    return NMF_OK;
}

extern "C" int nmf_vm_query_rows(const nmf_vm_params* p, const float* xyzt, int64_t M, const void* const dpk[3],
                                 const void* const dlk[3], int32_t tables_bf16, float* sigma_feat, float* sigma, float* grad,
                                 float* normal, void* stream) {
    NMF_REQUIRE(p && M >= 0, NMF_EINVAL, "nmf_vm_query_rows: params");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(xyzt && dpk && dlk && dpk[0] && dpk[1] && dpk[2] && dlk[0] && dlk[1] && dlk[2], NMF_EINVAL,
                "nmf_vm_query_rows: null");
    const dim3 grid((unsigned)cdiv(M * 16, 256)), block(256);
    if (tables_bf16) {
        PtrsT3<uint16_t> a, b;
        for (int i = 0; i < 3; ++i) { a.p[i] = (const uint16_t*)dpk[i]; b.p[i] = (const uint16_t*)dlk[i]; }
        NMF_LAUNCH(k_vm_rows_dn<uint16_t>, grid, block, 0, (hipStream_t)stream, *p, (const float4*)xyzt, M, a, b,
                           sigma_feat, sigma, grad, normal);
    } else {
        PtrsT3<float> a, b;
        for (int i = 0; i < 3; ++i) { a.p[i] = (const float*)dpk[i]; b.p[i] = (const float*)dlk[i]; }
        NMF_LAUNCH(k_vm_rows_dn<float>, grid, block, 0, (hipStream_t)stream, *p, (const float4*)xyzt, M, a, b,
                           sigma_feat, sigma, grad, normal);
    }
    NMF_CHECK_LAUNCH("nmf_vm_query_rows");
    return NMF_OK;
}

extern "C" int nmf_vm_query_fwd_bf16(const nmf_vm_params* p, const float* xyzt, int64_t M, const uint16_t* const dpk[3],
                                     const uint16_t* const dlk[3], const uint16_t* const app_planes[3],
                                     const uint16_t* const app_lines[3], const float* basis, float* sigma_feat,
                                     float* sigma, float* grad, float* normal, float* app, float* coef, void* stream) {
    return vm_query_fwd_impl<uint16_t>("nmf_vm_query_fwd_bf16", p, xyzt, M, dpk, dlk, app_planes, app_lines, basis,
                                       sigma_feat, sigma, grad, normal, app, coef, stream);
}

// counter copies per brick (power of two; measured on S1: 1 -> 64 us, 4 -> 45 us, 8 -> 64 us of binning per 0.88 M
// samples, the scan growing with the copies): 4 while the single-workgroup scan over [brick][copy] stays short
static int bin_copies(int64_t nb) {
    return nb <= 32768 ? 4 : 1;
}

// ---- layout of a plan (nmf_vm_bin_plan) and of the scratch of the walk itself ------------------------------------------
namespace {
struct PlanLayout {
    int nbx, nb, kc, item_size, n_chunks;
    int64_t max_items;
    int2* keyrank;        // [M]   (brick * kc + counter copy, rank inside it)
    int32_t* slot;        // [M]   position of sample m in brick order
    int32_t* counts;      // [(nb+1)*kc]
    unsigned long long* scan_state;   // [1 + n_scan_chunks]: ticket, chunk words of k_bins_scan (behind the counters: one memset)
    int n_scan_chunks;
    size_t zero_bytes;    // counts + scan_state
    int32_t* offsets;     // [nb+1]
    int32_t* cursor;      // [(nb+1)*kc]  start of every (brick, copy)
    int32_t* n_items;     // [2]
    int64_t* chunk_tot;   // [n_chunks]
    int2* items;          // [max_items]
    float4* rec0;         // [M]   positions in brick order
    int64_t bytes;
};
inline uintptr_t up16(uintptr_t q) { return (q + 15) & ~(uintptr_t)15; }
PlanLayout plan_layout(void* base, int64_t M, int32_t grid) {
    PlanLayout L;
    L.nbx = (grid + BR - 1) / BR;
    L.nb = L.nbx * L.nbx * L.nbx;
    L.kc = bin_copies(L.nb);
    int item_size = M > 400000 ? BWD_ITEM : BWD_ITEM / 2;   // measured: profiles/README.md (r01_i)
    L.item_size = (item_size + 3) & ~3;
    L.max_items = M / BWD_ITEM_MIN + L.nb + 1;
    L.n_chunks = (L.nb + SB_CHUNK - 1) / SB_CHUNK;
    uintptr_t q = up16((uintptr_t)base);
    L.keyrank = (int2*)q;                q = up16(q + sizeof(int2) * M);
    L.slot = (int32_t*)q;                q = up16(q + sizeof(int32_t) * M);
    L.counts = (int32_t*)q;              q = up16(q + sizeof(int32_t) * (size_t)(L.nb + 1) * L.kc);
    L.n_scan_chunks = (L.nb + SC_CHUNK - 1) / SC_CHUNK;
    L.scan_state = (unsigned long long*)q;   q = up16(q + sizeof(unsigned long long) * (size_t)(L.n_scan_chunks + 1));
    L.zero_bytes = (size_t)(q - (uintptr_t)L.counts);
    L.offsets = (int32_t*)q;             q = up16(q + sizeof(int32_t) * (size_t)(L.nb + 1));
    L.cursor = (int32_t*)q;              q = up16(q + sizeof(int32_t) * (size_t)(L.nb + 1) * L.kc);
    L.n_items = (int32_t*)q;             q = up16(q + sizeof(int32_t) * 2);
    L.chunk_tot = (int64_t*)q;           q = up16(q + sizeof(int64_t) * (size_t)(L.n_chunks + 1));
    L.items = (int2*)q;                  q = up16(q + sizeof(int2) * (size_t)L.max_items);
    L.rec0 = (float4*)q;                 q = up16(q + sizeof(float4) * M);
    L.bytes = (int64_t)(q - (uintptr_t)base) + 16;
    return L;
}
// the scratch a caller may keep between walks (nmf_vm_query_bwd_segments_clean): zero on entry, zero again on exit
struct CleanLayout {
    int32_t* counts;
    unsigned long long* scan_state;
    float* basis_copies;
    int64_t bytes;
};
CleanLayout clean_layout(void* base, int32_t grid) {
    const PlanLayout L = plan_layout(nullptr, 0, grid);
    CleanLayout C;
    uintptr_t q = up16((uintptr_t)base);
    C.counts = (int32_t*)q;                       q = up16(q + sizeof(int32_t) * (size_t)(L.nb + 1) * L.kc);
    C.scan_state = (unsigned long long*)q;        q = up16(q + sizeof(unsigned long long) * (size_t)(L.n_scan_chunks + 1));
    C.basis_copies = (float*)q;                   q = up16(q + sizeof(float) * BASIS_COPIES * AD * 3 * CA);
    C.bytes = (int64_t)(q - (uintptr_t)base) + 16;
    return C;
}
struct WalkLayout {
    float4* rec1;          // [M]
    float* dcoef;          // [M][72]
    float* d_app_sorted;   // [M][24]
    float* basis_copies;   // [BASIS_COPIES][24][72]
    int64_t bytes;
};
WalkLayout walk_layout(void* base, int64_t M) {
    WalkLayout W;
    uintptr_t q = up16((uintptr_t)base);
    W.rec1 = (float4*)q;                 q = up16(q + sizeof(float4) * M);
    W.dcoef = (float*)q;                 q = up16(q + sizeof(float) * M * 3 * CA);
    W.d_app_sorted = (float*)q;          q = up16(q + sizeof(float) * M * AD);
    W.basis_copies = (float*)q;          q = up16(q + sizeof(float) * BASIS_COPIES * AD * 3 * CA);
    W.bytes = (int64_t)(q - (uintptr_t)base) + 16;
    return W;
}

// host array of segments -> the kernels' view; returns the total sample count (or < 0 with the error set)
int64_t gather_segments(const nmf_vm_bwd_segment* segs, int32_t n_segs, Segs& sg, int& n) {
    memset(&sg, 0, sizeof(sg));
    int64_t M = 0;
    n = 0;
    for (int i = 0; i < n_segs; ++i) {
        const nmf_vm_bwd_segment& q = segs[i];
        if (q.M < 0 || (q.M > 0 && !q.xyzt)) return -1;
        if (q.M == 0) continue;
        sg.xyzt[n] = q.xyzt; sg.sigma_feat[n] = q.sigma_feat; sg.grad[n] = q.grad; sg.d_sigma[n] = q.d_sigma;
        sg.d_sigma_feat[n] = q.d_sigma_feat; sg.d_normal[n] = q.d_normal; sg.d_app[n] = q.d_app;
        sg.start[n] = M;
        M += q.M;
        ++n;
    }
    for (int i = n; i <= MAX_SEG; ++i) sg.start[i] = M;      // unused segments start past the end
    return M;
}

// place = false: the caller follows with k_place_records (the walk that sorts inside its own call)
// clean: L.counts / L.scan_state point into the caller's kept scratch (zero now, zero again afterwards): no memset
int launch_plan(const nmf_vm_params* p, const Segs& sg, int64_t M, const PlanLayout& L, hipStream_t st, bool place, bool clean = false) {
    // the look-back needs its chunks' workgroups resident together: 2048 of them fit the chip (8 per CU); grids beyond ~500^3 take
    // the two-launch scan (k_bins_partial + k_bins_final)
    const bool lookback = L.n_scan_chunks <= 2048;
    const size_t count_bytes = sizeof(int32_t) * (size_t)(L.nb + 1) * L.kc;
    if (!clean || !lookback) {
        hipError_t e = hipMemsetAsync(L.counts, 0, clean ? count_bytes : L.zero_bytes, st);
        if (e != hipSuccess) return nmf_fail((int)e, "nmf_vm_bin_plan: memset");
    }
    NMF_LAUNCH(k_plan_hist, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, st, *p, sg, M, L.nbx, L.kc, L.counts, L.keyrank);
    if (lookback) {
        NMF_LAUNCH(k_bins_scan, dim3(L.n_scan_chunks), dim3(SC_THREADS), 0, st, L.counts, L.nb, L.kc, L.scan_state, L.offsets,
                           L.cursor, L.item_size, L.items, L.n_items, clean ? 1 : 0);
    } else {
        NMF_LAUNCH(k_bins_partial, dim3(L.n_chunks), dim3(SB_THREADS), 0, st, L.counts, L.nb, L.kc, L.item_size, L.chunk_tot);
        NMF_LAUNCH(k_bins_final, dim3(L.n_chunks), dim3(SB_THREADS), 0, st, L.counts, L.nb, L.kc, L.chunk_tot, L.offsets,
                           L.cursor, L.item_size, L.items, L.n_items);
        if (clean) {         // (grids beyond 500^3: the kept counters are handed back zero by a second memset)
            hipError_t e = hipMemsetAsync(L.counts, 0, count_bytes, st);
            if (e != hipSuccess) return nmf_fail((int)e, "nmf_vm_bin_plan: memset");
        }
    }
    if (place)
        NMF_LAUNCH(k_plan_place, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, st, sg, M, L.keyrank, L.cursor, L.slot, L.rec0);
    NMF_CHECK_LAUNCH("nmf_vm_bin_plan");
    return NMF_OK;
}
}  // namespace

extern "C" int64_t nmf_vm_bin_plan_bytes(int64_t M, int32_t grid) {
    return plan_layout(nullptr, M < 0 ? 0 : M, grid).bytes;
}

extern "C" int64_t nmf_vm_walk_workspace_bytes(int64_t M) { return walk_layout(nullptr, M < 0 ? 0 : M).bytes; }

extern "C" int64_t nmf_vm_bwd_workspace_bytes(int64_t M, int32_t grid) {
    return nmf_vm_bin_plan_bytes(M, grid) + nmf_vm_walk_workspace_bytes(M);
}

extern "C" int nmf_vm_bin_plan(const nmf_vm_params* p, const float* const* xyzt, const int64_t* Ms, int32_t n_segs,
                               void* plan, int64_t plan_bytes, void* stream) {
    NMF_REQUIRE(p && n_segs >= 0 && n_segs <= MAX_SEG && ((xyzt && Ms) || n_segs == 0), NMF_EINVAL,
                "nmf_vm_bin_plan: params / segment count (at most NMF_VM_MAX_SEGMENTS)");
    nmf_vm_bwd_segment segs[MAX_SEG];
    memset(segs, 0, sizeof(segs));
    for (int i = 0; i < n_segs; ++i) { segs[i].xyzt = xyzt[i]; segs[i].M = Ms[i]; }
    Segs sg;
    int n;
    const int64_t M = gather_segments(segs, n_segs, sg, n);
    NMF_REQUIRE(M >= 0, NMF_EINVAL, "nmf_vm_bin_plan: M < 0 or xyzt null");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(M < (1ll << 31), NMF_ERANGE, "nmf_vm_bin_plan: M >= 2^31");
    NMF_REQUIRE(plan && plan_bytes >= nmf_vm_bin_plan_bytes(M, p->grid), NMF_EINVAL,
                "nmf_vm_bin_plan: plan buffer too small (see nmf_vm_bin_plan_bytes)");
    return launch_plan(p, sg, M, plan_layout(plan, M, p->grid), (hipStream_t)stream, true);
}

static int vm_bwd_impl(const nmf_vm_params* p, const nmf_vm_bwd_segment* segs, int32_t n_segs, const float* const dpk[3],
                       const float* const dlk[3], const float* const app_planes[3], const float* const app_lines[3],
                       const float* basis, float* const g_dpk[3], float* const g_dlk[3], float* const g_app_planes[3],
                       float* const g_app_lines[3], float* g_basis, const void* plan, int64_t plan_bytes, void* workspace,
                       int64_t workspace_bytes, void* stream, void* clean = nullptr, int64_t clean_bytes = 0) {
    NMF_REQUIRE(p && n_segs >= 0 && n_segs <= MAX_SEG && (segs || n_segs == 0), NMF_EINVAL,
                "nmf_vm_query_bwd: params / segment count (at most NMF_VM_MAX_SEGMENTS)");
    NMF_REQUIRE(!clean || (!plan && clean_bytes >= clean_layout(nullptr, p->grid).bytes && ((uintptr_t)clean & 15) == 0), NMF_EINVAL,
                "nmf_vm_query_bwd_segments_clean: scratch too small (nmf_vm_bwd_clean_bytes) or not 16-byte aligned");
    Segs sg;
    int n;
    const int64_t M = gather_segments(segs, n_segs, sg, n);
    NMF_REQUIRE(M >= 0, NMF_EINVAL, "nmf_vm_query_bwd: M < 0 or xyzt null");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(M < (1ll << 31), NMF_ERANGE, "nmf_vm_query_bwd: M >= 2^31");
    for (int i = 1; i < n; ++i)                              // the kernels branch on segment 0's adjoint set
        NMF_REQUIRE(!sg.d_sigma[i] == !sg.d_sigma[0] && !sg.d_normal[i] == !sg.d_normal[0] && !sg.d_app[i] == !sg.d_app[0],
                    NMF_EINVAL, "nmf_vm_query_bwd: segments must provide the same adjoints");
    const float *d_sigma = sg.d_sigma[0], *d_normal = sg.d_normal[0], *d_app = sg.d_app[0];
    bool any_dsf = false;
    for (int i = 0; i < n; ++i) {
        any_dsf = any_dsf || sg.d_sigma_feat[i];
        NMF_REQUIRE(!d_sigma || sg.sigma_feat[i], NMF_EINVAL, "nmf_vm_query_bwd: d_sigma needs saved sigma_feat");
        NMF_REQUIRE(!d_normal || sg.grad[i], NMF_EINVAL, "nmf_vm_query_bwd: d_normal needs saved grad");
    }
    const bool want_d = d_sigma || any_dsf || d_normal;
    const bool want_a = d_app != nullptr;
    NMF_REQUIRE(!want_d || (all3(dpk) && all3(dlk) && all3m(g_dpk) && all3m(g_dlk)), NMF_EINVAL,
                "nmf_vm_query_bwd: density tables missing");
    NMF_REQUIRE(!want_a || (all3(app_planes) && all3(app_lines) && basis && all3m(g_app_planes) && all3m(g_app_lines)),
                NMF_EINVAL, "nmf_vm_query_bwd: appearance tables missing");
    if (!want_d && !want_a) return NMF_OK;
    hipStream_t st = (hipStream_t)stream;
    PlanLayout L;
    void* walk_ws = workspace;
    int64_t walk_bytes = workspace_bytes;
    if (plan) {          // the sort was done when the positions became known (nmf_vm_bin_plan over the same segment sizes)
        NMF_REQUIRE(plan_bytes >= nmf_vm_bin_plan_bytes(M, p->grid), NMF_EINVAL, "nmf_vm_query_bwd_planned: plan buffer too small");
        L = plan_layout(const_cast<void*>(plan), M, p->grid);
    } else {             // sort here: the plan lives at the front of the workspace
        NMF_REQUIRE(workspace && workspace_bytes >= nmf_vm_bwd_workspace_bytes(M, p->grid), NMF_EINVAL,
                    "nmf_vm_query_bwd: workspace too small (see nmf_vm_bwd_workspace_bytes)");
        L = plan_layout(workspace, M, p->grid);
        if (clean) {
            const CleanLayout C = clean_layout(clean, p->grid);
            L.counts = C.counts;
            L.scan_state = C.scan_state;
        }
        const int rc = launch_plan(p, sg, M, L, st, false, clean != nullptr);
        if (rc != NMF_OK) return rc;
        walk_ws = (char*)workspace + L.bytes;
        walk_bytes = workspace_bytes - L.bytes;
    }
    NMF_REQUIRE(walk_ws && walk_bytes >= nmf_vm_walk_workspace_bytes(M), NMF_EINVAL,
                "nmf_vm_query_bwd: workspace too small (see nmf_vm_walk_workspace_bytes)");
    WalkLayout W = walk_layout(walk_ws, M);
    const bool use_copies = want_a && g_basis;
    if (use_copies && clean) W.basis_copies = clean_layout(clean, p->grid).basis_copies;       // (zero: no memset)
    else if (use_copies) {
        hipError_t e = hipMemsetAsync(W.basis_copies, 0, sizeof(float) * BASIS_COPIES * AD * 3 * CA, st);
        if (e != hipSuccess) return nmf_fail((int)e, "nmf_vm_query_bwd: memset");
    }
    // the chunk words of the look-back scan are cleared by the launch behind it
    unsigned long long* st_clear = (clean && L.n_scan_chunks <= 2048) ? L.scan_state : nullptr;
    const int n_state = L.n_scan_chunks + 1;
    const dim3 per_sample((unsigned)cdiv(M, 256));
    if (want_a) {
        if (plan) NMF_LAUNCH(k_brick_records<true>, per_sample, dim3(256), 0, st, *p, sg, L.slot, M, W.rec1, W.d_app_sorted);
        else NMF_LAUNCH(k_place_records<true>, per_sample, dim3(256), 0, st, *p, sg, M, L.keyrank, L.cursor, L.rec0, W.rec1, W.d_app_sorted, st_clear, n_state);
        NMF_LAUNCH(k_dcoef, dim3((unsigned)cdiv(M * (3 * CA / 4), 256)), dim3(256), 0, st, W.d_app_sorted, basis, M,
                           W.dcoef);
    } else if (plan)
        NMF_LAUNCH(k_brick_records<false>, per_sample, dim3(256), 0, st, *p, sg, L.slot, M, W.rec1, W.d_app_sorted);
    else
        NMF_LAUNCH(k_place_records<false>, per_sample, dim3(256), 0, st, *p, sg, M, L.keyrank, L.cursor, L.rec0, W.rec1, W.d_app_sorted, st_clear, n_state);
    const int nz = (want_d ? 1 : 0) + (want_a ? 1 : 0);
    const int z_density = want_d ? 0 : -1, z_app = want_a ? (want_d ? 1 : 0) : -1;
    int64_t gcap = 16384;
    const int64_t max_items = M / L.item_size + L.nb + 1;
    const int64_t grid_x = max_items < gcap ? max_items : gcap;       // single-wave workgroups per plane
    const dim3 grid((unsigned)grid_x, (unsigned)(3 * nz)), block(BWD_THREADS);
    const size_t lds_bytes = sizeof(float4) * 64 * (want_a ? 16 : 4);
#define NMF_LAUNCH_BWD_KERNEL(KERNEL)                                                                                         \
    NMF_LAUNCH(KERNEL, grid, block, lds_bytes, st, *p, L.rec0, W.rec1, L.offsets, L.items, L.n_items, L.item_size,    \
                       L.nbx, mk(dpk), mk(dlk), mk(app_planes), mk(app_lines), W.dcoef, W.d_app_sorted, mkm(g_dpk),           \
                       mkm(g_dlk), mkm(g_app_planes), mkm(g_app_lines), use_copies ? W.basis_copies : nullptr, z_density, z_app)
    if (want_d && want_a) {
        if (d_normal) NMF_LAUNCH_BWD_KERNEL((k_vm_bwd_brick<true, 2>));
        else NMF_LAUNCH_BWD_KERNEL((k_vm_bwd_brick<false, 2>));
    } else if (want_d) {
        if (d_normal) NMF_LAUNCH_BWD_KERNEL((k_vm_bwd_density<true>));
        else NMF_LAUNCH_BWD_KERNEL((k_vm_bwd_density<false>));
    } else
        NMF_LAUNCH_BWD_KERNEL((k_vm_bwd_brick<false, 1>));
#undef NMF_LAUNCH_BWD_KERNEL
    if (use_copies)
        NMF_LAUNCH(k_basis_reduce, dim3((unsigned)cdiv(AD * 3 * CA, 256)), dim3(256), 0, st, W.basis_copies, g_basis, clean ? 1 : 0);
    NMF_CHECK_LAUNCH("nmf_vm_query_bwd");
    return NMF_OK;
}

extern "C" int64_t nmf_vm_bwd_clean_bytes(int32_t grid) { return clean_layout(nullptr, grid).bytes; }

extern "C" int nmf_vm_query_bwd_segments_clean(const nmf_vm_params* p, const nmf_vm_bwd_segment* segs, int32_t n_segs,
                                               const float* const dpk[3], const float* const dlk[3],
                                               const float* const app_planes[3], const float* const app_lines[3],
                                               const float* basis, float* const g_dpk[3], float* const g_dlk[3],
                                               float* const g_app_planes[3], float* const g_app_lines[3], float* g_basis,
                                               void* clean, int64_t clean_bytes, void* workspace, int64_t workspace_bytes,
                                               void* stream) {
    NMF_REQUIRE(clean, NMF_EINVAL, "nmf_vm_query_bwd_segments_clean: scratch null");
    return vm_bwd_impl(p, segs, n_segs, dpk, dlk, app_planes, app_lines, basis, g_dpk, g_dlk, g_app_planes, g_app_lines, g_basis,
                       nullptr, 0, workspace, workspace_bytes, stream, clean, clean_bytes);
}

extern "C" int nmf_vm_query_bwd_segments(const nmf_vm_params* p, const nmf_vm_bwd_segment* segs, int32_t n_segs,
                                         const float* const dpk[3], const float* const dlk[3],
                                         const float* const app_planes[3], const float* const app_lines[3],
                                         const float* basis, float* const g_dpk[3], float* const g_dlk[3],
                                         float* const g_app_planes[3], float* const g_app_lines[3], float* g_basis,
                                         void* workspace, int64_t workspace_bytes, void* stream) {
    return vm_bwd_impl(p, segs, n_segs, dpk, dlk, app_planes, app_lines, basis, g_dpk, g_dlk, g_app_planes, g_app_lines, g_basis,
                       nullptr, 0, workspace, workspace_bytes, stream);
}

extern "C" int nmf_vm_query_bwd_planned(const nmf_vm_params* p, const nmf_vm_bwd_segment* segs, int32_t n_segs,
                                        const float* const dpk[3], const float* const dlk[3],
                                        const float* const app_planes[3], const float* const app_lines[3],
                                        const float* basis, float* const g_dpk[3], float* const g_dlk[3],
                                        float* const g_app_planes[3], float* const g_app_lines[3], float* g_basis,
                                        const void* plan, int64_t plan_bytes, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
    NMF_REQUIRE(plan, NMF_EINVAL, "nmf_vm_query_bwd_planned: plan null");
    return vm_bwd_impl(p, segs, n_segs, dpk, dlk, app_planes, app_lines, basis, g_dpk, g_dlk, g_app_planes, g_app_lines, g_basis,
                       plan, plan_bytes, workspace, workspace_bytes, stream);
}

extern "C" int nmf_vm_query_bwd(const nmf_vm_params* p, const float* xyzt, int64_t M, const float* const dpk[3],
                                const float* const dlk[3], const float* const app_planes[3],
                                const float* const app_lines[3], const float* basis, const float* sigma_feat,
                                const float* grad, const float* d_sigma, const float* d_sigma_feat,
                                const float* d_normal, const float* d_app, float* const g_dpk[3],
                                float* const g_dlk[3], float* const g_app_planes[3], float* const g_app_lines[3],
                                float* g_basis, void* workspace, int64_t workspace_bytes, void* stream) {
    NMF_REQUIRE(p && M >= 0, NMF_EINVAL, "nmf_vm_query_bwd: params");
    if (M == 0) return NMF_OK;
    nmf_vm_bwd_segment seg;
    seg.xyzt = xyzt; seg.M = M; seg.sigma_feat = sigma_feat; seg.grad = grad; seg.d_sigma = d_sigma;
    seg.d_sigma_feat = d_sigma_feat; seg.d_normal = d_normal; seg.d_app = d_app;
    return nmf_vm_query_bwd_segments(p, &seg, 1, dpk, dlk, app_planes, app_lines, basis, g_dpk, g_dlk, g_app_planes,
                                     g_app_lines, g_basis, workspace, workspace_bytes, stream);
}
