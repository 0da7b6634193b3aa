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
// backward: one lane per sample, atomics into the packed gradient tables
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_vm_bwd(nmf_vm_params p, const float4* __restrict__ xyzt, int64_t M,
                                                Ptrs3 dpk, Ptrs3 dlk, Ptrs3 apl, Ptrs3 ali,
                                                const float* __restrict__ basis,
                                                const float* __restrict__ sigma_feat, const float* __restrict__ grad,
                                                const float* __restrict__ d_sigma,
                                                const float* __restrict__ d_sigma_feat,
                                                const float* __restrict__ d_normal, const float* __restrict__ d_app,
                                                MPtrs3 g_dpk, MPtrs3 g_dlk, MPtrs3 g_apl, MPtrs3 g_ali) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int G = p.grid;
    float xn[3];
    normalized(p, xyzt[m], xn);

    const bool has_density = dpk.p[0] && (d_sigma || d_sigma_feat || d_normal);
    if (has_density) {
        // adjoint of the raw feature
        float dsf = d_sigma_feat ? d_sigma_feat[m] : 0.f;
        if (d_sigma) {
            float f = sigma_feat[m];
            float x = fminf(fmaxf(f, -15.f), 1e3f) + p.density_shift;
            float ds = x > 20.f ? 1.f : 1.f / (1.f + expf(-x));             // softplus'
            if (f < -15.f || f > 1e3f) ds = 0.f;                            // clamp'
            dsf += d_sigma[m] * ds;
        }
        // adjoint of the raw gradient g (through n = -g / sqrt(max(|g|^2, eps)))
        float dg[3] = {0.f, 0.f, 0.f};
        if (d_normal) {
            float g0 = grad[m * 3], g1 = grad[m * 3 + 1], g2 = grad[m * 3 + 2];
            float dn0 = d_normal[m * 3], dn1 = d_normal[m * 3 + 1], dn2 = d_normal[m * 3 + 2];
            float n2 = g0 * g0 + g1 * g1 + g2 * g2;
            const float eps = 1.1920929e-07f;
            float inv = 1.f / sqrtf(fmaxf(n2, eps));
            // n = -g*inv ; d inv / d g = -g * inv^3 when n2 > eps else 0
            float dot = dn0 * g0 + dn1 * g1 + dn2 * g2;
            float k = n2 > eps ? dot * inv * inv * inv : 0.f;
            dg[0] = -dn0 * inv + k * g0;
            dg[1] = -dn1 * inv + k * g1;
            dg[2] = -dn2 * inv + k * g2;
            dg[0] *= p.inv_size[0]; dg[1] *= p.inv_size[1]; dg[2] *= p.inv_size[2];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float dga = dg[MAT0[i]], dgb = dg[MAT1[i]], dgw = dg[VEC[i]];
            const Tap1 tl = make_tap1(xn[VEC[i]], G);
            const Tap2 tp = make_tap2(xn[MAT0[i]], xn[MAT1[i]], G);
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
            // line adjoints need the interpolated plane values; accumulate them while scattering the
            // plane adjoints (which only need the line values)
            float aL[CD], aDL[CD];
#pragma unroll
            for (int c = 0; c < CD; ++c) { aL[c] = 0.f; aDL[c] = 0.f; }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (tp.idx[t] < 0) continue;
                const int64_t off = (int64_t)tp.idx[t] * DP;
                float run[DP];
                load_run<DP / 4>(dpk.p[i] + off, run);
                float* gq = g_dpk.p[i] + off;
                const float w = tp.w[t];
#pragma unroll
                for (int c = 0; c < CD; ++c) {
                    aL[c] += w * (dsf * run[c] + dga * run[CD + c] + dgb * run[2 * CD + c]);
                    aDL[c] += w * (dgw * run[c]);
                    atomicAdd(gq + c, w * (dsf * Lc[c] + dgw * DLc[c]));
                }
                if (d_normal) {
#pragma unroll
                    for (int c = 0; c < CD; ++c) {
                        atomicAdd(gq + CD + c, w * dga * Lc[c]);
                        atomicAdd(gq + 2 * CD + c, w * dgb * Lc[c]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (tl.idx[t] < 0) continue;
                float* gq = g_dlk.p[i] + (int64_t)tl.idx[t] * DL;
                const float w = tl.w[t];
#pragma unroll
                for (int c = 0; c < CD; ++c) atomicAdd(gq + c, w * aL[c]);
                if (d_normal) {
#pragma unroll
                    for (int c = 0; c < CD; ++c) atomicAdd(gq + CD + c, w * aDL[c]);
                }
            }
        }
    }

    if (apl.p[0] && d_app) {
        float da[AD];
        load_run<AD / 4>(d_app + m * AD, da);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float dcoef[CA];
#pragma unroll
            for (int c = 0; c < CA; ++c) dcoef[c] = 0.f;
#pragma unroll
            for (int j = 0; j < AD; ++j) {
                const float* wrow = basis + j * (3 * CA) + i * CA;
#pragma unroll
                for (int c = 0; c < CA; ++c) dcoef[c] += wrow[c] * da[j];
            }
            const Tap1 tl = make_tap1(xn[VEC[i]], G);
            const Tap2 tp = make_tap2(xn[MAT0[i]], xn[MAT1[i]], G);
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
            float aLa[CA];
#pragma unroll
            for (int c = 0; c < CA; ++c) aLa[c] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (tp.idx[t] < 0) continue;
                const int64_t off = (int64_t)tp.idx[t] * CA;
                float run[CA];
                load_run<CA / 4>(apl.p[i] + off, run);
                float* gq = g_apl.p[i] + off;
                const float w = tp.w[t];
#pragma unroll
                for (int c = 0; c < CA; ++c) {
                    aLa[c] += w * run[c] * dcoef[c];
                    atomicAdd(gq + c, w * dcoef[c] * La[c]);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (tl.idx[t] < 0) continue;
                float* gq = g_ali.p[i] + (int64_t)tl.idx[t] * CA;
#pragma unroll
                for (int c = 0; c < CA; ++c) atomicAdd(gq + c, tl.w[t] * aLa[c]);
            }
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

extern "C" int nmf_vm_query_bwd(const nmf_vm_params* p, const float* xyzt, int64_t M, const float* const dpk[3],
                                const float* const dlk[3], const float* const app_planes[3],
                                const float* const app_lines[3], const float* basis, const float* sigma_feat,
                                const float* grad, const float* d_sigma, const float* d_sigma_feat,
                                const float* d_normal, const float* d_app, float* const g_dpk[3],
                                float* const g_dlk[3], float* const g_app_planes[3], float* const g_app_lines[3],
                                void* stream) {
    NMF_REQUIRE(p && M >= 0, NMF_EINVAL, "nmf_vm_query_bwd: params");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(xyzt, NMF_EINVAL, "nmf_vm_query_bwd: xyzt null");
    const bool want_d = d_sigma || d_sigma_feat || d_normal;
    const bool want_a = d_app != nullptr;
    NMF_REQUIRE(!want_d || (all3(dpk) && all3(dlk) && all3m(g_dpk) && all3m(g_dlk)), NMF_EINVAL,
                "nmf_vm_query_bwd: density tables missing");
    NMF_REQUIRE(!d_sigma || sigma_feat, NMF_EINVAL, "nmf_vm_query_bwd: d_sigma needs saved sigma_feat");
    NMF_REQUIRE(!d_normal || grad, NMF_EINVAL, "nmf_vm_query_bwd: d_normal needs saved grad");
    NMF_REQUIRE(!want_a || (all3(app_planes) && all3(app_lines) && basis && all3m(g_app_planes) && all3m(g_app_lines)),
                NMF_EINVAL, "nmf_vm_query_bwd: appearance tables missing");
    hipLaunchKernelGGL(k_vm_bwd, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, *p,
                       (const float4*)xyzt, M, want_d ? mk(dpk) : mk(nullptr), want_d ? mk(dlk) : mk(nullptr),
                       want_a ? mk(app_planes) : mk(nullptr), want_a ? mk(app_lines) : mk(nullptr), basis, sigma_feat,
                       grad, d_sigma, d_sigma_feat, d_normal, d_app, want_d ? mkm(g_dpk) : mkm(nullptr),
                       want_d ? mkm(g_dlk) : mkm(nullptr), want_a ? mkm(g_app_planes) : mkm(nullptr),
                       want_a ? mkm(g_app_lines) : mkm(nullptr));
    NMF_CHECK_LAUNCH("nmf_vm_query_bwd");
    return NMF_OK;
}
