// Pieces of the per-bounce-row backward shared by csrc/shade.hip (k_bounce_prep_bwd) and csrc/heads.hip (the fused row adjoint):
// the SH irradiance of a normal and the adjoint of one row of nmf_bounce_prep_fwd (models/microfacet.py:304-316,333-361).
#pragma once
#include "common.hpp"

namespace nmf_rows {

constexpr int HEADS = 11;       // albedo 3 | tint 3 | f0 3 | roughness 2 (nmf_heads_fwd)
constexpr int FEAT = NMF_APP_DIM;

struct Conv {
    const float* c;   // [9][3] device pointer, uniform -> scalar loads
};

// the 9 real SH bases of modules/sh.py:97-142 (all-positive SH_C2 table of :67-73)
__device__ __forceinline__ void sh9(float x, float y, float z, float (&Y)[9]) {
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C20 = 1.0925484305920792f, C22 = 0.31539156525252005f, C24 = 0.5462742152960396f;
    Y[0] = C0;
    Y[1] = C1 * y;
    Y[2] = C1 * z;
    Y[3] = C1 * x;
    Y[4] = C20 * (x * y);
    Y[5] = C20 * (y * z);
    Y[6] = C22 * (3.f * (z * z) - 1.f);
    Y[7] = C20 * (x * z);
    Y[8] = C24 * (x * x - y * y);
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }


// SH irradiance factors E[c] of a normal, as ONE piece of code for both forms of the backward below (not inlined: the row-based fast
// path and the general path must give the same bits -- tests compare them -- and inlined copies are contracted into fmas differently
// depending on what surrounds them)
static __device__ __noinline__ void irradiance_E(float nx, float ny, float nz, Conv conv, float* __restrict__ E3) {
    float Y[9];
    sh9(nx, ny, nz, Y);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float E = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) E += conv.c[k * 3 + c] * Y[k];
        E3[c] = E;
    }
}


// one bounce row t of the backward with everything given per row (row_inputs = 2): writes d_normals[t], returns the adjoint of the
// row's heads (gh) and of its feature row (gf).  Every load stands in front of the stores (bidx -> ray_id -> ray is the only chain).
struct RowsBwdIn {
    const int32_t* bidx;
    const float* normals;
    const float* heads;
    const int32_t* ray_id;
    const float* rays;
    Conv conv;
    float min_rough;
    int detach_n;
    const float *dN, *dr1, *df0, *ddiff;
    int sN, sr, sf, sd;
    const float* dfeat;
    float* d_normals;
};
__device__ __forceinline__ void prep_bwd_row(const RowsBwdIn& q, int64_t t, float (&gh)[HEADS], float4 (&gf)[FEAT / 4]) {
    const bool want_n = !q.detach_n && q.dN;
    const int64_t m = q.bidx[t];
    const float nx = q.normals[t * 3], ny = q.normals[t * 3 + 1], nz = q.normals[t * 3 + 2];
    float dn[3] = {0.f, 0.f, 0.f}, dd[3] = {0.f, 0.f, 0.f}, d0[3] = {0.f, 0.f, 0.f}, dr = 0.f;
    if (want_n) { dn[0] = q.dN[t * q.sN]; dn[1] = q.dN[t * q.sN + 1]; dn[2] = q.dN[t * q.sN + 2]; }
    if (q.ddiff) { dd[0] = q.ddiff[t * q.sd]; dd[1] = q.ddiff[t * q.sd + 1]; dd[2] = q.ddiff[t * q.sd + 2]; }
    if (q.df0) { d0[0] = q.df0[t * q.sf]; d0[1] = q.df0[t * q.sf + 1]; d0[2] = q.df0[t * q.sf + 2]; }
    if (q.dr1) dr = q.dr1[t * q.sr];
    const float h9 = q.heads[t * HEADS + 9];
#pragma unroll
    for (int i = 0; i < FEAT / 4; ++i)
        gf[i] = q.dfeat ? reinterpret_cast<const float4*>(q.dfeat + t * FEAT)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float gn[3] = {0.f, 0.f, 0.f};
    if (want_n) {
        const float* d = q.rays + (int64_t)q.ray_id[m] * 6 + 3;
        const float s = sgn(-(d[0] * nx + d[1] * ny + d[2] * nz));
        gn[0] = dn[0] * s; gn[1] = dn[1] * s; gn[2] = dn[2] * s;
    }
    q.d_normals[t * 3] = gn[0]; q.d_normals[t * 3 + 1] = gn[1]; q.d_normals[t * 3 + 2] = gn[2];
#pragma unroll
    for (int j = 0; j < HEADS; ++j) gh[j] = 0.f;
    float E3[3];
    irradiance_E(nx, ny, nz, q.conv, E3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gh[c] = q.ddiff ? dd[c] * E3[c] : 0.f;
        gh[6 + c] = d0[c];
    }
    gh[9] = (q.dr1 && h9 >= q.min_rough) ? dr : 0.f;
}

}  // namespace nmf_rows
