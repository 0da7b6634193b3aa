// BRDF-MLP input features for gfx950: the 66-wide vector
//   [ app feature (24) | ISH(half; kappa) (18) | half (3) | ISH(diff; kappa) (18) | diff (3) ]
// of MLPBRDF.forward with feape=0, dotpe=-1, h/d encoders = ListISH([0,1,2,4])
// (reference: modules/brdf.py:177-261, modules/ish.py:94-105, modules/sh.py:251-308).
// One lane per secondary ray; the per-bounce-point feature row is gathered through src_idx so the
// reference's [R,24] "efeatures" expansion (models/microfacet.py:381-385) is never materialised.
#include "common.hpp"

namespace {

// sh_basis(degs=[0,1,2,4], dirs, kappa) -- quirks are the reference's (degree-2 entry 3 is -x*y,
// degree 4 is not attenuated by Al)
__device__ __forceinline__ void ish18(float x, float y, float z, float kappa, float* o) {
    const float k = kappa + 1e-8f;
    const float a1 = expf(-1.f / k);          // Al(1) = exp(-l(l+1)/2/(kappa+1e-8))
    const float a2 = expf(-3.f / k);
    const float xx = x * x, yy = y * y, zz = z * z;
    const float x4 = xx * xx, y4 = yy * yy, z4 = zz * zz;
    o[0] = 0.28209479177387814f;               // Al(0) = 1
    o[1] = -a1 * 0.488603f * x;
    o[2] = a1 * 0.488603f * z;
    o[3] = -a1 * 0.488603f * y;
    o[4] = a2 * 1.092548f * y * x;
    o[5] = -a2 * 1.092548f * y * z;
    o[6] = a2 * 0.315392f * (3.f * zz - 1.f);
    o[7] = -a2 * 1.092548f * x * y;
    o[8] = a2 * 0.546274f * (xx - yy);
    o[9] = 2.50334f * x * y * (xx - yy);
    o[10] = -1.77013f * y * z * (-3.f * xx + yy);
    o[11] = 0.946175f * x * y * (7.f * zz - 1.f);
    o[12] = 0.669047f * y * z * (7.f * zz - 3.f);
    o[13] = 3.70251f * z4 - 3.17358f * zz + 0.317358f;
    o[14] = 0.669047f * x * z * (7.f * zz - 3.f);
    o[15] = (0.473087f * xx - 0.473087f * yy) * (7.f * zz - 1.f);
    o[16] = 1.77013f * x * z * (xx - 3.f * yy);
    o[17] = 0.625836f * x4 - 3.755016f * xx * yy + 0.625836f * y4;
}

__global__ void __launch_bounds__(256) k_brdf_features(const float* __restrict__ half_v, const float* __restrict__ diff_v,
                                                       const float* __restrict__ feat_src,
                                                       const float* __restrict__ rough_src,
                                                       const int32_t* __restrict__ src_idx, int64_t R,
                                                       float* __restrict__ X) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int64_t b = src_idx ? src_idx[r] : r;
    float* x = X + r * NMF_MLP_IN;
    const float4* f = reinterpret_cast<const float4*>(feat_src + b * NMF_APP_DIM);
#pragma unroll
    for (int i = 0; i < NMF_APP_DIM / 4; ++i) {
        const float4 v = f[i];
        x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
    const float kappa = 1.f / (rough_src[b] + 1e-3f);      // modules/ish.py:103
    float o[18];
    const float hx = half_v[r * 3], hy = half_v[r * 3 + 1], hz = half_v[r * 3 + 2];
    ish18(hx, hy, hz, kappa, o);
#pragma unroll
    for (int i = 0; i < 18; ++i) x[24 + i] = o[i];
    x[42] = hx; x[43] = hy; x[44] = hz;
    const float dx = diff_v[r * 3], dy = diff_v[r * 3 + 1], dz = diff_v[r * 3 + 2];
    ish18(dx, dy, dz, kappa, o);
#pragma unroll
    for (int i = 0; i < 18; ++i) x[45 + i] = o[i];
    x[63] = dx; x[64] = dy; x[65] = dz;
}

// out[s][0:D] = sum over rows r in [offsets[s], offsets[s+1]) of vals[r*stride + 0:D].  One wave per segment; the wave
// is split into 64/G row slots of G = pow2 >= D lanes (2 slots for D = 24, 8 for D = 6) that take alternating rows and
// are combined with a shuffle tree.
template <int G>
__global__ void __launch_bounds__(256) k_segment_sum_wide(const float* __restrict__ vals, int64_t stride, int D,
                                                          const int64_t* __restrict__ offsets, int64_t n_seg,
                                                          float* __restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_seg) return;
    const int lane = lane_id();
    const int ch = lane & (G - 1), slot = lane / G;
    constexpr int SLOTS = 64 / G;
    const int64_t b = offsets[s], e = offsets[s + 1];
    float a = 0.f;
    if (ch < D)
        for (int64_t r = b + slot; r < e; r += SLOTS) a += vals[r * stride + ch];
#pragma unroll
    for (int sh = 32; sh >= G; sh >>= 1) a += __shfl_down(a, sh, 64);
    if (lane < D) out[s * D + lane] = a;
}

}  // namespace

extern "C" int nmf_brdf_features(const float* half_vec, const float* diff_vec, const float* feat_src,
                                 const float* rough_src, const int32_t* src_idx, int64_t R, float* X, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_brdf_features: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(half_vec && diff_vec && feat_src && rough_src && X, NMF_EINVAL, "nmf_brdf_features: null");
    hipLaunchKernelGGL(k_brdf_features, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, half_vec,
                       diff_vec, feat_src, rough_src, src_idx, R, X);
    NMF_CHECK_LAUNCH("nmf_brdf_features");
    return NMF_OK;
}

extern "C" int nmf_segment_sum_wide(const float* vals, int64_t row_stride, int32_t D, const int64_t* offsets,
                                    int64_t n_seg, float* out, void* stream) {
    NMF_REQUIRE(n_seg >= 0, NMF_EINVAL, "nmf_segment_sum_wide: n_seg < 0");
    if (n_seg == 0) return NMF_OK;
    NMF_REQUIRE(vals && offsets && out && D > 0 && D <= 64 && row_stride >= D, NMF_EINVAL, "nmf_segment_sum_wide: args");
    const dim3 grid((unsigned)cdiv(n_seg, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (D <= 8) hipLaunchKernelGGL(k_segment_sum_wide<8>, grid, block, 0, st, vals, row_stride, D, offsets, n_seg, out);
    else if (D <= 16) hipLaunchKernelGGL(k_segment_sum_wide<16>, grid, block, 0, st, vals, row_stride, D, offsets, n_seg, out);
    else if (D <= 32) hipLaunchKernelGGL(k_segment_sum_wide<32>, grid, block, 0, st, vals, row_stride, D, offsets, n_seg, out);
    else hipLaunchKernelGGL(k_segment_sum_wide<64>, grid, block, 0, st, vals, row_stride, D, offsets, n_seg, out);
    NMF_CHECK_LAUNCH("nmf_segment_sum_wide");
    return NMF_OK;
}
