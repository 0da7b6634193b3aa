// Alpha compositing over compacted per-ray segments for gfx950.
// Replaces raw2alpha (modules/tensor_nerf.py:19-35) and row_mask_sum
// (modules/row_mask_sum.py:15-22) of the reference.
//
// The reference runs cumprod over the DENSE [rays x N] matrix; culled steps have sigma = 0, i.e.
// alpha = 0 and a transmittance factor fp32(1 - 0 + 1e-10) == 1.0f exactly, so walking only the
// kept samples of a ray in order is bit-equivalent.  torch's CPU cumprod keeps a float64 running
// product and rounds every element to fp32; the same is done here, one lane per ray
// (segments are short: ~40 samples for a primary ray, ~4 for a secondary ray).
#include "common.hpp"

namespace {

__global__ void __launch_bounds__(256) k_composite_fwd(const float* __restrict__ sigma,
                                                       const float* __restrict__ dist,
                                                       const int64_t* __restrict__ offsets, int64_t b, float scale,
                                                       float* __restrict__ weight, float* __restrict__ acc) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b) return;
    const int64_t s = offsets[r], e = offsets[r + 1];
    double T = 1.0;
    float a_sum = 0.f;
    for (int64_t k = s; k < e; ++k) {
        const float d = fmul(dist[k], scale);                       // dists * distance_scale (:366)
        const float alpha = 1.0f - expf(-fmul(sigma[k], d));        // :22
        const float w = fmul(alpha, (float)T);                      // :34
        weight[k] = w;
        a_sum += w;
        const float f = fadd(fsub(1.0f, alpha), 1e-10f);            // :28
        T *= (double)f;
    }
    if (acc) acc[r] = a_sum;
}

// d sigma from d weight:  w_k = a_k T_k,  T_k = prod_{j<k} f_j,  f_j = 1 - a_j + 1e-10
//   dL/da_k = dw_k T_k - (sum_{j>k} dw_j w_j) / f_k ;   da_k/dsigma_k = d_k (1 - a_k)
__global__ void __launch_bounds__(256) k_composite_bwd(const float* __restrict__ sigma,
                                                       const float* __restrict__ dist,
                                                       const float* __restrict__ weight,
                                                       const int64_t* __restrict__ offsets, int64_t b, float scale,
                                                       const float* __restrict__ d_weight,
                                                       float* __restrict__ d_sigma) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b) return;
    const int64_t s = offsets[r], e = offsets[r + 1];
    double suffix = 0.0;
    for (int64_t k = e - 1; k >= s; --k) {
        const float d = fmul(dist[k], scale);
        const float ex = expf(-fmul(sigma[k], d));
        const float alpha = 1.0f - ex;
        const float f = fadd(fsub(1.0f, alpha), 1e-10f);
        const float w = weight[k];
        const float dw = d_weight[k];
        // T_k = w_k / a_k is ill-conditioned for tiny alpha; recover it from the suffix-free identity
        // T_k = T_{k+1} / f_k is equally lossy, so recompute forward products lazily: we carry
        // sum_{j>k} dw_j w_j instead and obtain T_k from a second forward sweep below.
        d_sigma[k] = (float)(-(suffix / (double)f)) * (d * ex);     // second term; first term added below
        suffix += (double)dw * (double)w;
    }
    double T = 1.0;
    for (int64_t k = s; k < e; ++k) {
        const float d = fmul(dist[k], scale);
        const float ex = expf(-fmul(sigma[k], d));
        const float alpha = 1.0f - ex;
        const float f = fadd(fsub(1.0f, alpha), 1e-10f);
        d_sigma[k] += d_weight[k] * (float)T * (d * ex);
        T *= (double)f;
    }
}

// ---- wave-per-ray variants ---------------------------------------------------------------------------------------
// A primary-ray batch is few rays (4096) with long segments (~45 samples): one lane per ray leaves 16 workgroups walking
// 45-deep dependent chains (0.06 / 0.15 ms).  Here one WAVE owns a ray: 64 samples per pass, the transmittance is a
// float64 multiplicative wave scan (same products as the sequential walk up to float64 rounding, i.e. identical after
// the per-element rounding to fp32 except for rare ties), carried across passes.
constexpr int WPR_MAX_RAYS = 16384;     // batches up to this size use the wave-per-ray kernels

__device__ __forceinline__ double wave_incl_prod(double v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double t = __shfl_up(v, d, 64);
        if (lane >= d) v *= t;
    }
    return v;
}

__global__ void __launch_bounds__(256) k_composite_fwd_wave(const float* __restrict__ sigma,
                                                            const float* __restrict__ dist,
                                                            const int64_t* __restrict__ offsets, int64_t b, float scale,
                                                            float* __restrict__ weight, float* __restrict__ acc) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= b) return;
    const int lane = lane_id();
    const int64_t s = offsets[r], e = offsets[r + 1];
    double carry = 1.0;
    float a_sum = 0.f;
    for (int64_t k0 = s; k0 < e; k0 += 64) {
        const int64_t k = k0 + lane;
        const bool in = k < e;
        float alpha = 0.f;
        if (in) alpha = 1.0f - expf(-fmul(sigma[k], fmul(dist[k], scale)));
        const float f = in ? fadd(fsub(1.0f, alpha), 1e-10f) : 1.0f;
        const double incl = wave_incl_prod((double)f);
        const double T = carry * (incl / (double)f);          // exclusive product (f >= 1e-10 > 0)
        const float w = fmul(alpha, (float)T);
        if (in) weight[k] = w;
        a_sum += in ? w : 0.f;
        carry *= __shfl(incl, 63, 64);
    }
    if (acc) {
        for (int d = 32; d > 0; d >>= 1) a_sum += __shfl_down(a_sum, d, 64);
        if (lane == 0) acc[r] = a_sum;
    }
}

__global__ void __launch_bounds__(256) k_composite_bwd_wave(const float* __restrict__ sigma,
                                                            const float* __restrict__ dist,
                                                            const float* __restrict__ weight,
                                                            const int64_t* __restrict__ offsets, int64_t b, float scale,
                                                            const float* __restrict__ d_weight,
                                                            float* __restrict__ d_sigma) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= b) return;
    const int lane = lane_id();
    const int64_t s = offsets[r], e = offsets[r + 1];
    // pass 1 (back to front): suffix_k = sum_{j>k} dw_j w_j, second term of dL/da_k
    double carry = 0.0;
    for (int64_t k1 = e; k1 > s; k1 -= 64) {
        const int64_t k = k1 - 1 - lane;                      // lane 0 = last sample of the chunk
        const bool in = k >= s;
        const double v = in ? (double)d_weight[k] * (double)weight[k] : 0.0;
        const double incl = wave_incl_scan(v);
        if (in) {
            const float d = fmul(dist[k], scale);
            const float ex = expf(-fmul(sigma[k], d));
            const float f = fadd(fsub(1.0f, 1.0f - ex), 1e-10f);
            d_sigma[k] = (float)(-((carry + incl - v) / (double)f)) * (d * ex);
        }
        carry += __shfl(incl, 63, 64);
    }
    // pass 2 (front to back): first term dw_k T_k
    double cp = 1.0;
    for (int64_t k0 = s; k0 < e; k0 += 64) {
        const int64_t k = k0 + lane;
        const bool in = k < e;
        float d = 0.f, ex = 1.f;
        if (in) { d = fmul(dist[k], scale); ex = expf(-fmul(sigma[k], d)); }
        const float f = in ? fadd(fsub(1.0f, 1.0f - ex), 1e-10f) : 1.0f;
        const double incl = wave_incl_prod((double)f);
        const double T = cp * (incl / (double)f);
        if (in) d_sigma[k] += d_weight[k] * (float)T * (d * ex);
        cp *= __shfl(incl, 63, 64);
    }
}

// out[r][:] = sum_k scale[k] * vals[k][:] in index order (fp32, like scatter_add_ on the CPU)
template <int D>
__global__ void __launch_bounds__(256) k_segment_sum(const float* __restrict__ vals, const float* __restrict__ scale,
                                                     const int64_t* __restrict__ offsets, int64_t n_seg,
                                                     float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_seg) return;
    const int64_t s = offsets[r], e = offsets[r + 1];
    float a[D];
#pragma unroll
    for (int d = 0; d < D; ++d) a[d] = 0.f;
    for (int64_t k = s; k < e; ++k) {
        const float sc = scale ? scale[k] : 1.f;
#pragma unroll
        for (int d = 0; d < D; ++d) a[d] = fadd(a[d], fmul(sc, vals[k * D + d]));
    }
#pragma unroll
    for (int d = 0; d < D; ++d) out[r * D + d] = a[d];
}

}  // namespace

extern "C" int nmf_composite_fwd(const float* sigma, const float* dist, const int64_t* offsets, int64_t b,
                                 float distance_scale, float* weight, float* acc, void* stream) {
    NMF_REQUIRE(b >= 0, NMF_EINVAL, "nmf_composite_fwd: b < 0");
    if (b == 0) return NMF_OK;
    NMF_REQUIRE(sigma && dist && offsets && weight, NMF_EINVAL, "nmf_composite_fwd: null");
    if (b <= WPR_MAX_RAYS)
        hipLaunchKernelGGL(k_composite_fwd_wave, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, (hipStream_t)stream, sigma,
                           dist, offsets, b, distance_scale, weight, acc);
    else
        hipLaunchKernelGGL(k_composite_fwd, dim3((unsigned)cdiv(b, 256)), dim3(256), 0, (hipStream_t)stream, sigma, dist,
                           offsets, b, distance_scale, weight, acc);
    NMF_CHECK_LAUNCH("nmf_composite_fwd");
    return NMF_OK;
}

extern "C" int nmf_composite_bwd(const float* sigma, const float* dist, const float* weight, const int64_t* offsets,
                                 int64_t b, float distance_scale, const float* d_weight, float* d_sigma,
                                 void* stream) {
    NMF_REQUIRE(b >= 0, NMF_EINVAL, "nmf_composite_bwd: b < 0");
    if (b == 0) return NMF_OK;
    NMF_REQUIRE(sigma && dist && weight && offsets && d_weight && d_sigma, NMF_EINVAL, "nmf_composite_bwd: null");
    if (b <= WPR_MAX_RAYS)
        hipLaunchKernelGGL(k_composite_bwd_wave, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, (hipStream_t)stream, sigma,
                           dist, weight, offsets, b, distance_scale, d_weight, d_sigma);
    else
        hipLaunchKernelGGL(k_composite_bwd, dim3((unsigned)cdiv(b, 256)), dim3(256), 0, (hipStream_t)stream, sigma, dist,
                           weight, offsets, b, distance_scale, d_weight, d_sigma);
    NMF_CHECK_LAUNCH("nmf_composite_bwd");
    return NMF_OK;
}

extern "C" int nmf_segment_sum(const float* vals, const float* scale, const int64_t* offsets, int64_t n_seg,
                               int32_t D, float* out, void* stream) {
    NMF_REQUIRE(n_seg >= 0, NMF_EINVAL, "nmf_segment_sum: n_seg < 0");
    if (n_seg == 0) return NMF_OK;
    NMF_REQUIRE(vals && offsets && out, NMF_EINVAL, "nmf_segment_sum: null");
    dim3 grid((unsigned)cdiv(n_seg, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 1: hipLaunchKernelGGL(k_segment_sum<1>, grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
        case 2: hipLaunchKernelGGL(k_segment_sum<2>, grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
        case 3: hipLaunchKernelGGL(k_segment_sum<3>, grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
        case 4: hipLaunchKernelGGL(k_segment_sum<4>, grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
        default: return nmf_fail(NMF_ERANGE, "nmf_segment_sum: D must be 1..4");
    }
    NMF_CHECK_LAUNCH("nmf_segment_sum");
    return NMF_OK;
}
