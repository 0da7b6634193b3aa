// Bounce-ray budgeting for gfx950: per-sample secondary-ray counts.
// Replaces select_bounces (reference: modules/pt_selectors.py:5-60).  The reference builds a dense
// [samples x m] boolean ray_mask = arange(m) < floor(pt_limit); since m = clip(max floor(pt_limit), 0, 400)
// the row sums are simply clamp(floor(pt_limit), 0, 400) and the mask is a per-row prefix, so the
// compact representation is one int32 count per kept sample (+ an exclusive scan, nmf_march_scan).
// Compiled with -ffp-contract=off: floor() of these expressions must be bit-exact.
#include "common.hpp"

#pragma clang fp contract(off)

namespace {

__global__ void __launch_bounds__(256) k_select_bounces(const float* __restrict__ w, const float* __restrict__ u,
                                                        int64_t M, int mode, float mul, float add, float S,
                                                        const float* __restrict__ S_dev, int32_t* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    if (S_dev) S = *S_dev;
    float pt;
    if (mode == 0) {
        // recursion level 0 (:20-22): pt = w * rays_per_ray + U - 0.5
        pt = fsub(fadd(fmul(w[i], mul), u[i]), 0.5f);
    } else {
        // level >= 1 (:24-33): w' = w + 1e-3 U ; pt = w' / clip(sum w', 1e-3) * N + add
        const float wp = fadd(w[i], fmul(1e-3f, u[i]));
        pt = fadd(fmul(fdiv(wp, S), mul), add);
    }
    float f = floorf(pt);
    f = fminf(fmaxf(f, 0.f), 400.f);
    counts[i] = (int32_t)f;
}

// seg_id[r] = index of the segment that owns element r, local[r] = r - offsets[seg]
__global__ void __launch_bounds__(256) k_expand_segments(const int64_t* __restrict__ offsets, int64_t n_seg,
                                                         int32_t* __restrict__ seg_id, int32_t* __restrict__ local) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seg) return;
    const int64_t s = offsets[i], e = offsets[i + 1];
    for (int64_t r = s; r < e; ++r) {
        if (seg_id) seg_id[r] = (int32_t)i;
        if (local) local[r] = (int32_t)(r - s);
    }
}

}  // namespace

extern "C" int nmf_select_bounces(const float* weights, const float* u, int64_t M, int32_t mode, float mul, float add,
                                  float sum_w, const float* sum_w_dev, int32_t* counts, void* stream) {
    NMF_REQUIRE(M >= 0, NMF_EINVAL, "nmf_select_bounces: M < 0");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(weights && u && counts, NMF_EINVAL, "nmf_select_bounces: null");
    NMF_REQUIRE(mode == 0 || mode == 1, NMF_EINVAL, "nmf_select_bounces: mode");
    hipLaunchKernelGGL(k_select_bounces, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, weights, u, M,
                       mode, mul, add, sum_w, sum_w_dev, counts);
    NMF_CHECK_LAUNCH("nmf_select_bounces");
    return NMF_OK;
}

extern "C" int nmf_expand_segments(const int64_t* offsets, int64_t n_seg, int32_t* seg_id, int32_t* local,
                                   void* stream) {
    NMF_REQUIRE(n_seg >= 0, NMF_EINVAL, "nmf_expand_segments: n_seg < 0");
    if (n_seg == 0) return NMF_OK;
    NMF_REQUIRE(offsets && (seg_id || local), NMF_EINVAL, "nmf_expand_segments: null");
    hipLaunchKernelGGL(k_expand_segments, dim3((unsigned)cdiv(n_seg, 256)), dim3(256), 0, (hipStream_t)stream, offsets,
                       n_seg, seg_id, local);
    NMF_CHECK_LAUNCH("nmf_expand_segments");
    return NMF_OK;
}
