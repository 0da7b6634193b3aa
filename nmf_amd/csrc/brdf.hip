// Wide segmented sums for gfx950: per-segment sums of rows with up to 64 columns (adjoints of the gathered BRDF-MLP
// feature rows, of (f0 | diffuse) and of (normal | roughness | view) per bounce point; reference: the autograd of the
// `[ri, rj]` expansions in models/microfacet.py:377-385).
#include "common.hpp"

namespace {



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
    // four rows of a slot in flight (one dependent load per iteration made a 33-row segment 17 memory round trips long)
    float a = 0.f;
    if (ch < D) {
        float a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int64_t r = b + slot;
        for (; r + 3 * SLOTS < e; r += 4 * SLOTS) {
            const float v0 = vals[r * stride + ch], v1 = vals[(r + SLOTS) * stride + ch];
            const float v2 = vals[(r + 2 * SLOTS) * stride + ch], v3 = vals[(r + 3 * SLOTS) * stride + ch];
            a += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; r < e; r += SLOTS) a += vals[r * stride + ch];
        a = (a + a1) + (a2 + a3);
    }
#pragma unroll
    for (int sh = 32; sh >= G; sh >>= 1) a += __shfl_down(a, sh, 64);
    if (lane < D) out[s * D + lane] = a;
}

}  // namespace


extern "C" int nmf_segment_sum_wide(const float* vals, int64_t row_stride, int32_t D, const int64_t* offsets,
                                    int64_t n_seg, float* out, void* stream) {
    NMF_REQUIRE(n_seg >= 0, NMF_EINVAL, "nmf_segment_sum_wide: n_seg < 0");
    if (n_seg == 0) return NMF_OK;
    NMF_REQUIRE(vals && offsets && out && D > 0 && D <= 64 && row_stride >= D, NMF_EINVAL, "nmf_segment_sum_wide: args");
    const dim3 grid((unsigned)cdiv(n_seg, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (D <= 8) NMF_LAUNCH(k_segment_sum_wide<8>, grid, block, 0, st, vals, row_stride, D, offsets, n_seg, out);
    else if (D <= 16) NMF_LAUNCH(k_segment_sum_wide<16>, grid, block, 0, st, vals, row_stride, D, offsets, n_seg, out);
    else if (D <= 32) NMF_LAUNCH(k_segment_sum_wide<32>, grid, block, 0, st, vals, row_stride, D, offsets, n_seg, out);
    else NMF_LAUNCH(k_segment_sum_wide<64>, grid, block, 0, st, vals, row_stride, D, offsets, n_seg, out);
    NMF_CHECK_LAUNCH("nmf_segment_sum_wide");
    return NMF_OK;
}
