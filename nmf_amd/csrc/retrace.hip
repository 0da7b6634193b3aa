// Retrace selection for gfx950 (reference: models/microfacet.py:475-537): which secondary rays are re-traced through
// the field and which only look up the environment map.
//   score_r = max_c(brdf_r) * [V.N > 0] * exp(log pdf_r) * w_row / (count_row + 1e-8)            (:480-500)
//   cc = score / sum(score) * num_retrace + U ;  order = argsort(cc) ;  re-trace order[R - num_retrace:]  (:501-537)
// nmf_retrace_scores is one streaming pass over the compact ray list; nmf_argsort_f32 is an LSD radix sort of
// (key, index) pairs (rocPRIM device primitive -- a library sort, like the reference's torch.argsort) on the caller's
// workspace.  In the steady state every ray is re-traced and neither runs.
#include "common.hpp"

#include <rocprim/rocprim.hpp>

namespace {

__global__ void __launch_bounds__(256) k_retrace_scores(const float* __restrict__ brdf, const float* __restrict__ V_rows,
                                                        const float* __restrict__ N_rows, const float* __restrict__ lpdf,
                                                        const float* __restrict__ w_rows,
                                                        const int32_t* __restrict__ cnt_rows,
                                                        const int32_t* __restrict__ row_of_ray, int64_t R,
                                                        float* __restrict__ score) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int64_t row = row_of_ray[i];
    const float m = fmaxf(fmaxf(brdf[i * 3], brdf[i * 3 + 1]), brdf[i * 3 + 2]);
    const float vn = V_rows[row * 3] * N_rows[row * 3] + V_rows[row * 3 + 1] * N_rows[row * 3 + 1] +
                     V_rows[row * 3 + 2] * N_rows[row * 3 + 2];
    const float per_ray = m * (vn > 0.f ? 1.f : 0.f) * expf(lpdf[i]);
    const float per_sample = w_rows[row] / ((float)cnt_rows[row] + 1e-8f);
    score[i] = per_ray * per_sample;
}

__global__ void __launch_bounds__(256) k_iota(int32_t* __restrict__ v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

size_t sort_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const float*)nullptr, (float*)nullptr, (const int32_t*)nullptr,
                                    (int32_t*)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
    return bytes;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" int nmf_retrace_scores(const float* brdf, const float* V_rows, const float* N_rows, const float* lpdf,
                                  const float* w_rows, const int32_t* cnt_rows, const int32_t* row_of_ray, int64_t R,
                                  float* score, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_retrace_scores: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(brdf && V_rows && N_rows && lpdf && w_rows && cnt_rows && row_of_ray && score, NMF_EINVAL,
                "nmf_retrace_scores: null");
    hipLaunchKernelGGL(k_retrace_scores, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, brdf, V_rows,
                       N_rows, lpdf, w_rows, cnt_rows, row_of_ray, R, score);
    NMF_CHECK_LAUNCH("nmf_retrace_scores");
    return NMF_OK;
}

extern "C" int64_t nmf_argsort_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    // [keys_out n floats][values_in n int32][rocPRIM temporary storage]
    return (int64_t)(align256((size_t)n * 4) * 2 + align256(sort_temp_bytes(n)) + 256);
}

extern "C" int nmf_argsort_f32(const float* keys, int64_t n, int32_t* order, void* workspace, int64_t workspace_bytes,
                               void* stream) {
    NMF_REQUIRE(n >= 0, NMF_EINVAL, "nmf_argsort_f32: n < 0");
    if (n == 0) return NMF_OK;
    NMF_REQUIRE(keys && order && workspace, NMF_EINVAL, "nmf_argsort_f32: null");
    NMF_REQUIRE(n < (1ll << 31), NMF_ERANGE, "nmf_argsort_f32: n >= 2^31");
    NMF_REQUIRE(workspace_bytes >= nmf_argsort_workspace_bytes(n), NMF_EINVAL, "nmf_argsort_f32: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* keys_out = (float*)base;
    int32_t* iota = (int32_t*)(base + align256((size_t)n * 4));
    void* temp = base + 2 * align256((size_t)n * 4);
    size_t temp_bytes = sort_temp_bytes(n);
    hipLaunchKernelGGL(k_iota, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, iota, n);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_out, (const int32_t*)iota, order, (size_t)n, 0,
                                             32, st);
    if (e != hipSuccess) return nmf_fail((int)e, "nmf_argsort_f32: rocprim::radix_sort_pairs");
    NMF_CHECK_LAUNCH("nmf_argsort_f32");
    return NMF_OK;
}
