// Retrace selection for gfx950 (reference: models/microfacet.py:475-537): which secondary rays are re-traced through
// the field and which only look up the environment map.
//   score_r = max_c(brdf_r) * [V.N > 0] * exp(log pdf_r) * w_row / (count_row + 1e-8)            (:480-500)
//   cc = score / sum(score) * num_retrace + U ;  order = argsort(cc) ;  re-trace order[R - num_retrace:]  (:501-537)
// nmf_retrace_scores is one streaming pass over the compact ray list; nmf_topk_select (R4) is a hand-written radix SELECT:
// the reference only ever uses the argsort as a partition (cc_as[M:] / cc_as[:M], :501-537), so three histogram passes (11 / 11 / 10 bits)
// find the key of rank n - k, a count + scatter pair splits the indices, and only the k selected ones are sorted (the
// early phase re-traces 1000 of 0.24 M rays: one workgroup sorts them in LDS).  nmf_argsort_f32 is an LSD radix sort of
// (key, index) pairs (rocPRIM device primitive -- a library sort, like the reference's torch.argsort) on the caller's
// workspace: the full order, for traces and for selections of more than NMF_TOPK_SORT_IN_LDS rays.  In the steady state every
// ray is re-traced and none of them runs.
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
    NMF_LAUNCH(k_retrace_scores, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, brdf, V_rows,
                       N_rows, lpdf, w_rows, cnt_rows, row_of_ray, R, score);
    NMF_CHECK_LAUNCH("nmf_retrace_scores");
    return NMF_OK;
}

// ---- radix select ----------------------------------------------------------------------------------------------------
namespace {

constexpr int SEL_THREADS = 1024, SEL_PER = 2, SEL_CHUNK = SEL_THREADS * SEL_PER;
constexpr int TOPK_SORT_IN_LDS = 4096;

// fp32 -> uint32 whose unsigned order is the ascending float order (-0.0 < +0.0 like a stable radix sort of the bits; NaNs last)
__device__ __forceinline__ uint32_t f2u(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

constexpr int SEL_PASSES = 3;
constexpr int SEL_BITS[3] = {11, 11, 10};       // digit widths, most significant first
constexpr int SEL_BINS = 2048;

struct SelState {
    uint32_t hist[SEL_PASSES][SEL_BINS];
    uint32_t ticket[8];
    uint32_t prefix;        // the digits of the threshold key found so far
    uint32_t pad;
    int64_t rank;           // rank of the threshold among the keys that share `prefix` (ascending)
    int64_t n_gt, n_tie;    // after the count pass: keys above the threshold / equal to it
};

// pass P (most significant digit first): histogram of digit P over the keys that match the prefix of the earlier passes; the
// workgroup that finishes last scans the bins, fixes the digit and leaves the rank inside that bin.  Scores are products of
// probabilities: nearly all keys share their leading digit, so lanes with the digit of the wave's first active lane are counted
// with one ballot (two rounds of that), only the others go through LDS atomics one by one.
template <int P>
__global__ void __launch_bounds__(SEL_THREADS) k_sel_hist(const float* __restrict__ keys, int64_t n, int64_t k,
                                                          SelState* __restrict__ st) {
    constexpr int BITS = SEL_BITS[P];
    constexpr int HI = P == 0 ? 0 : (P == 1 ? SEL_BITS[0] : SEL_BITS[0] + SEL_BITS[1]);     // bits fixed by the earlier passes
    constexpr int NB = 1 << BITS;
    __shared__ uint32_t h[SEL_BINS];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int b = tid; b < NB; b += SEL_THREADS) h[b] = 0;
    __syncthreads();
    const uint32_t prefix = P ? st->prefix : 0u;
    const int64_t i0 = (int64_t)blockIdx.x * SEL_CHUNK + tid;
#pragma unroll
    for (int q = 0; q < SEL_PER; ++q) {
        const int64_t i = i0 + (int64_t)q * SEL_THREADS;
        bool act = false;
        uint32_t d = 0;
        if (i < n) {
            const uint32_t u = f2u(keys[i]);
            act = P == 0 || (u >> (32 - HI)) == prefix;
            d = (u >> (32 - HI - BITS)) & (uint32_t)(NB - 1);
        }
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            const uint64_t am = __ballot(act);
            if (am == 0) break;
            const int first = __ffsll((long long)am) - 1;
            const uint32_t d0 = __shfl(d, first, 64);
            const uint64_t same = __ballot(act && d == d0);
            if (lane == first) atomicAdd(&h[d0], (uint32_t)__popcll(same));
            if (d == d0) act = false;
        }
        if (act) atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    for (int b = tid; b < NB; b += SEL_THREADS)
        if (h[b]) atomicAdd(&st->hist[P][b], h[b]);
}

// the bin that holds rank r (one workgroup, behind the histogram launch: a "last workgroup finishes the job" ticket costs a
// device-scope fence per workgroup, which on this part -- eight L2s -- was 0.3 us per workgroup, 35 us per pass):
// every thread takes NB / 1024 consecutive bins, block scan of the partial sums
template <int P>
__global__ void __launch_bounds__(SEL_THREADS) k_sel_pick(int64_t n, int64_t k, SelState* __restrict__ st) {
    constexpr int BITS = SEL_BITS[P];
    constexpr int NB = 1 << BITS;
    __shared__ uint32_t wsum[SEL_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t prefix = P ? st->prefix : 0u;
    const int64_t r = P ? st->rank : (n - k);          // ascending rank of the smallest selected key
    constexpr int PER = NB / SEL_THREADS > 0 ? NB / SEL_THREADS : 1;
    uint32_t c[PER], tsum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int b = tid * PER + q;
        c[q] = b < NB ? st->hist[P][b] : 0u;
        tsum += c[q];
    }
    uint32_t incl = tsum;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const uint32_t t = __shfl_up(incl, dd, 64);
        if (lane >= dd) incl += t;
    }
    if (lane == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < (tid >> 6); ++w) off += wsum[w];
    int64_t lo = (int64_t)off + incl - tsum;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int b = tid * PER + q;
        const int64_t hi = lo + c[q];
        if (b < NB && ((r >= lo && r < hi) || (b == NB - 1 && r >= hi))) {     // (rank past every key: the largest digit)
            st->prefix = (prefix << BITS) | (uint32_t)b;
            st->rank = r - lo;
        }
        lo = hi;
    }
}

__device__ __forceinline__ int64_t sel_flags(const float* __restrict__ keys, int64_t i, int64_t n, uint32_t T) {
    if (i >= n) return 0;
    const uint32_t u = f2u(keys[i]);
    return u > T ? 1ll : (u == T ? (1ll << 32) : 0ll);          // (ties << 32) | greater
}

// block-wide inclusive scan of a packed (ties << 32 | greater) count
__device__ __forceinline__ int64_t sel_block_scan(int64_t v, int64_t* ws, int64_t& total) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) ws[wid] = incl;
    __syncthreads();
    int64_t off = 0, tot = 0;
    for (int w = 0; w < SEL_THREADS / 64; ++w) {
        const int64_t t = ws[w];
        if (w < wid) off += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return incl + off;
}

// per-chunk counts of keys above / equal to the threshold
__global__ void __launch_bounds__(SEL_THREADS) k_sel_count(const float* __restrict__ keys, int64_t n, const SelState* __restrict__ st,
                                                           int64_t* __restrict__ chunk) {
    __shared__ int64_t ws[SEL_THREADS / 64];
    const int tid = threadIdx.x;
    const uint32_t T = st->prefix;
    const int64_t i0 = (int64_t)blockIdx.x * SEL_CHUNK + (int64_t)tid * SEL_PER;
    int64_t v = 0;
#pragma unroll
    for (int q = 0; q < SEL_PER; ++q) v += sel_flags(keys, i0 + q, n, T);
    int64_t total;
    sel_block_scan(v, ws, total);
    if (tid == 0) chunk[blockIdx.x] = total;
}

// -> exclusive chunk offsets (one workgroup)
__global__ void __launch_bounds__(SEL_THREADS) k_sel_chunk_scan(SelState* __restrict__ st, int64_t* __restrict__ chunk, int n_chunks) {
    __shared__ int64_t ws[SEL_THREADS / 64];
    const int tid = threadIdx.x;
    int64_t carry = 0;
    for (int base = 0; base < n_chunks; base += SEL_THREADS) {
        const int c = base + tid;
        const int64_t x = c < n_chunks ? chunk[c] : 0;
        int64_t tot;
        const int64_t incl = sel_block_scan(x, ws, tot);
        if (c < n_chunks) chunk[c] = carry + incl - x;
        carry += tot;
    }
    if (tid == 0) { st->n_gt = carry & 0xffffffffll; st->n_tie = carry >> 32; }
}

// indices above the threshold (and the ties with the highest indices, as a stable ascending sort would place them) -> top, in
// index order; everything else -> rest, in index order.  key64 = (sortable key << 32 | index) of the top entries for the sort.
__global__ void __launch_bounds__(SEL_THREADS) k_sel_scatter(const float* __restrict__ keys, int64_t n, int64_t k,
                                                             const SelState* __restrict__ st, const int64_t* __restrict__ chunk,
                                                             int32_t* __restrict__ top, int32_t* __restrict__ rest,
                                                             uint64_t* __restrict__ key64) {
    __shared__ int64_t ws[SEL_THREADS / 64];
    const int tid = threadIdx.x;
    const uint32_t T = st->prefix;
    const int64_t skip_ties = st->rank;                 // ties below the cut (lowest indices)
    const int64_t i0 = (int64_t)blockIdx.x * SEL_CHUNK + (int64_t)tid * SEL_PER;
    int64_t f[SEL_PER], v = 0;
#pragma unroll
    for (int q = 0; q < SEL_PER; ++q) { f[q] = sel_flags(keys, i0 + q, n, T); v += f[q]; }
    int64_t total;
    int64_t run = chunk[blockIdx.x] + sel_block_scan(v, ws, total) - v;       // exclusive, packed
#pragma unroll
    for (int q = 0; q < SEL_PER; ++q) {
        const int64_t i = i0 + q;
        if (i < n) {
            const int64_t gt_before = run & 0xffffffffll, tie_before = run >> 32;
            const int64_t taken_before = tie_before > skip_ties ? tie_before - skip_ties : 0;
            const bool is_gt = (f[q] & 1) != 0, is_tie = (f[q] >> 32) != 0;
            const bool sel = is_gt || (is_tie && tie_before >= skip_ties);
            const int64_t pos_top = gt_before + taken_before;
            if (sel) {
                if (pos_top < k) {
                    top[pos_top] = (int32_t)i;
                    if (key64) key64[pos_top] = ((uint64_t)f2u(keys[i]) << 32) | (uint32_t)i;
                }
            } else if (rest) {
                rest[i - pos_top] = (int32_t)i;
            }
        }
        run += f[q];
    }
}

// ascending (key, index) order of up to TOPK_SORT_IN_LDS selected entries: bitonic sort of the 64-bit keys in LDS, one workgroup
__global__ void __launch_bounds__(SEL_THREADS) k_sel_sort(const uint64_t* __restrict__ key64, int k, int32_t* __restrict__ top) {
    __shared__ uint64_t s[TOPK_SORT_IN_LDS];
    int m = 1;
    while (m < k) m <<= 1;
    for (int i = threadIdx.x; i < m; i += SEL_THREADS) s[i] = i < k ? key64[i] : ~0ull;
    __syncthreads();
    for (int size = 2; size <= m; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (m >> 1); t += SEL_THREADS) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t a = s[lo], b = s[hi];
                if ((a > b) == up) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < k; i += SEL_THREADS) top[i] = (int32_t)(s[i] & 0xffffffffull);
}

__global__ void __launch_bounds__(256) k_sel_gather_keys(const float* __restrict__ keys, const int32_t* __restrict__ idx, int64_t k,
                                                         float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) out[i] = keys[idx[i]];
}
__global__ void __launch_bounds__(256) k_sel_compose(const int32_t* __restrict__ idx, const int32_t* __restrict__ order, int64_t k,
                                                     int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) out[i] = idx[order[i]];
}

}  // namespace

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
    NMF_LAUNCH(k_iota, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, iota, n);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_out, (const int32_t*)iota, order, (size_t)n, 0,
                                             32, st);
    if (e != hipSuccess) return nmf_fail((int)e, "nmf_argsort_f32: rocprim::radix_sort_pairs");
    NMF_CHECK_LAUNCH("nmf_argsort_f32");
    return NMF_OK;
}

// workspace: [SelState][chunk offsets][key64 k <= n][top indices in index order][keys of the top entries][argsort scratch]
extern "C" int64_t nmf_topk_select_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    const int64_t n_chunks = cdiv(n, SEL_CHUNK);
    return (int64_t)(align256(sizeof(SelState)) + align256((size_t)n_chunks * 8) + align256((size_t)n * 8) + 3 * align256((size_t)n * 4)) +
           nmf_argsort_workspace_bytes(n) + 256;
}

extern "C" int nmf_topk_select(const float* keys, int64_t n, int64_t k, int32_t* idx_top, int32_t* idx_rest, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    NMF_REQUIRE(n >= 0 && k >= 0 && k <= n, NMF_EINVAL, "nmf_topk_select: 0 <= k <= n");
    if (n == 0) return NMF_OK;
    NMF_REQUIRE(keys && workspace && (k == 0 || idx_top) && (k == n || idx_rest), NMF_EINVAL, "nmf_topk_select: null");
    NMF_REQUIRE(n < (1ll << 31), NMF_ERANGE, "nmf_topk_select: n >= 2^31");
    NMF_REQUIRE(workspace_bytes >= nmf_topk_select_workspace_bytes(n), NMF_EINVAL, "nmf_topk_select: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    SelState* state = (SelState*)base;                      base += align256(sizeof(SelState));
    const int64_t n_chunks = cdiv(n, SEL_CHUNK);
    int64_t* chunk = (int64_t*)base;                        base += align256((size_t)n_chunks * 8);
    uint64_t* key64 = (uint64_t*)base;                      base += align256((size_t)n * 8);
    int32_t* top_raw = (int32_t*)base;                      base += align256((size_t)n * 4);
    float* top_keys = (float*)base;                         base += align256((size_t)n * 4);
    int32_t* order = (int32_t*)base;                        base += align256((size_t)n * 4);
    void* sort_ws = base;
    if (k == 0) {                     // nothing selected: the rest is every index
        NMF_LAUNCH(k_iota, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, idx_rest, n);
        NMF_CHECK_LAUNCH("nmf_topk_select");
        return NMF_OK;
    }
    if (k == n) return nmf_argsort_f32(keys, n, idx_top, sort_ws, nmf_argsort_workspace_bytes(n), stream);      // the whole order
    hipError_t e = hipMemsetAsync(state, 0, sizeof(SelState), st);
    if (e != hipSuccess) return nmf_fail((int)e, "nmf_topk_select: memset");
    const dim3 grid((unsigned)n_chunks), block(SEL_THREADS);
    NMF_LAUNCH(k_sel_hist<0>, grid, block, 0, st, keys, n, k, state);
    NMF_LAUNCH(k_sel_pick<0>, dim3(1), block, 0, st, n, k, state);
    NMF_LAUNCH(k_sel_hist<1>, grid, block, 0, st, keys, n, k, state);
    NMF_LAUNCH(k_sel_pick<1>, dim3(1), block, 0, st, n, k, state);
    NMF_LAUNCH(k_sel_hist<2>, grid, block, 0, st, keys, n, k, state);
    NMF_LAUNCH(k_sel_pick<2>, dim3(1), block, 0, st, n, k, state);
    NMF_LAUNCH(k_sel_count, grid, block, 0, st, keys, n, state, chunk);
    NMF_LAUNCH(k_sel_chunk_scan, dim3(1), block, 0, st, state, chunk, (int)n_chunks);
    const bool in_lds = k <= TOPK_SORT_IN_LDS;
    NMF_LAUNCH(k_sel_scatter, grid, block, 0, st, keys, n, k, state, chunk, k ? top_raw : nullptr, idx_rest,
                       (k && in_lds) ? key64 : nullptr);
    if (in_lds) {
        NMF_LAUNCH(k_sel_sort, dim3(1), block, 0, st, key64, (int)k, idx_top);
    } else {             // many selected rays (a transient of the re-trace controller): the library sort, over the k keys only
        NMF_LAUNCH(k_sel_gather_keys, dim3((unsigned)cdiv(k, 256)), dim3(256), 0, st, keys, top_raw, k, top_keys);
        const int rc = nmf_argsort_f32(top_keys, k, order, sort_ws, nmf_argsort_workspace_bytes(k), stream);
        if (rc != NMF_OK) return rc;
        NMF_LAUNCH(k_sel_compose, dim3((unsigned)cdiv(k, 256)), dim3(256), 0, st, top_raw, order, k, idx_top);
    }
    NMF_CHECK_LAUNCH("nmf_topk_select");
    return NMF_OK;
}
