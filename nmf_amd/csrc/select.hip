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
    // eight lanes per segment, consecutive lanes on consecutive elements (a thread per segment wrote its ~30 elements one
    // after the other: 7 k threads for 0.24 M elements, 12 us)
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    if (i >= n_seg) return;
    const int64_t s = offsets[i], e = offsets[i + 1];
    for (int64_t r = s + (threadIdx.x & 7); r < e; r += 8) {
        if (seg_id) seg_id[r] = (int32_t)i;
        if (local) local[r] = (int32_t)(r - s);
    }
}

// total = clip(float(sum(w) + 1e-3 * (sum(u) + extra)), 1e-3) with float64 sums: the normaliser of the level >= 1 bounce
// selection (modules/pt_selectors.py:24-31: the dense weight matrix perturbed by 1e-3 U, then divided by its sum).  One
// launch: every workgroup stores its partial sums, the last one to arrive (ticket counter) adds them in workgroup order,
// writes the result and resets the ticket for the next call.  ws = {ticket, unused, (sum_w, sum_u) per workgroup}.
__global__ void __launch_bounds__(256) k_select_total(const float* __restrict__ w, const float* __restrict__ u, int64_t M,
                                                      double extra, double* __restrict__ ws, float* __restrict__ total) {
    double a = 0.0, b = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t done = 0;
    if (((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(u)) & 15) == 0) {
        // 16-byte runs, two independent partial sums per array: the scalar loop was a 27-deep chain of dependent loads + adds
        const float4* w4 = reinterpret_cast<const float4*>(w);
        const float4* u4 = reinterpret_cast<const float4*>(u);
        const int64_t n4 = M >> 2;
        double a1 = 0.0, b1 = 0.0;
        for (int64_t i = tid; i < n4; i += stride) {
            const float4 x = w4[i], y = u4[i];
            a += (double)x.x + (double)x.y; a1 += (double)x.z + (double)x.w;
            b += (double)y.x + (double)y.y; b1 += (double)y.z + (double)y.w;
        }
        a += a1; b += b1;
        done = n4 << 2;
    }
    for (int64_t i = done + tid; i < M; i += stride) { a += (double)w[i]; b += (double)u[i]; }
    for (int d = 32; d > 0; d >>= 1) { a += __shfl_down(a, d, 64); b += __shfl_down(b, d, 64); }
    __shared__ double sa[4], sb[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sa[wave] = a; sb[wave] = b; }
    __syncthreads();
    // per-workgroup partial sums, added by the last workgroup in a FIXED order (strided pairs, then a shuffle tree): the total
    // does not depend on the order in which the workgroups finish (float64 atomics did: the last fp32 bit of `total`, hence a
    // floor() in select_bounces, could differ from run to run under a fixed seed -- ADVICE round 2)
    __shared__ int is_last;
    if (threadIdx.x == 0) {
        ws[2 + 2 * blockIdx.x] = sa[0] + sa[1] + sa[2] + sa[3];
        ws[3 + 2 * blockIdx.x] = sb[0] + sb[1] + sb[2] + sb[3];
        __threadfence();
        unsigned long long* ticket = reinterpret_cast<unsigned long long*>(ws);
        is_last = atomicAdd(ticket, 1ull) + 1ull == (unsigned long long)gridDim.x;
        if (is_last) *ticket = 0ull;
    }
    __syncthreads();
    if (!is_last || threadIdx.x >= 64) return;
    __threadfence();
    double sw = 0.0, su = 0.0;
    for (unsigned bb = threadIdx.x; bb < gridDim.x; bb += 64) {       // <= 128 workgroups: two per lane
        sw += __builtin_nontemporal_load(&ws[2 + 2 * bb]);
        su += __builtin_nontemporal_load(&ws[3 + 2 * bb]);
    }
    for (int d = 32; d > 0; d >>= 1) { sw += __shfl_down(sw, d, 64); su += __shfl_down(su, d, 64); }
    if (threadIdx.x == 0) *total = fmaxf((float)(sw + 1e-3 * (su + extra)), 1e-3f);
}

// d_rays[ray_of(row)][3 + k] -= dv_a[row][k] + dv_b[row][k]: the adjoint of the rows' view vector V = -direction scattered
// back onto the rays the rows belong to (ray_id[bidx[row]]); rows of one ray are few, plain atomics
__global__ void __launch_bounds__(256) k_view_adjoint_to_rays(const int32_t* __restrict__ ray_id,
                                                              const int32_t* __restrict__ bidx,
                                                              const float* __restrict__ dv_a, int lda,
                                                              const float* __restrict__ dv_b, int ldb, int64_t Mb,
                                                              float* __restrict__ d_rays) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Mb) return;
    // (the six values are requested before the first atomic: loads do not move across atomics by themselves, and the launch was a
    // chain of seven memory round trips for 82 instructions)
    const int64_t m = bidx[r];
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = dv_a[r * lda + k];
    if (dv_b) {
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] += dv_b[r * ldb + k];
    }
    const int64_t ray = ray_id[m];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (v[k] != 0.f) atomicAdd(&d_rays[ray * 6 + 3 + k], -v[k]);
}

}  // namespace

extern "C" int nmf_select_total(const float* weights, const float* u, int64_t M, double extra, double* workspace,
                                float* total, void* stream) {
    NMF_REQUIRE(M > 0 && weights && u && workspace && total, NMF_EINVAL, "nmf_select_total: null / empty");
    // every workgroup ends in two float64 atomics on the same words, a fence and the ticket: 128 workgroups (0.9 M
    // samples: 14.5 us) beat 512 (24 us) and 64 (20.7 us)
    int64_t blocks = cdiv(M, 256 * 8);
    blocks = blocks > 128 ? 128 : (blocks < 1 ? 1 : blocks);
    NMF_LAUNCH(k_select_total, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, weights, u, M, extra,
                       workspace, total);
    NMF_CHECK_LAUNCH("nmf_select_total");
    return NMF_OK;
}

extern "C" int nmf_view_adjoint_to_rays(const int32_t* ray_id, const int32_t* bidx, const float* dv_a, int32_t lda,
                                        const float* dv_b, int32_t ldb, int64_t Mb, float* d_rays, void* stream) {
    NMF_REQUIRE(Mb >= 0, NMF_EINVAL, "nmf_view_adjoint_to_rays: Mb < 0");
    if (Mb == 0) return NMF_OK;
    NMF_REQUIRE(ray_id && bidx && dv_a && d_rays && lda >= 3 && (!dv_b || ldb >= 3), NMF_EINVAL,
                "nmf_view_adjoint_to_rays: null");
    NMF_LAUNCH(k_view_adjoint_to_rays, dim3((unsigned)cdiv(Mb, 256)), dim3(256), 0, (hipStream_t)stream, ray_id, bidx,
                       dv_a, (int)lda, dv_b, (int)ldb, Mb, d_rays);
    NMF_CHECK_LAUNCH("nmf_view_adjoint_to_rays");
    return NMF_OK;
}

extern "C" int nmf_select_bounces(const float* weights, const float* u, int64_t M, int32_t mode, float mul, float add,
                                  float sum_w, const float* sum_w_dev, int32_t* counts, void* stream) {
    NMF_REQUIRE(M >= 0, NMF_EINVAL, "nmf_select_bounces: M < 0");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(weights && u && counts, NMF_EINVAL, "nmf_select_bounces: null");
    NMF_REQUIRE(mode == 0 || mode == 1, NMF_EINVAL, "nmf_select_bounces: mode");
    NMF_LAUNCH(k_select_bounces, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, weights, u, M,
                       mode, mul, add, sum_w, sum_w_dev, counts);
    NMF_CHECK_LAUNCH("nmf_select_bounces");
    return NMF_OK;
}

extern "C" int nmf_expand_segments(const int64_t* offsets, int64_t n_seg, int32_t* seg_id, int32_t* local,
                                   void* stream) {
    NMF_REQUIRE(n_seg >= 0, NMF_EINVAL, "nmf_expand_segments: n_seg < 0");
    if (n_seg == 0) return NMF_OK;
    NMF_REQUIRE(offsets && (seg_id || local), NMF_EINVAL, "nmf_expand_segments: null");
    NMF_LAUNCH(k_expand_segments, dim3((unsigned)cdiv(n_seg, 32)), dim3(256), 0, (hipStream_t)stream, offsets,
                       n_seg, seg_id, local);
    NMF_CHECK_LAUNCH("nmf_expand_segments");
    return NMF_OK;
}
