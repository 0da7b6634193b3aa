// Alpha compositing over compacted per-ray segments for gfx950.
// Replaces raw2alpha (modules/tensor_nerf.py:19-35) and row_mask_sum
// (modules/row_mask_sum.py:15-22) of the reference.
//
// The reference runs cumprod over the DENSE [rays x N] matrix; culled steps have sigma = 0, i.e.
// alpha = 0 and a transmittance factor fp32(1 - 0 + 1e-10) == 1.0f exactly, so walking only the
// kept samples of a ray in order is bit-equivalent.  torch's CPU cumprod keeps a float64 running
// product and rounds every element to fp32; the same is done here with a float64 scan
// over the lanes that share a ray (segments: ~40 samples for a primary ray, ~4 for a secondary ray).
//
// d sigma from d weight:  w_k = a_k T_k,  T_k = prod_{j<k} f_j,  f_j = 1 - a_j + 1e-10
//   dL/da_k = dw_k T_k - (sum_{j>k} dw_j w_j) / f_k ;   da_k/dsigma_k = d_k (1 - a_k)
#include "common.hpp"

namespace {

// ---- lane-group-per-ray variants -------------------------------------------------------------------------------------
// One lane per ray serialises a wave on its longest segment: 4096 primary rays with ~45 samples leave 16 workgroups
// walking 45-deep dependent chains (0.06 / 0.15 ms), and 0.24 M secondary rays with 3.7 samples on average but tails of
// 40 cost 0.05 / 0.10 ms.  Here a GROUP of W lanes owns a ray (W = 64 for primary-ray batches, 8 for the re-traced rays):
// W samples per pass, the transmittance is a float64 multiplicative scan inside the group (same products as the
// sequential walk up to float64 rounding, i.e. identical after the per-element rounding to fp32 except for rare ties),
// carried across passes.
constexpr int WPR_MAX_RAYS = 16384;     // batches up to this size use one wave per ray, larger ones 8 lanes per ray

// value of the lane D below (garbage where that lane is outside the row of 16: callers only use it for lane >= D inside
// groups that do not straddle rows).  Groups of 8 move lanes with DPP row_shr -- same operands, same association as the
// ds_bpermute-based __shfl_up, without the LDS-crossbar round trip per step.
template <int W, int D>
__device__ __forceinline__ double lane_below(double v) {
    if constexpr (W <= 16) {
        return dpp_f64<0x110 + D, 0xf, 0xf>(v);
    } else {
        return __shfl_up(v, D, W);
    }
}
template <int W>
__device__ __forceinline__ double group_incl_prod(double v, int lane) {
    { const double t = lane_below<W, 1>(v); if (lane >= 1) v *= t; }
    if constexpr (W > 2) { const double t = lane_below<W, 2>(v); if (lane >= 2) v *= t; }
    if constexpr (W > 4) { const double t = lane_below<W, 4>(v); if (lane >= 4) v *= t; }
    if constexpr (W > 8) { const double t = lane_below<W, 8>(v); if (lane >= 8) v *= t; }
    if constexpr (W > 16) { const double t = __shfl_up(v, 16, W); if (lane >= 16) v *= t; }
    if constexpr (W > 32) { const double t = __shfl_up(v, 32, W); if (lane >= 32) v *= t; }
    return v;
}
template <int W>
__device__ __forceinline__ double group_incl_sum(double v, int lane) {
    { const double t = lane_below<W, 1>(v); if (lane >= 1) v += t; }
    if constexpr (W > 2) { const double t = lane_below<W, 2>(v); if (lane >= 2) v += t; }
    if constexpr (W > 4) { const double t = lane_below<W, 4>(v); if (lane >= 4) v += t; }
    if constexpr (W > 8) { const double t = lane_below<W, 8>(v); if (lane >= 8) v += t; }
    if constexpr (W > 16) { const double t = __shfl_up(v, 16, W); if (lane >= 16) v += t; }
    if constexpr (W > 32) { const double t = __shfl_up(v, 32, W); if (lane >= 32) v += t; }
    return v;
}

template <int W>
__global__ void __launch_bounds__(256) k_composite_fwd_wave(const float* __restrict__ sigma,
                                                            const float* __restrict__ dist,
                                                            const int64_t* __restrict__ offsets, int64_t b, float scale,
                                                            float* __restrict__ weight, float* __restrict__ acc) {
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / W;
    const int lane = threadIdx.x & (W - 1);
    const bool ray_ok = r < b;                 // whole groups are in or out; shuffles below stay inside the group
    const int64_t s = ray_ok ? offsets[r] : 0, e = ray_ok ? offsets[r + 1] : 0;
    double carry = 1.0;
    float a_sum = 0.f;
    // the operands of chunk c + 1 are requested before chunk c is worked on (same arithmetic, same order: a ray of n samples is
    // ceil(n / W) dependent memory round trips otherwise, and a wave is as slow as its longest ray)
    float sg_n = 0.f, ds_n = 0.f;
    if (s + lane < e) { sg_n = sigma[s + lane]; ds_n = dist[s + lane]; }
    for (int64_t k0 = s; k0 < e; k0 += W) {
        const int64_t k = k0 + lane;
        const bool in = k < e;
        const float sg = sg_n, ds = ds_n;
        if (k + W < e) { sg_n = sigma[k + W]; ds_n = dist[k + W]; }
        float alpha = 0.f;
        if (in) alpha = 1.0f - expf(-fmul(sg, fmul(ds, scale)));
        const float f = in ? fadd(fsub(1.0f, alpha), 1e-10f) : 1.0f;
        const double incl = group_incl_prod<W>((double)f, lane);
        // exclusive product = the inclusive product of the lane below (a float64 division by f costs ~30 instructions and
        // two more roundings)
        const double below = lane_below<W, 1>(incl);
        const double T = carry * (lane >= 1 ? below : 1.0);
        const float w = fmul(alpha, (float)T);
        if (in) weight[k] = w;
        a_sum += in ? w : 0.f;
        carry *= __shfl(incl, W - 1, W);
    }
    if (acc) {
#pragma unroll
        for (int d = W / 2; d > 0; d >>= 1) a_sum += __shfl_down(a_sum, d, W);
        if (lane == 0 && ray_ok) acc[r] = a_sum;
    }
}

template <int W>
__global__ void __launch_bounds__(256) k_composite_bwd_wave(const float* __restrict__ sigma,
                                                            const float* __restrict__ dist,
                                                            const float* __restrict__ weight,
                                                            const int64_t* __restrict__ offsets, int64_t b, float scale,
                                                            const float* __restrict__ d_weight,
                                                            float* __restrict__ d_sigma, int one_chunk_path) {
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / W;
    const int lane = threadIdx.x & (W - 1);
    const bool ray_ok = r < b;
    const int64_t s = ray_ok ? offsets[r] : 0, e = ray_ok ? offsets[r + 1] : 0;
    if (one_chunk_path && e - s <= W) {
        // The whole ray sits in one chunk of the group (the re-traced rays keep ~4 samples): both passes below on ONE set of loads.
        // Pass 1 runs with lane 0 = last sample, pass 2 with lane 0 = first sample, exactly as in the loops -- the values change
        // lane mapping through a reversal shuffle instead of a second trip to memory, and the two terms are added in registers
        // instead of through a read-modify-write of d_sigma: the same operations on the same numbers (a carry of 0.0 / 1.0 is
        // exact), two dependent memory round trips instead of four.
        const int n = (int)(e - s);
        const bool in = lane < n;
        const int64_t k = e - 1 - lane;
        float dw = 0.f, d = 0.f, ex = 1.f, f = 1.0f;
        double v = 0.0;
        if (in) {
            dw = d_weight[k];
            v = (double)dw * (double)weight[k];
            d = fmul(dist[k], scale);
            ex = expf(-fmul(sigma[k], d));
            f = fadd(fsub(1.0f, 1.0f - ex), 1e-10f);
        }
        const double incl = group_incl_sum<W>(v, lane);
        const float A = in ? (-(float)(0.0 + incl - v) / f) * (d * ex) : 0.f;
        const int src = in ? n - 1 - lane : lane;                       // forward mapping: lane L = sample s + L
        const float d2 = __shfl(d, src, W), ex2 = __shfl(ex, src, W), dw2 = __shfl(dw, src, W), A2 = __shfl(A, src, W);
        const float f2 = in ? __shfl(f, src, W) : 1.0f;
        const double incl2 = group_incl_prod<W>((double)f2, lane);
        const double below = lane_below<W, 1>(incl2);
        const double T = 1.0 * (lane >= 1 ? below : 1.0);
        if (in) d_sigma[s + lane] = A2 + dw2 * (float)T * (d2 * ex2);
        return;
    }
    // pass 1 (back to front): suffix_k = sum_{j>k} dw_j w_j, second term of dL/da_k
    // (both passes request the operands of the next chunk before they work on the current one: see k_composite_fwd_wave)
    double carry = 0.0;
    float p_dw = 0.f, p_w = 0.f, p_di = 0.f, p_sg = 0.f;
    if (e - 1 - lane >= s) { const int64_t q = e - 1 - lane; p_dw = d_weight[q]; p_w = weight[q]; p_di = dist[q]; p_sg = sigma[q]; }
    for (int64_t k1 = e; k1 > s; k1 -= W) {
        const int64_t k = k1 - 1 - lane;                      // lane 0 = last sample of the chunk
        const bool in = k >= s;
        const float c_dw = p_dw, c_w = p_w, c_di = p_di, c_sg = p_sg;
        if (k - W >= s) { const int64_t q = k - W; p_dw = d_weight[q]; p_w = weight[q]; p_di = dist[q]; p_sg = sigma[q]; }
        const double v = in ? (double)c_dw * (double)c_w : 0.0;
        const double incl = group_incl_sum<W>(v, lane);
        if (in) {
            const float d = fmul(c_di, scale);
            const float ex = expf(-fmul(c_sg, d));
            const float f = fadd(fsub(1.0f, 1.0f - ex), 1e-10f);
            d_sigma[k] = (-(float)(carry + incl - v) / f) * (d * ex);      // float64 suffix sum, fp32 quotient
        }
        carry += __shfl(incl, W - 1, W);
    }
    // pass 2 (front to back): first term dw_k T_k
    double cp = 1.0;
    float q_di = 0.f, q_sg = 0.f, q_dw = 0.f, q_ds = 0.f;
    if (s + lane < e) { const int64_t q = s + lane; q_di = dist[q]; q_sg = sigma[q]; q_dw = d_weight[q]; q_ds = d_sigma[q]; }
    for (int64_t k0 = s; k0 < e; k0 += W) {
        const int64_t k = k0 + lane;
        const bool in = k < e;
        const float c_di = q_di, c_sg = q_sg, c_dw = q_dw, c_ds = q_ds;      // (c_ds: pass 1's term, written by this wave above)
        if (k + W < e) { const int64_t q = k + W; q_di = dist[q]; q_sg = sigma[q]; q_dw = d_weight[q]; q_ds = d_sigma[q]; }
        float d = 0.f, ex = 1.f;
        if (in) { d = fmul(c_di, scale); ex = expf(-fmul(c_sg, d)); }
        const float f = in ? fadd(fsub(1.0f, 1.0f - ex), 1e-10f) : 1.0f;
        const double incl = group_incl_prod<W>((double)f, lane);
        const double below = lane_below<W, 1>(incl);
        const double T = cp * (lane >= 1 ? below : 1.0);
        if (in) d_sigma[k] = c_ds + c_dw * (float)T * (d * ex);
        cp *= __shfl(incl, W - 1, W);
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

// the same sums with W lanes per segment (strided partial sums + a shuffle tree): not index order, for the adjoint
// reductions over the rays of a bounce point (27 rays per row at level 0), where one lane per row leaves the chip idle
template <int D, int W>
__global__ void __launch_bounds__(256) k_segment_sum_group(const float* __restrict__ vals, const float* __restrict__ scale,
                                                           const int64_t* __restrict__ offsets, int64_t n_seg,
                                                           float* __restrict__ out) {
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / W;
    const int lane = threadIdx.x & (W - 1);
    const bool ok = r < n_seg;
    const int64_t s = ok ? offsets[r] : 0, e = ok ? offsets[r + 1] : 0;
    float a[D];
#pragma unroll
    for (int d = 0; d < D; ++d) a[d] = 0.f;
    // four rows of a lane in flight (a 32-row segment on 8 lanes was four dependent memory round trips; these launches have a few
    // hundred waves and nothing else to hide them), added in the order the one-row loop added them
    int64_t k = s + lane;
    for (; k + 3 * W < e; k += 4 * W) {
        float sc[4], v[4][D];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[q] = scale ? scale[k + q * W] : 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) v[q][d] = vals[(k + q * W) * D + d];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int d = 0; d < D; ++d) a[d] += sc[q] * v[q][d];
    }
    for (; k < e; k += W) {
        const float sc = scale ? scale[k] : 1.f;
#pragma unroll
        for (int d = 0; d < D; ++d) a[d] += sc * vals[k * D + d];
    }
#pragma unroll
    for (int sh = W / 2; sh > 0; sh >>= 1) {
#pragma unroll
        for (int d = 0; d < D; ++d) a[d] += __shfl_down(a[d], sh, W);
    }
    if (ok && lane == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) out[r * D + d] = a[d];
    }
}

}  // namespace

// Lanes per ray of the large batches.  The BACKWARD takes 16 (R4): 23 k of the 247 k re-traced rays of a step keep more than 8
// samples (11 k more than 16, the longest 200) and a wave is as slow as its longest ray -- every chunk of a ray beyond the first
// costs four dependent memory round trips there; 41 -> 34 us, the same bits (tools/composite_bench.py).  The forward stays at 8:
// no faster with 16, and its per-ray opacity sum would change its association.
static int group_width(int dflt) { return dflt; }

extern "C" int nmf_composite_fwd(const float* sigma, const float* dist, const int64_t* offsets, int64_t b,
                                 float distance_scale, float* weight, float* acc, void* stream) {
    NMF_REQUIRE(b >= 0, NMF_EINVAL, "nmf_composite_fwd: b < 0");
    if (b == 0) return NMF_OK;
    NMF_REQUIRE(sigma && dist && offsets && weight, NMF_EINVAL, "nmf_composite_fwd: null");
    if (b <= WPR_MAX_RAYS)
        NMF_LAUNCH(k_composite_fwd_wave<64>, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, (hipStream_t)stream, sigma,
                           dist, offsets, b, distance_scale, weight, acc);
    else if (group_width(8) == 16)
        NMF_LAUNCH(k_composite_fwd_wave<16>, dim3((unsigned)cdiv(b, 16)), dim3(256), 0, (hipStream_t)stream, sigma,
                           dist, offsets, b, distance_scale, weight, acc);
    else
        NMF_LAUNCH(k_composite_fwd_wave<8>, dim3((unsigned)cdiv(b, 32)), dim3(256), 0, (hipStream_t)stream, sigma,
                           dist, offsets, b, distance_scale, weight, acc);
    NMF_CHECK_LAUNCH("nmf_composite_fwd");
    return NMF_OK;
}

extern "C" int nmf_composite_bwd(const float* sigma, const float* dist, const float* weight, const int64_t* offsets,
                                 int64_t b, float distance_scale, const float* d_weight, float* d_sigma,
                                 void* stream) {
    NMF_REQUIRE(b >= 0, NMF_EINVAL, "nmf_composite_bwd: b < 0");
    if (b == 0) return NMF_OK;
    NMF_REQUIRE(sigma && dist && weight && offsets && d_weight && d_sigma, NMF_EINVAL, "nmf_composite_bwd: null");
    // NMF_COMPOSITE_ONE_CHUNK=0 (tests): every ray through the chunk loops -- the one-chunk path must give the same bits
    static const int one_chunk = !(getenv("NMF_COMPOSITE_ONE_CHUNK") && atoi(getenv("NMF_COMPOSITE_ONE_CHUNK")) == 0);
    if (b <= WPR_MAX_RAYS)
        NMF_LAUNCH(k_composite_bwd_wave<64>, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, (hipStream_t)stream, sigma,
                           dist, weight, offsets, b, distance_scale, d_weight, d_sigma, one_chunk);
    else if (group_width(16) == 16)
        NMF_LAUNCH(k_composite_bwd_wave<16>, dim3((unsigned)cdiv(b, 16)), dim3(256), 0, (hipStream_t)stream, sigma,
                           dist, weight, offsets, b, distance_scale, d_weight, d_sigma, one_chunk);
    else
        NMF_LAUNCH(k_composite_bwd_wave<8>, dim3((unsigned)cdiv(b, 32)), dim3(256), 0, (hipStream_t)stream, sigma,
                           dist, weight, offsets, b, distance_scale, d_weight, d_sigma, one_chunk);
    NMF_CHECK_LAUNCH("nmf_composite_bwd");
    return NMF_OK;
}

extern "C" int nmf_segment_sum(const float* vals, const float* scale, const int64_t* offsets, int64_t n_seg,
                               int32_t D, int32_t lanes, float* out, void* stream) {
    NMF_REQUIRE(n_seg >= 0, NMF_EINVAL, "nmf_segment_sum: n_seg < 0");
    if (n_seg == 0) return NMF_OK;
    NMF_REQUIRE(vals && offsets && out, NMF_EINVAL, "nmf_segment_sum: null");
    NMF_REQUIRE(lanes == 1 || lanes == 8, NMF_ERANGE, "nmf_segment_sum: lanes must be 1 (index order) or 8");
    hipStream_t st = (hipStream_t)stream;
    if (lanes == 8) {
        dim3 grid((unsigned)cdiv(n_seg, 32)), block(256);
        switch (D) {
            case 1: NMF_LAUNCH((k_segment_sum_group<1, 8>), grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
            case 2: NMF_LAUNCH((k_segment_sum_group<2, 8>), grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
            case 3: NMF_LAUNCH((k_segment_sum_group<3, 8>), grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
            case 4: NMF_LAUNCH((k_segment_sum_group<4, 8>), grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
            default: return nmf_fail(NMF_ERANGE, "nmf_segment_sum: D must be 1..4");
        }
        NMF_CHECK_LAUNCH("nmf_segment_sum");
        return NMF_OK;
    }
    dim3 grid((unsigned)cdiv(n_seg, 256)), block(256);
    switch (D) {
        case 1: NMF_LAUNCH(k_segment_sum<1>, grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
        case 2: NMF_LAUNCH(k_segment_sum<2>, grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
        case 3: NMF_LAUNCH(k_segment_sum<3>, grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
        case 4: NMF_LAUNCH(k_segment_sum<4>, grid, block, 0, st, vals, scale, offsets, n_seg, out); break;
        default: return nmf_fail(NMF_ERANGE, "nmf_segment_sum: D must be 1..4");
    }
    NMF_CHECK_LAUNCH("nmf_segment_sum");
    return NMF_OK;
}
