// Per-sample / per-ray glue of the shading step for gfx950 -- the spans of models/microfacet.py and
// modules/tensor_nerf.py that sit between the big operators, each fused into one streaming pass:
//
//   nmf_bounce_index      which samples spawn secondary rays: compact row list, ray offsets per row, inverse map
//                         (models/microfacet.py:333-350: ray_mask / bounce_mask / torch.where bookkeeping)
//   nmf_bounce_prep_*     everything the secondary-ray kernels need per bounce row, gathered in one pass: view
//                         vector, facing normal (:356), clipped roughness (:361), f0, diffuse = albedo * SH irradiance
//                         (:304-316, modules/sh.py:97-142), noised appearance feature (:297), position
//   nmf_ray_compose_*     weights x radiance -> pixel: acc / rgb segment sums (modules/tensor_nerf.py:448-452), the
//                         orientation loss term (:583-587), sRGB tonemap (modules/tonemap.py:34-55) and background
//                         blend (:658-659)
//
// All three are HBM streaming kernels (a few hundred bytes per sample); their point is launch count: the torch
// formulation is ~150 elementwise / index launches per level and as many again in the backward.
#include "common.hpp"
#include "heads_eval.hpp"
#include "rows_bwd.hpp"

namespace {

// ---- bounce index ------------------------------------------------------------------------------------------------
constexpr int IDX_CHUNK = 1024;
constexpr int FLAG_SHIFT = 40;   // packed scan value = count + (count > 0) << 40

__device__ __forceinline__ int64_t wave_scan_i64(int64_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int64_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// inclusive scan over a 1024-thread block; returns the inclusive value, *total = block sum
__device__ __forceinline__ int64_t block_scan_i64(int64_t v, int64_t* ws /*[17]*/, int64_t* total) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int64_t incl = wave_scan_i64(v, lane);
    if (lane == 63) ws[wid] = incl;
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int w = 0; w < IDX_CHUNK / 64; ++w) {
            int64_t t = ws[w];
            ws[w] = run;
            run += t;
        }
        ws[16] = run;
    }
    __syncthreads();
    incl += ws[wid];
    *total = ws[16];
    __syncthreads();
    return incl;
}

// The bounce counts of the samples computed HERE instead of read (R4, nmf_bounce_index_select): modules/pt_selectors.py:5-60 is one
// expression per sample -- k_select_bounces (csrc/select.hip) was a launch of its own on the forward's chain for it; same operations
// (explicitly rounded, no contraction), same bits.  w == NULL: read counts[].
struct SelectArgs {
    const float* w;
    const float* u;
    int mode;
    float mul, add, S;
    const float* S_dev;
};
__device__ __forceinline__ int32_t bounce_count(const SelectArgs& q, const int32_t* __restrict__ counts, int64_t i) {
    if (!q.w) return counts[i];
    const float S = q.S_dev ? *q.S_dev : q.S;
    float pt;
    if (q.mode == 0) {
        pt = fsub(fadd(fmul(q.w[i], q.mul), q.u[i]), 0.5f);
    } else {
        const float wp = fadd(q.w[i], fmul(1e-3f, q.u[i]));
        pt = fadd(fmul(fdiv(wp, S), q.mul), q.add);
    }
    float f = floorf(pt);
    f = fminf(fmaxf(f, 0.f), 400.f);
    return (int32_t)f;
}

__global__ void __launch_bounds__(IDX_CHUNK) k_idx_partial(const int32_t* __restrict__ counts, int64_t M,
                                                          int64_t* __restrict__ chunk_sum, const int64_t* __restrict__ M_live,
                                                          SelectArgs sel) {
    __shared__ int64_t ws[17];
    if (M_live && *M_live < M) M = *M_live;          // (the launch was sized by a bound: nmf_bounce_index_live)
    const int64_t i = (int64_t)blockIdx.x * IDX_CHUNK + threadIdx.x;
    const int32_t c = i < M ? bounce_count(sel, counts, i) : 0;
    const int64_t v = c > 0 ? ((int64_t)c + ((int64_t)1 << FLAG_SHIFT)) : 0;
    int64_t total;
    block_scan_i64(v, ws, &total);
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = total;
}

__global__ void __launch_bounds__(IDX_CHUNK) k_idx_top(int64_t* __restrict__ chunk_sum, int n_chunks,
                                                      int64_t* __restrict__ totals) {
    __shared__ int64_t ws[17];
    int64_t carry = 0;
    for (int base = 0; base < n_chunks; base += IDX_CHUNK) {
        const int i = base + threadIdx.x;
        const int64_t v = i < n_chunks ? chunk_sum[i] : 0;
        int64_t total;
        const int64_t incl = block_scan_i64(v, ws, &total);
        if (i < n_chunks) chunk_sum[i] = carry + incl - v;   // exclusive base of chunk i
        carry += total;
    }
    if (threadIdx.x == 0) {
        totals[0] = carry & (((int64_t)1 << FLAG_SHIFT) - 1);   // R  = secondary rays
        totals[1] = carry >> FLAG_SHIFT;                        // Mb = samples with at least one
    }
}

__global__ void __launch_bounds__(IDX_CHUNK) k_idx_final(const int32_t* __restrict__ counts, int64_t M,
                                                        const int64_t* __restrict__ chunk_base,
                                                        const int64_t* __restrict__ totals, int32_t* __restrict__ bidx,
                                                        int64_t* __restrict__ row_off, int32_t* __restrict__ cnt_rows,
                                                        int32_t* __restrict__ inv, const float4* __restrict__ xyzt,
                                                        float4* __restrict__ xyzt_rows) {
    __shared__ int64_t ws[17];
    const int64_t i = (int64_t)blockIdx.x * IDX_CHUNK + threadIdx.x;
    const int32_t c = i < M ? counts[i] : 0;
    const int64_t v = c > 0 ? ((int64_t)c + ((int64_t)1 << FLAG_SHIFT)) : 0;
    int64_t total;
    const int64_t excl = chunk_base[blockIdx.x] + block_scan_i64(v, ws, &total) - v;
    if (i < M) {
        if (c > 0) {
            const int64_t row = excl >> FLAG_SHIFT;
            bidx[row] = (int32_t)i;
            cnt_rows[row] = c;
            row_off[row] = excl & (((int64_t)1 << FLAG_SHIFT) - 1);
            inv[i] = (int32_t)row;
            if (xyzt_rows) xyzt_rows[row] = xyzt[i];
        } else {
            inv[i] = -1;
        }
    }
    if (i == 0) row_off[totals[1]] = totals[0];
}

// k_idx_top folded into the last pass for up to IDX_CHUNK chunks (1 M samples): every block scans the chunk sums itself
// (one load per thread), which removes a one-block kernel from the forward's dependent chain.
__global__ void __launch_bounds__(IDX_CHUNK) k_idx_final_fused(const int32_t* __restrict__ counts, int64_t M,
                                                              const int64_t* __restrict__ chunk_sum, int n_chunks,
                                                              int64_t* __restrict__ totals, int32_t* __restrict__ bidx,
                                                              int64_t* __restrict__ row_off, int32_t* __restrict__ cnt_rows,
                                                              int32_t* __restrict__ inv, const float4* __restrict__ xyzt,
                                                              float4* __restrict__ xyzt_rows, int64_t* pub, int64_t pub_seq,
                                                              const int64_t* __restrict__ M_live, SelectArgs sel) {
    __shared__ int64_t ws[17];
    __shared__ int64_t base_s;
    if (M_live && *M_live < M) M = *M_live;
    const int tid = threadIdx.x;
    const int64_t cs = tid < n_chunks ? chunk_sum[tid] : 0;
    int64_t all;
    const int64_t cincl = block_scan_i64(cs, ws, &all);
    if (tid == (int)blockIdx.x) base_s = cincl - cs;
    __syncthreads();
    const int64_t R = all & (((int64_t)1 << FLAG_SHIFT) - 1), Mb = all >> FLAG_SHIFT;
    const int64_t i = (int64_t)blockIdx.x * IDX_CHUNK + tid;
    const int32_t c = i < M ? bounce_count(sel, counts, i) : 0;
    const int64_t v = c > 0 ? ((int64_t)c + ((int64_t)1 << FLAG_SHIFT)) : 0;
    int64_t total;
    const int64_t excl = base_s + block_scan_i64(v, ws, &total) - v;
    if (i < M) {
        if (c > 0) {
            const int64_t row = excl >> FLAG_SHIFT;
            bidx[row] = (int32_t)i;
            cnt_rows[row] = c;
            row_off[row] = excl & (((int64_t)1 << FLAG_SHIFT) - 1);
            inv[i] = (int32_t)row;
            if (xyzt_rows) xyzt_rows[row] = xyzt[i];
        } else {
            inv[i] = -1;
        }
    }
    if (i == 0) {
        row_off[Mb] = R;
        totals[0] = R;
        totals[1] = Mb;
        publish_sizes(pub, R, Mb, pub_seq);
    }
}

// ---- bounce prep -------------------------------------------------------------------------------------------------
using nmf_rows::HEADS;
using nmf_rows::FEAT;
using nmf_rows::Conv;
using nmf_rows::sh9;
using nmf_rows::sgn;
using nmf_rows::irradiance_E;

struct HeadsIn {            // W == NULL: the heads are read from `heads`
    const float* W;
    const float* b;
    nmf_heads::HeadP hp;
    float* out;
};
__global__ void __launch_bounds__(256) k_bounce_prep_fwd(
    const int32_t* __restrict__ bidx, int64_t Mb, const float* __restrict__ normals, const float* __restrict__ app,
    const float* __restrict__ heads, const float4* __restrict__ xyzt, const int32_t* __restrict__ ray_id,
    const float* __restrict__ rays, Conv conv, const float* __restrict__ feat_noise, float anoise, float min_rough,
    int row_inputs, float* __restrict__ V, float* __restrict__ N, float* __restrict__ r1, float* __restrict__ f0,
    float* __restrict__ diffuse, float* __restrict__ feat, float* __restrict__ xyz, HeadsIn hin) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= Mb) return;
    // every load first (three dependent round trips: bidx -> ray_id and the row's inputs -> the ray), the stores behind them:
    // interleaved as the values are produced, the launch was a chain of seven
    const int64_t m = bidx[row];
    const int64_t ia = row_inputs ? row : m;       // app / heads / noise given per bounce row or per sample
    const int64_t in = row_inputs == 2 ? row : m;  // normals per bounce row as well (row_inputs 2)
    const int64_t rid = ray_id[m];
    const float nx = normals[in * 3], ny = normals[in * 3 + 1], nz = normals[in * 3 + 2];
    float h0, h1, h2, h6, h7, h8, h9;
    if (!hin.W) {
        const float* h = heads + ia * HEADS;
        h0 = h[0]; h1 = h[1]; h2 = h[2]; h6 = h[6]; h7 = h[7]; h8 = h[8]; h9 = h[9];
    }
    const float4 p = xyzt[m];
    const float4* a4 = reinterpret_cast<const float4*>(app + ia * FEAT);
    const float4* n4 = feat_noise ? reinterpret_cast<const float4*>(feat_noise + ia * FEAT) : nullptr;
    float4 av[FEAT / 4], zv[FEAT / 4];
#pragma unroll
    for (int i = 0; i < FEAT / 4; ++i) {
        av[i] = a4[i];
        zv[i] = n4 ? n4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (hin.W) {
        // the material heads of this row evaluated HERE (nmf_bounce_prep_fwd_heads: rows given per bounce row) and written out for
        // the backward -- k_heads_fwd was a launch of its own between the appearance query and this one; nmf_heads::heads_eval is
        // the code both run, so the bits are the same
        float fa[nmf_heads::F], ho[nmf_heads::O];
#pragma unroll
        for (int i = 0; i < FEAT / 4; ++i) { fa[4 * i] = av[i].x; fa[4 * i + 1] = av[i].y; fa[4 * i + 2] = av[i].z; fa[4 * i + 3] = av[i].w; }
        nmf_heads::heads_eval(fa, hin.W, hin.b, hin.hp, ho);
#pragma unroll
        for (int j = 0; j < nmf_heads::O; ++j) hin.out[ia * HEADS + j] = ho[j];
        h0 = ho[0]; h1 = ho[1]; h2 = ho[2]; h6 = ho[6]; h7 = ho[7]; h8 = ho[8]; h9 = ho[9];
    }
    const float* d = rays + rid * 6 + 3;
    const float vx = -d[0], vy = -d[1], vz = -d[2];
    const float s = sgn(vx * nx + vy * ny + vz * nz);                       // models/microfacet.py:356
    V[row * 3] = vx; V[row * 3 + 1] = vy; V[row * 3 + 2] = vz;
    N[row * 3] = nx * s; N[row * 3 + 1] = ny * s; N[row * 3 + 2] = nz * s;
    r1[row] = fmaxf(h9, min_rough);
    float Y[9];
    sh9(nx, ny, nz, Y);
    const float hc[3] = {h0, h1, h2}, hf[3] = {h6, h7, h8};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float E = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) E += conv.c[k * 3 + c] * Y[k];
        diffuse[row * 3 + c] = hc[c] * E;
        f0[row * 3 + c] = hf[c];
    }
    xyz[row * 3] = p.x; xyz[row * 3 + 1] = p.y; xyz[row * 3 + 2] = p.z;
    float4* o4 = reinterpret_cast<float4*>(feat + row * FEAT);
#pragma unroll
    for (int i = 0; i < FEAT / 4; ++i) {
        float4 a = av[i];
        if (n4) {
            const float4 z = zv[i];
            a.x += z.x * anoise; a.y += z.y * anoise; a.z += z.z * anoise; a.w += z.w * anoise;
        }
        o4[i] = a;
    }
}

// one thread per SAMPLE: rows scatter back through the inverse map, everything else is written as zero, so the
// three gradient tensors need no separate fill.  With row_inputs the head / feature adjoints stay per bounce row
// ([Mb][11], [Mb][24], written by the first Mb threads) and only d_normals covers all samples.
__global__ void __launch_bounds__(256) k_bounce_prep_bwd(
    const int32_t* __restrict__ inv, int64_t M, const int32_t* __restrict__ bidx, int64_t Mb,
    const float* __restrict__ normals, const float* __restrict__ heads,
    const int32_t* __restrict__ ray_id, const float* __restrict__ rays, Conv conv, float min_rough, int detach_n,
    int row_inputs, const float* __restrict__ dN, const float* __restrict__ dr1, const float* __restrict__ df0,
    const float* __restrict__ ddiff, int sN, int sr, int sf, int sd, const float* __restrict__ dfeat,
    float* __restrict__ d_normals, float* __restrict__ d_heads, float* __restrict__ d_app) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row_inputs == 2 && heads && d_heads && d_app) {
        // Everything per bounce row (the training pass with sparse normals): the same arithmetic as the general form below with
        // every load in front of the stores -- bidx -> ray_id -> ray is the only chain (three round trips; the general form,
        // which interleaves loads, stores and mode tests, waits eleven times)
        if (t >= Mb) return;
        nmf_rows::RowsBwdIn rin{bidx, normals, heads, ray_id, rays, conv, min_rough, detach_n, dN, dr1, df0, ddiff, sN, sr, sf, sd, dfeat,
                                d_normals};
        float gh[HEADS];
        float4 gf[FEAT / 4];
        nmf_rows::prep_bwd_row(rin, t, gh, gf);
        float4* o4 = reinterpret_cast<float4*>(d_app + t * FEAT);
#pragma unroll
        for (int i = 0; i < FEAT / 4; ++i) o4[i] = gf[i];
#pragma unroll
        for (int j = 0; j < HEADS; ++j) d_heads[t * HEADS + j] = gh[j];
        return;
    }
    if (row_inputs == 2) {          // normals and their adjoint per bounce row: [Mb][3], nothing sample-sized is touched
        if (t < Mb) {
            float gn[3] = {0.f, 0.f, 0.f};
            if (!detach_n && dN) {
                const float nx = normals[t * 3], ny = normals[t * 3 + 1], nz = normals[t * 3 + 2];
                const float* d = rays + (int64_t)ray_id[bidx[t]] * 6 + 3;
                const float s = sgn(-(d[0] * nx + d[1] * ny + d[2] * nz));
                gn[0] = dN[t * sN] * s; gn[1] = dN[t * sN + 1] * s; gn[2] = dN[t * sN + 2] * s;
            }
            d_normals[t * 3] = gn[0]; d_normals[t * 3 + 1] = gn[1]; d_normals[t * 3 + 2] = gn[2];
        }
    } else if (t < M) {          // normal adjoint of sample t
        const int64_t row = inv[t];
        float gn[3] = {0.f, 0.f, 0.f};
        if (row >= 0 && !detach_n && dN) {
            const float nx = normals[t * 3], ny = normals[t * 3 + 1], nz = normals[t * 3 + 2];
            const float* d = rays + (int64_t)ray_id[t] * 6 + 3;
            const float s = sgn(-(d[0] * nx + d[1] * ny + d[2] * nz));
            gn[0] = dN[row * sN] * s; gn[1] = dN[row * sN + 1] * s; gn[2] = dN[row * sN + 2] * s;
        }
        d_normals[t * 3] = gn[0]; d_normals[t * 3 + 1] = gn[1]; d_normals[t * 3 + 2] = gn[2];
    }
    // head / feature adjoints: slot t of the output belongs to sample t (dense) or to bounce row t (row_inputs)
    const int64_t n_out = row_inputs ? Mb : M;
    if (t >= n_out) return;
    const int64_t row = row_inputs ? t : inv[t];
    const int64_t m = row_inputs ? (int64_t)bidx[t] : t;
    float gh[HEADS];
#pragma unroll
    for (int j = 0; j < HEADS; ++j) gh[j] = 0.f;
    float4* o4 = reinterpret_cast<float4*>(d_app + t * FEAT);
    if (row >= 0) {
        const int64_t in = row_inputs == 2 ? t : m;
        const float nx = normals[in * 3], ny = normals[in * 3 + 1], nz = normals[in * 3 + 2];
        const float* h = heads + t * HEADS;
        float E3[3];
        irradiance_E(nx, ny, nz, conv, E3);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gh[c] = ddiff ? ddiff[row * sd + c] * E3[c] : 0.f;
            gh[6 + c] = df0 ? df0[row * sf + c] : 0.f;
        }
        gh[9] = (dr1 && h[9] >= min_rough) ? dr1[row * sr] : 0.f;
        const float4* g4 = dfeat ? reinterpret_cast<const float4*>(dfeat + row * FEAT) : nullptr;
#pragma unroll
        for (int i = 0; i < FEAT / 4; ++i) o4[i] = g4 ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (int i = 0; i < FEAT / 4; ++i) o4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < HEADS; ++j) d_heads[t * HEADS + j] = gh[j];
}

// ---- ray compose -------------------------------------------------------------------------------------------------
constexpr float SRGB_LIMIT = 0.0031308f;

__device__ __forceinline__ float srgb(float x, int noclip) {
    // modules/tonemap.py:34-55: where(x > limit, 1.055 * clip(x, min=limit)^(1/2.4) - 0.055, 12.92 x), clip(0, 1)
    float o = x > SRGB_LIMIT ? 1.055f * powf(fmaxf(x, SRGB_LIMIT), 1.0f / 2.4f) - 0.055f : 12.92f * x;
    return noclip ? o : fminf(fmaxf(o, 0.f), 1.f);
}

__device__ __forceinline__ float srgb_grad(float x, int noclip) {
    float o, g;
    if (x > SRGB_LIMIT) {
        const float p = powf(x, 1.0f / 2.4f);
        o = 1.055f * p - 0.055f;
        g = 1.055f * (1.0f / 2.4f) * p / x;
    } else {
        o = 12.92f * x;
        g = 12.92f;
    }
    if (!noclip && (o < 0.f || o > 1.f)) g = 0.f;
    return g;
}

template <int W>      // lanes per ray: 64 for primary-ray batches, 8 for the re-traced rays (short segments, many rays)
__global__ void __launch_bounds__(256) k_ray_compose_fwd_wave(
    const float* __restrict__ weight, const float* __restrict__ refl_rows, const int32_t* __restrict__ inv,
    const float* __restrict__ normals, const float* __restrict__ rays, const int64_t* __restrict__ offsets, int64_t B,
    const float* __restrict__ bg, int bg_per_ray, int tonemap, int noclip, float* __restrict__ rgb_map,
    float* __restrict__ acc_out, float* __restrict__ rgb_lin, float* __restrict__ ori_out) {
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / W;
    const int lane = threadIdx.x & (W - 1);
    const bool ok = r < B;                       // whole lane groups are in or out; the shuffles below stay inside a group
    const int64_t s = ok ? offsets[r] : 0, e = ok ? offsets[r + 1] : 0;
    const int64_t rq = ok ? r : 0;
    const float dx = rays[rq * 6 + 3], dy = rays[rq * 6 + 4], dz = rays[rq * 6 + 5];
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};      // acc, c0, c1, c2, ori
    // two samples of a lane per pass, their loads issued together, the row lookups (inv -> refl_rows) unconditional on a clamped
    // row (the sums keep their order and their bits): a pass is two dependent memory round trips, and a wave is as slow as its
    // longest ray (R4: 9 % of the re-traced rays
    // keep more than 8 samples, 4.5 % more than 16, the longest 200)
    const bool rows = inv && refl_rows;
    for (int64_t k = s + lane; k < e; k += 2 * W) {
        const int64_t k1 = k + W;
        const bool two = k1 < e;
        const int64_t kb = two ? k1 : k;
        const float w0 = weight[k], w1 = two ? weight[kb] : 0.f;
        int64_t row0 = -1, row1 = -1;
        if (rows) { row0 = inv[k]; row1 = inv[kb]; }
        float n0[3] = {0.f, 0.f, 0.f}, n1[3] = {0.f, 0.f, 0.f};
        if (ori_out) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { n0[c] = normals[k * 3 + c]; n1[c] = normals[kb * 3 + c]; }
        }
        float c0[3] = {0.f, 0.f, 0.f}, c1[3] = {0.f, 0.f, 0.f};
        if (rows) {
            const int64_t q0 = row0 >= 0 ? row0 : 0, q1 = row1 >= 0 ? row1 : 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) { c0[c] = refl_rows[q0 * 3 + c]; c1[c] = refl_rows[q1 * 3 + c]; }
        }
        v[0] += w0;
        if (row0 >= 0) { v[1] += w0 * c0[0]; v[2] += w0 * c0[1]; v[3] += w0 * c0[2]; }
        if (ori_out) {
            const float ndv = fminf(-(dx * n0[0] + dy * n0[1] + dz * n0[2]), 0.f);
            v[4] += w0 * (ndv * ndv);
        }
        if (two) {
            v[0] += w1;
            if (row1 >= 0) { v[1] += w1 * c1[0]; v[2] += w1 * c1[1]; v[3] += w1 * c1[2]; }
            if (ori_out) {
                const float ndv = fminf(-(dx * n1[0] + dy * n1[1] + dz * n1[2]), 0.f);
                v[4] += w1 * (ndv * ndv);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q)
        for (int d = W / 2; d > 0; d >>= 1) v[q] += __shfl_down(v[q], d, W);
    if (lane != 0 || !ok) return;
    rgb_lin[r * 3] = v[1]; rgb_lin[r * 3 + 1] = v[2]; rgb_lin[r * 3 + 2] = v[3];
    acc_out[r] = v[0];
    if (ori_out) ori_out[r] = v[4];
    const float* b = bg + (bg_per_ray ? r * 3 : 0);
    const float t = 1.f - v[0];
    rgb_map[r * 3] = (tonemap ? srgb(v[1], noclip) : v[1]) + t * b[0];
    rgb_map[r * 3 + 1] = (tonemap ? srgb(v[2], noclip) : v[2]) + t * b[1];
    rgb_map[r * 3 + 2] = (tonemap ? srgb(v[3], noclip) : v[3]) + t * b[2];
}

// one thread per sample
__global__ void __launch_bounds__(256) k_ray_compose_bwd(
    const float* __restrict__ weight, const float* __restrict__ refl_rows, const int32_t* __restrict__ inv,
    const float* __restrict__ normals, const float* __restrict__ rays, const int32_t* __restrict__ ray_id, int64_t M,
    const float* __restrict__ bg, int bg_per_ray, int tonemap, int noclip, const float* __restrict__ rgb_lin,
    const float* __restrict__ d_rgb_map, const float* __restrict__ d_acc, const float* __restrict__ d_ori,
    float* __restrict__ d_weight, float* __restrict__ d_refl, float* __restrict__ d_normals) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int64_t r = ray_id[m];
    const float w = weight[m];
    float g[3] = {0.f, 0.f, 0.f};
    float ga = d_acc ? d_acc[r] : 0.f;
    if (d_rgb_map) {
        const float* b = bg + (bg_per_ray ? r * 3 : 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float dm = d_rgb_map[r * 3 + c];
            g[c] = tonemap ? dm * srgb_grad(rgb_lin[r * 3 + c], noclip) : dm;
            ga -= dm * b[c];
        }
    }
    float dw = ga;
    const int64_t row = (inv && refl_rows) ? inv[m] : -1;
    if (row >= 0) {
        dw += g[0] * refl_rows[row * 3] + g[1] * refl_rows[row * 3 + 1] + g[2] * refl_rows[row * 3 + 2];
        d_refl[row * 3] = w * g[0]; d_refl[row * 3 + 1] = w * g[1]; d_refl[row * 3 + 2] = w * g[2];
    }
    float gn[3] = {0.f, 0.f, 0.f};
    if (d_ori) {
        const float dx = rays[r * 6 + 3], dy = rays[r * 6 + 4], dz = rays[r * 6 + 5];
        const float ndv = -(dx * normals[m * 3] + dy * normals[m * 3 + 1] + dz * normals[m * 3 + 2]);
        if (ndv < 0.f) {
            const float go = d_ori[r];
            dw += go * ndv * ndv;
            const float k = go * w * 2.f * ndv;
            gn[0] = -k * dx; gn[1] = -k * dy; gn[2] = -k * dz;
        }
    }
    d_weight[m] = dw;
    if (d_normals) {
        d_normals[m * 3] = gn[0]; d_normals[m * 3 + 1] = gn[1]; d_normals[m * 3 + 2] = gn[2];
    }
}

// d_bg[r][c] = (1 - acc[r]) * d_rgb[r][c]: the background adjoint nmf_ray_compose_bwd leaves to the caller (three torch
// launches as `(1 - acc).unsqueeze(1) * d_rgb`, the same two roundings)
__global__ void __launch_bounds__(256) k_bg_adjoint(const float* __restrict__ acc, const float* __restrict__ d_rgb, int64_t n,
                                                    float* __restrict__ d_bg) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) d_bg[k] = (1.f - acc[k / 3]) * d_rgb[k];
}

}  // namespace

extern "C" int64_t nmf_bounce_index_workspace_bytes(int64_t M) { return (cdiv(M > 0 ? M : 1, IDX_CHUNK) + 1) * 8; }

extern "C" int nmf_bounce_index(const int32_t* counts, int64_t M, int32_t* bidx, int64_t* row_off,
                                int32_t* cnt_rows, int32_t* inv, int64_t* totals, const float* xyzt, float* xyzt_rows,
                                void* workspace, int64_t workspace_bytes, void* stream) {
    return nmf_bounce_index_live(counts, M, nullptr, bidx, row_off, cnt_rows, inv, totals, xyzt, xyzt_rows, workspace, workspace_bytes,
                                 nullptr, 0, stream);
}

extern "C" int nmf_bounce_index_publish(const int32_t* counts, int64_t M, int32_t* bidx, int64_t* row_off,
                                        int32_t* cnt_rows, int32_t* inv, int64_t* totals, const float* xyzt, float* xyzt_rows,
                                        void* workspace, int64_t workspace_bytes, void* publish_mapped_dev,
                                        int64_t publish_seq, void* stream) {
    return nmf_bounce_index_live(counts, M, nullptr, bidx, row_off, cnt_rows, inv, totals, xyzt, xyzt_rows, workspace, workspace_bytes,
                                 publish_mapped_dev, publish_seq, stream);
}

static int bounce_index_impl(const int32_t* counts, SelectArgs sel, int64_t M, const int64_t* M_live, int32_t* bidx, int64_t* row_off,
                             int32_t* cnt_rows, int32_t* inv, int64_t* totals, const float* xyzt, float* xyzt_rows, void* workspace,
                             int64_t workspace_bytes, void* publish_mapped_dev, int64_t publish_seq, void* stream);
extern "C" int nmf_bounce_index_live(const int32_t* counts, int64_t M, const int64_t* M_live, int32_t* bidx, int64_t* row_off,
                                     int32_t* cnt_rows, int32_t* inv, int64_t* totals, const float* xyzt, float* xyzt_rows,
                                     void* workspace, int64_t workspace_bytes, void* publish_mapped_dev,
                                     int64_t publish_seq, void* stream) {
    NMF_REQUIRE(counts || M == 0, NMF_EINVAL, "nmf_bounce_index: null");
    return bounce_index_impl(counts, SelectArgs{nullptr, nullptr, 0, 0.f, 0.f, 0.f, nullptr}, M, M_live, bidx, row_off, cnt_rows, inv, totals,
                             xyzt, xyzt_rows, workspace, workspace_bytes, publish_mapped_dev, publish_seq, stream);
}
extern "C" int nmf_bounce_index_select(const float* weights, const float* u, int32_t mode, float mul, float add, float sum_w,
                                       const float* sum_w_dev, int64_t M, const int64_t* M_live, int32_t* bidx, int64_t* row_off,
                                       int32_t* cnt_rows, int32_t* inv, int64_t* totals, const float* xyzt, float* xyzt_rows,
                                       void* workspace, int64_t workspace_bytes, void* publish_mapped_dev, int64_t publish_seq,
                                       void* stream) {
    NMF_REQUIRE((weights && u) || M == 0, NMF_EINVAL, "nmf_bounce_index_select: null");
    NMF_REQUIRE(mode == 0 || mode == 1, NMF_EINVAL, "nmf_bounce_index_select: mode");
    NMF_REQUIRE(M <= (int64_t)IDX_CHUNK * IDX_CHUNK, NMF_ERANGE, "nmf_bounce_index_select: more than 2^20 samples (use nmf_select_bounces + nmf_bounce_index)");
    return bounce_index_impl(nullptr, SelectArgs{weights, u, (int)mode, mul, add, sum_w, sum_w_dev}, M, M_live, bidx, row_off, cnt_rows, inv,
                             totals, xyzt, xyzt_rows, workspace, workspace_bytes, publish_mapped_dev, publish_seq, stream);
}
static int bounce_index_impl(const int32_t* counts, SelectArgs sel, int64_t M, const int64_t* M_live, int32_t* bidx, int64_t* row_off,
                             int32_t* cnt_rows, int32_t* inv, int64_t* totals, const float* xyzt, float* xyzt_rows, void* workspace,
                             int64_t workspace_bytes, void* publish_mapped_dev, int64_t publish_seq, void* stream) {
    int64_t* pub = static_cast<int64_t*>(publish_mapped_dev);
    NMF_REQUIRE(M >= 0, NMF_EINVAL, "nmf_bounce_index: M < 0");
    NMF_REQUIRE(totals && row_off, NMF_EINVAL, "nmf_bounce_index: null");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        hipError_t e = hipMemsetAsync(totals, 0, 16, st);
        if (e == hipSuccess) e = hipMemsetAsync(row_off, 0, 8, st);
        if (e != hipSuccess) return nmf_fail((int)e, "nmf_bounce_index: memset");
        if (pub) return nmf_publish_i64x2(totals, pub, publish_seq, stream);
        return NMF_OK;
    }
    NMF_REQUIRE((counts || sel.w) && bidx && cnt_rows && inv && workspace, NMF_EINVAL, "nmf_bounce_index: null");
    NMF_REQUIRE(!xyzt_rows || xyzt, NMF_EINVAL, "nmf_bounce_index: xyzt_rows needs xyzt");
    const float4* x4 = reinterpret_cast<const float4*>(xyzt);
    float4* r4 = reinterpret_cast<float4*>(xyzt_rows);
    const int64_t n_chunks = cdiv(M, IDX_CHUNK);
    NMF_REQUIRE(workspace_bytes >= nmf_bounce_index_workspace_bytes(M), NMF_EINVAL, "nmf_bounce_index: workspace too small");
    NMF_REQUIRE(n_chunks <= (1 << 22), NMF_ERANGE, "nmf_bounce_index: M too large");
    int64_t* chunk = static_cast<int64_t*>(workspace);
    NMF_REQUIRE(!M_live || n_chunks <= IDX_CHUNK, NMF_ERANGE, "nmf_bounce_index_live: bound too large for a device-side count");
    NMF_LAUNCH(k_idx_partial, dim3((unsigned)n_chunks), dim3(IDX_CHUNK), 0, st, counts, M, chunk, M_live, sel);
    if (n_chunks <= IDX_CHUNK) {
        NMF_LAUNCH(k_idx_final_fused, dim3((unsigned)n_chunks), dim3(IDX_CHUNK), 0, st, counts, M, chunk,
                           (int)n_chunks, totals, bidx, row_off, cnt_rows, inv, x4, r4, pub, publish_seq, M_live, sel);
    } else {
        NMF_LAUNCH(k_idx_top, dim3(1), dim3(IDX_CHUNK), 0, st, chunk, (int)n_chunks, totals);
        NMF_LAUNCH(k_idx_final, dim3((unsigned)n_chunks), dim3(IDX_CHUNK), 0, st, counts, M, chunk, totals, bidx,
                           row_off, cnt_rows, inv, x4, r4);
    }
    NMF_CHECK_LAUNCH("nmf_bounce_index");
    if (pub && n_chunks > IDX_CHUNK) return nmf_publish_i64x2(totals, pub, publish_seq, stream);
    return NMF_OK;
}

static Conv load_conv(const float* conv) {
    Conv c;
    c.c = conv;
    return c;
}

extern "C" int nmf_bounce_prep_fwd(const int32_t* bidx, int64_t Mb, const float* normals, const float* app,
                                   const float* heads, const float* xyzt, const int32_t* ray_id, const float* rays,
                                   const float* conv, const float* feat_noise, float anoise, float min_rough,
                                   int32_t row_inputs, float* V, float* N, float* r1, float* f0, float* diffuse,
                                   float* feat, float* xyz, void* stream) {
    NMF_REQUIRE(Mb >= 0, NMF_EINVAL, "nmf_bounce_prep_fwd: Mb < 0");
    if (Mb == 0) return NMF_OK;
    NMF_REQUIRE(bidx && normals && app && heads && xyzt && ray_id && rays && conv && V && N && r1 && f0 &&
                    diffuse && feat && xyz, NMF_EINVAL, "nmf_bounce_prep_fwd: null");
    NMF_LAUNCH(k_bounce_prep_fwd, dim3((unsigned)cdiv(Mb, 256)), dim3(256), 0, (hipStream_t)stream, bidx, Mb,
                       normals, app, heads, reinterpret_cast<const float4*>(xyzt), ray_id, rays, load_conv(conv),
                       feat_noise, anoise, min_rough, (int)row_inputs, V, N, r1, f0, diffuse, feat, xyz,
                       HeadsIn{nullptr, nullptr, nmf_heads::HeadP{0.f, 0.f, 0.f, 0.f, 0.f}, nullptr});
    NMF_CHECK_LAUNCH("nmf_bounce_prep_fwd");
    return NMF_OK;
}

extern "C" int nmf_bounce_prep_fwd_heads(const int32_t* bidx, int64_t Mb, const float* normals, const float* app, const float* head_W,
                                         const float* head_b, float diffuse_mul, float diffuse_bias, float tint_bias, float f0_bias,
                                         float rough_bias, const float* xyzt, const int32_t* ray_id, const float* rays,
                                         const float* conv, const float* feat_noise, float anoise, float min_rough,
                                         int32_t row_inputs, float* heads_out, float* V, float* N, float* r1, float* f0,
                                         float* diffuse, float* feat, float* xyz, void* stream) {
    NMF_REQUIRE(Mb >= 0, NMF_EINVAL, "nmf_bounce_prep_fwd_heads: Mb < 0");
    if (Mb == 0) return NMF_OK;
    NMF_REQUIRE(row_inputs == 1 || row_inputs == 2, NMF_EINVAL, "nmf_bounce_prep_fwd_heads: app must be given per bounce row");
    NMF_REQUIRE(bidx && normals && app && head_W && head_b && heads_out && xyzt && ray_id && rays && conv && V && N && r1 && f0 &&
                    diffuse && feat && xyz, NMF_EINVAL, "nmf_bounce_prep_fwd_heads: null");
    NMF_LAUNCH(k_bounce_prep_fwd, dim3((unsigned)cdiv(Mb, 256)), dim3(256), 0, (hipStream_t)stream, bidx, Mb,
                       normals, app, heads_out, reinterpret_cast<const float4*>(xyzt), ray_id, rays, load_conv(conv),
                       feat_noise, anoise, min_rough, (int)row_inputs, V, N, r1, f0, diffuse, feat, xyz,
                       HeadsIn{head_W, head_b, nmf_heads::HeadP{diffuse_mul, diffuse_bias, tint_bias, f0_bias, rough_bias}, heads_out});
    NMF_CHECK_LAUNCH("nmf_bounce_prep_fwd_heads");
    return NMF_OK;
}

extern "C" int nmf_bounce_prep_bwd(const int32_t* inv, int64_t M, const int32_t* bidx, int64_t Mb, const float* normals,
                                   const float* heads, const int32_t* ray_id, const float* rays, const float* conv,
                                   float min_rough, int32_t detach_normals, int32_t row_inputs, const float* dN,
                                   const float* dr1, const float* df0,
                                   const float* ddiffuse, const int32_t row_strides[4], const float* dfeat,
                                   float* d_normals, float* d_heads, float* d_app, void* stream) {
    const int sN = row_strides ? row_strides[0] : 3, sr = row_strides ? row_strides[1] : 1;
    const int sf = row_strides ? row_strides[2] : 3, sd = row_strides ? row_strides[3] : 3;
    NMF_REQUIRE(M >= 0, NMF_EINVAL, "nmf_bounce_prep_bwd: M < 0");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE((inv || row_inputs == 2) && normals && ray_id && rays && conv && d_normals, NMF_EINVAL,
                "nmf_bounce_prep_bwd: null");
    NMF_REQUIRE(Mb >= 0 && Mb <= M, NMF_EINVAL, "nmf_bounce_prep_bwd: Mb outside [0, M]");
    NMF_REQUIRE((row_inputs ? Mb == 0 : false) || (heads && d_heads && d_app), NMF_EINVAL, "nmf_bounce_prep_bwd: null");
    NMF_REQUIRE(!row_inputs || Mb == 0 || bidx, NMF_EINVAL, "nmf_bounce_prep_bwd: row_inputs needs bidx");
    const int64_t n_threads = row_inputs == 2 ? (Mb > 0 ? Mb : 1) : M;
    NMF_LAUNCH(k_bounce_prep_bwd, dim3((unsigned)cdiv(n_threads, 256)), dim3(256), 0, (hipStream_t)stream, inv, M, bidx,
                       Mb, normals, heads, ray_id, rays, load_conv(conv), min_rough, (int)detach_normals, (int)row_inputs,
                       dN, dr1, df0, ddiffuse, sN, sr, sf, sd, dfeat, d_normals, d_heads, d_app);
    NMF_CHECK_LAUNCH("nmf_bounce_prep_bwd");
    return NMF_OK;
}

extern "C" int nmf_ray_compose_fwd(const float* weight, const float* refl_rows, const int32_t* inv, const float* normals,
                                   const float* rays, const int64_t* offsets, int64_t B, const float* bg,
                                   int32_t bg_per_ray, int32_t tonemap, int32_t noclip, float* rgb_map, float* acc,
                                   float* rgb_lin, float* ori, void* stream) {
    NMF_REQUIRE(B >= 0, NMF_EINVAL, "nmf_ray_compose_fwd: B < 0");
    if (B == 0) return NMF_OK;
    // weight may be NULL when no ray has a sample (every segment empty)
    NMF_REQUIRE(rays && offsets && bg && rgb_map && acc && rgb_lin, NMF_EINVAL, "nmf_ray_compose_fwd: null");
    NMF_REQUIRE(!ori || normals, NMF_EINVAL, "nmf_ray_compose_fwd: ori needs normals");
    if (B <= 16384)      // few rays with long segments (primary rays): one wave per ray
        NMF_LAUNCH(k_ray_compose_fwd_wave<64>, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, weight,
                           refl_rows, inv, normals, rays, offsets, B, bg, (int)bg_per_ray, (int)tonemap, (int)noclip,
                           rgb_map, acc, rgb_lin, ori);
    else                 // many rays with short segments (re-traced rays): eight lanes per ray
        NMF_LAUNCH(k_ray_compose_fwd_wave<8>, dim3((unsigned)cdiv(B, 32)), dim3(256), 0, (hipStream_t)stream, weight,
                           refl_rows, inv, normals, rays, offsets, B, bg, (int)bg_per_ray, (int)tonemap, (int)noclip,
                           rgb_map, acc, rgb_lin, ori);
    NMF_CHECK_LAUNCH("nmf_ray_compose_fwd");
    return NMF_OK;
}

extern "C" int nmf_ray_compose_bwd(const float* weight, const float* refl_rows, const int32_t* inv, const float* normals,
                                   const float* rays, const int32_t* ray_id, int64_t M, const float* bg,
                                   int32_t bg_per_ray, int32_t tonemap, int32_t noclip, const float* rgb_lin,
                                   const float* d_rgb_map, const float* d_acc, const float* d_ori, float* d_weight,
                                   float* d_refl, float* d_normals, void* stream) {
    NMF_REQUIRE(M >= 0, NMF_EINVAL, "nmf_ray_compose_bwd: M < 0");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(weight && rays && ray_id && bg && rgb_lin && d_weight, NMF_EINVAL, "nmf_ray_compose_bwd: null");
    NMF_REQUIRE(!(inv && refl_rows) || d_refl, NMF_EINVAL, "nmf_ray_compose_bwd: d_refl missing");
    NMF_REQUIRE(!d_ori || normals, NMF_EINVAL, "nmf_ray_compose_bwd: d_ori needs normals");
    NMF_LAUNCH(k_ray_compose_bwd, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, weight,
                       refl_rows, inv, normals, rays, ray_id, M, bg, (int)bg_per_ray, (int)tonemap, (int)noclip, rgb_lin,
                       d_rgb_map, d_acc, d_ori, d_weight, d_refl, d_normals);
    NMF_CHECK_LAUNCH("nmf_ray_compose_bwd");
    return NMF_OK;
}

extern "C" int nmf_bg_adjoint(const float* acc, const float* d_rgb_map, int64_t B, float* d_bg, void* stream) {
    NMF_REQUIRE(B >= 0, NMF_EINVAL, "nmf_bg_adjoint: B < 0");
    if (B == 0) return NMF_OK;
    NMF_REQUIRE(acc && d_rgb_map && d_bg, NMF_EINVAL, "nmf_bg_adjoint: null");
    NMF_LAUNCH(k_bg_adjoint, dim3((unsigned)cdiv(3 * B, 256)), dim3(256), 0, (hipStream_t)stream, acc, d_rgb_map, 3 * B,
                       d_bg);
    NMF_CHECK_LAUNCH("nmf_bg_adjoint");
    return NMF_OK;
}
