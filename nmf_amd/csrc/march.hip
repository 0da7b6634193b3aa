// Ray marching, occupancy culling and compaction for gfx950.
// Replaces AlphaGridSampler.sample / sample_ray / AlphaGridMask.sample_alpha
// (reference: samplers/alphagrid.py:131-207, 23-45, 279-370).
//
// Mapping: ONE 64-lane wavefront per ray.  Lane l owns steps k = 64*j + l; the cumulative jitter
// of alphagrid.py:168-173 is a wave-level inclusive scan carried across the j iterations in
// float64 (torch's CPU cumsum accumulates in float64 and rounds each element to fp32 -- SURVEY
// F14; the partial sums of <=4096 fp32 step lengths are exact in float64, so the parallel scan
// is bit-identical to the sequential one).  The kept/culled decision of every step is a
// wavefront ballot -> one 64-bit mask word per (ray, j); compaction indices are popcounts of
// those words, so no atomics and the output order is (ray, step) exactly like
// xyz_sampled[ray_valid] of the reference.  The dense [rays x N] tensors of the reference are
// never materialised.
#include "common.hpp"

#pragma clang fp contract(off)   // bit-exact bookkeeping: never fuse mul+add (also -ffp-contract=off in build.sh)

namespace {

struct RayCtx {
    float ox, oy, oz, dx, dy, dz, tmin;
    float tfar;   // distance at which the ray leaves the AABB (used only to stop marching early, with a margin)
};

// [t0, t1] = parameter interval of the ray inside p.occ_min..occ_max (t1 < t0: misses the box).  Conservative: axes the
// ray is parallel to only decide inside / outside.  Without a box (occ_min > occ_max) the interval is everything.
__device__ __forceinline__ void occ_interval(const nmf_march_params& p, const RayCtx& c, float& t0, float& t1) {
    t0 = -3.0e38f; t1 = 3.0e38f;
    if (p.occ_min[0] > p.occ_max[0] || p.occ_min[1] > p.occ_max[1] || p.occ_min[2] > p.occ_max[2]) return;
    const float o[3] = {c.ox, c.oy, c.oz}, d[3] = {c.dx, c.dy, c.dz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (fabsf(d[a]) < 1e-12f) {
            if (o[a] < p.occ_min[a] || o[a] > p.occ_max[a]) { t0 = 1.f; t1 = 0.f; return; }
        } else {
            // (hardware reciprocal, 1 ulp: the interval only bounds the march, every use adds a 1e-2 margin; two IEEE divisions per
            // axis were ~70 of the ~900 instructions a wave of four secondary rays executes)
            const float inv = __builtin_amdgcn_rcpf(d[a]);
            const float ta = (p.occ_min[a] - o[a]) * inv, tb = (p.occ_max[a] - o[a]) * inv;
            t0 = fmaxf(t0, fminf(ta, tb));
            t1 = fminf(t1, fmaxf(ta, tb));
        }
    }
}

__device__ __forceinline__ RayCtx load_ray(const nmf_march_params& p, const float* rays, int64_t r) {
    RayCtx c;
    const float* q = rays + r * 6;
    c.ox = q[0]; c.oy = q[1]; c.oz = q[2]; c.dx = q[3]; c.dy = q[4]; c.dz = q[5];
    // alphagrid.py:149-152
    float vx = c.dx == 0.f ? 1e-6f : c.dx, vy = c.dy == 0.f ? 1e-6f : c.dy, vz = c.dz == 0.f ? 1e-6f : c.dz;
    float ax = fdiv(fsub(p.aabb_max[0], c.ox), vx), bx = fdiv(fsub(p.aabb_min[0], c.ox), vx);
    float ay = fdiv(fsub(p.aabb_max[1], c.oy), vy), by = fdiv(fsub(p.aabb_min[1], c.oy), vy);
    float az = fdiv(fsub(p.aabb_max[2], c.oz), vz), bz = fdiv(fsub(p.aabb_min[2], c.oz), vz);
    float t = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    c.tmin = fminf(fmaxf(t, p.near_t), p.far_t);
    c.tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return c;
}

// The four uniforms of one Philox counter belong to four CONSECUTIVE steps, and a wave walks 64 steps per round: instead of
// every lane running the 10 Philox rounds for its own step (and discarding 3 of the 4 outputs), lane l evaluates counter
// 64 g + l once per GROUP of four rounds (steps [256 g, 256 g + 256)) and the rounds fetch their uniform with shuffles.
// Same stream, same values: ~100 of ~190 VALU instructions per step gone (the marcher is VALU-bound).
struct JitterCache {     // four scalar members, always passed by reference: stays in VGPRs (an o[4] reached through a
    uint32_t o0, o1, o2, o3;   // conditional pointer was placed in scratch: one 16-byte scratch load per round)
    bool on;
};
__device__ __forceinline__ void jc_fill(JitterCache& c, const Philox& rng, const nmf_march_params& p, int64_t r, int g) {
    uint32_t o[4];
    rng((uint64_t)r * 1024u + (uint64_t)(64 * g + lane_id()), p.offset, o);
    c.o0 = o[0]; c.o1 = o[1]; c.o2 = o[2]; c.o3 = o[3];
}
// uniform of step 256 g + k_local (all lanes of the wave must call this together)
__device__ __forceinline__ float jc_get(const JitterCache& c, int k_local) {
    const int src = (k_local >> 2) & 63, comp = k_local & 3;
    const uint32_t v0 = __shfl(c.o0, src, 64), v1 = __shfl(c.o1, src, 64);
    const uint32_t v2 = __shfl(c.o2, src, 64), v3 = __shfl(c.o3, src, 64);
    return u32_to_unit(comp == 0 ? v0 : (comp == 1 ? v1 : (comp == 2 ? v2 : v3)));
}

// step length of candidate k (train) -- alphagrid.py:169-172
__device__ __forceinline__ float step_len(const nmf_march_params& p, const float* jitter, const Philox& rng,
                                          int64_t r, int k, const JitterCache& jc) {
    float u;
    if (jitter) {
        u = jitter[r * p.n_steps + k];
    } else if (jc.on) {
        u = jc_get(jc, k & 255);
    } else {
        uint32_t o[4];
        rng((uint64_t)r * 1024u + (uint64_t)(k >> 2), p.offset, o);
        u = u32_to_unit(o[k & 3]);
    }
    return fadd(fmul(u, p.stepsize), p.half_step);
}

// occupancy test of alphagrid.py:23-45 + ":343 alphas > 0": trilinear sample of a 0/1 volume with
// zero padding is positive iff an in-range corner with a set bit has a positive weight product.
constexpr int CB = 8;     // coarse occupancy cell = 8^3 fine voxels (+1 halo: the 8-corner footprint of any point inside)

__device__ __forceinline__ bool alpha_hit(const nmf_march_params& p, const uint32_t* bits, const uint32_t* coarse,
                                          float x, float y, float z) {
    const int gx = p.grid[0], gy = p.grid[1], gz = p.grid[2];
    float cx = fsub(fmul(fsub(x, p.aabb_min[0]), p.alpha_inv[0]), 1.f);
    float cy = fsub(fmul(fsub(y, p.aabb_min[1]), p.alpha_inv[1]), 1.f);
    float cz = fsub(fmul(fsub(z, p.aabb_min[2]), p.alpha_inv[2]), 1.f);
    // grid_sampler_unnormalize, align_corners=True: ((c + 1) / 2) * (size - 1)
    // (c + 1) / 2 == (c + 1) * 0.5 exactly (power of two)
    float ix = fmul(fmul(fadd(cx, 1.f), 0.5f), (float)(gx - 1));
    float iy = fmul(fmul(fadd(cy, 1.f), 0.5f), (float)(gy - 1));
    float iz = fmul(fmul(fadd(cz, 1.f), 0.5f), (float)(gz - 1));
    float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
    int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
    float wx[2] = {fsub(fadd(fx0, 1.f), ix), fsub(ix, fx0)};
    float wy[2] = {fsub(fadd(fy0, 1.f), iy), fsub(iy, fy0)};
    float wz[2] = {fsub(fadd(fz0, 1.f), iz), fsub(iz, fz0)};
    if (coarse) {   // LDS-resident coarse mask: clear = no set bit among the 8 corners of ANY point of the cell
        const int cgx = (gx + CB - 1) / CB, cgy = (gy + CB - 1) / CB;
        const int cx0 = min(max(x0, 0), gx - 1) / CB, cy0 = min(max(y0, 0), gy - 1) / CB, cz0 = min(max(z0, 0), gz - 1) / CB;
        const int ci = (cz0 * cgy + cy0) * cgx + cx0;
        if (!((coarse[ci >> 5] >> (ci & 31)) & 1u)) return false;
    }
    // Branch-free: the two x-neighbours of a (y, z) row are adjacent bits, fetched as one 64-bit window (two words, all
    // eight loads independent and in flight together); a corner counts when it is inside the volume, its bit is set
    // and its weight product is positive.
    const int n_words = (int)(((int64_t)gx * gy * gz + 31) >> 5);     // volumes up to 2^31 voxels (checked on the host)
    const bool okx0 = x0 >= 0 && x0 < gx, okx1 = x0 + 1 >= 0 && x0 + 1 < gx;
    unsigned any = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int by = c & 1, bz = c >> 1;
        const int Y = y0 + by, Z = z0 + bz;
        const bool okyz = Y >= 0 && Z >= 0 && Y < gy && Z < gz;
        const int Yc = min(max(Y, 0), gy - 1), Zc = min(max(Z, 0), gz - 1);
        const int idx = (Zc * gy + Yc) * gx + min(max(x0, 0), gx - 1);                  // bit of (x0 clamped, Y, Z)
        const int wi = idx >> 5;
        const uint64_t lo = bits[wi], hi = bits[min(wi + 1, n_words - 1)];
        const unsigned two = (unsigned)(((lo | (hi << 32)) >> (idx & 31)) & 3u);         // bit0: x0, bit1: x0 + 1
        const float wyz = fmul(wy[by], wz[bz]);
        const bool h0 = okx0 && (two & 1u) && fmul(fmul(wx[0], wy[by]), wz[bz]) > 0.f;
        // when x0 < 0 the clamped window starts at x = 0 = x0 + 1: its bit 0 is the x0+1 corner
        const unsigned b1 = x0 < 0 ? (two & 1u) : ((two >> 1) & 1u);
        const bool h1 = okx1 && b1 && fmul(fmul(wx[1], wy[by]), wz[bz]) > 0.f;
        (void)wyz;
        any |= (okyz && (h0 || h1)) ? 1u : 0u;
    }
    return any != 0;
}

// One j-iteration of the march for this lane: returns z (distance along the ray) of step k and
// whether the step is kept.  `carry` holds the float64 cumulative sum of all previous steps.
struct StepOut {
    float z, px, py, pz;
    bool keep;
};

__device__ __forceinline__ StepOut march_one(const nmf_march_params& p, const RayCtx& c, const float* jitter,
                                             const Philox& rng, const uint32_t* bits, const uint32_t* coarse, int64_t r,
                                             int k, double& carry, double* cum_out, const JitterCache& jc) {
    const bool in_range = k < p.n_steps;
    float step;
    double cum = 0.0;
    if (p.is_train) {
        float s = step_len(p, jitter, rng, r, in_range ? k : 0, jc);      // wave-uniform call (shuffles inside)
        s = in_range ? s : 0.f;
        double incl = wave_incl_scan_dpp((double)s);      // exact partial sums (header comment): order-free
        cum = carry + incl;
        carry += __shfl(incl, 63, 64);
        step = (float)cum;
    } else {
        step = fmul(p.stepsize, (float)k);   // alphagrid.py:190
    }
    if (cum_out) *cum_out = cum;
    StepOut o;
    o.z = fadd(c.tmin, step);                                              // :192
    o.px = fadd(c.ox, fmul(c.dx, o.z));                                    // :194
    o.py = fadd(c.oy, fmul(c.dy, o.z));
    o.pz = fadd(c.oz, fmul(c.dz, o.z));
    bool outside = (p.aabb_min[0] > o.px) | (o.px > p.aabb_max[0]) | (p.aabb_min[1] > o.py) |
                   (o.py > p.aabb_max[1]) | (p.aabb_min[2] > o.pz) | (o.pz > p.aabb_max[2]);   // :195
    o.keep = in_range && !outside;
    if (o.keep && bits) o.keep = alpha_hit(p, bits, coarse, o.px, o.py, o.pz);    // :341-346
    return o;
}

// Pass 1: one wave per ray, lane = step within the current group of 64 candidates.  Workgroups are persistent
// (grid-stride over rays) so the coarse occupancy mask is staged into LDS once per workgroup.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) k_march_count(nmf_march_params p, const float* __restrict__ rays, int64_t B,
                                                     const float* __restrict__ jitter,
                                                     const uint32_t* __restrict__ bits,
                                                     const uint32_t* __restrict__ coarse, int coarse_words,
                                                     uint64_t* __restrict__ valid, int32_t* __restrict__ counts) {
    extern __shared__ uint32_t s_coarse_buf[];
    const uint32_t* s_coarse = nullptr;
    if (coarse && bits) {
        for (int i = threadIdx.x; i < coarse_words; i += blockDim.x) s_coarse_buf[i] = coarse[i];
        __syncthreads();
        s_coarse = s_coarse_buf;
    }
    const int lane = lane_id();
    const int W = (p.n_steps + 63) >> 6;
    Philox rng(p.seed);
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < B; r += (int64_t)gridDim.x * 4) {
        RayCtx c = load_ray(p, rays, r);
        double carry = 0.0;
        int total = 0;
        int j = 0;
        // with an alpha mask nothing is kept outside the box around its set voxels: rays that miss it need no round at
        // all, the others stop once they have left it (1e-2 margin as for the AABB exit below)
        float t_stop = c.tfar;
        bool hopeless = false;
        if (bits) {
            float o0, o1;
            occ_interval(p, c, o0, o1);
            hopeless = o1 < o0 || o1 < c.tmin - 1e-2f || o0 > c.tfar + 1e-2f;
            t_stop = fminf(t_stop, o1);
        }
        JitterCache jc;
        jc.o0 = jc.o1 = jc.o2 = jc.o3 = 0u;
        const bool use_jc = p.is_train && !jitter;
        jc.on = use_jc;
        for (; j < W && !hopeless; ++j) {
            if (use_jc && (j & 3) == 0) jc_fill(jc, rng, p, r, j >> 2);
            StepOut o = march_one(p, c, jitter, rng, bits, s_coarse, r, j * 64 + lane, carry, nullptr,
                                  jc);
            uint64_t m = __ballot(o.keep);
            total += __popcll(m);
            if (lane == 0) valid[r * W + j] = m;
            // z grows monotonically along the ray: once the last step of this chunk is clearly beyond the AABB exit
            // every later step fails the in-box test (:195) too.  The 1e-2 margin dwarfs the fp32 error of the positions.
            const float z_last = __shfl(o.z, 63, 64);
            if (z_last > t_stop + 1e-2f) { ++j; break; }
        }
        for (int jj = j + lane; jj < W; jj += 64) valid[r * W + jj] = 0ull;
        if (lane == 0) counts[r] = total;
    }
}

// ---- 16 lanes per ray -----------------------------------------------------------------------------------------------
// The re-traced secondary rays (0.24 M per step) keep ~4 samples each, all within their first few steps, and leave the
// occupied box right after: a wave per ray spends a whole 64-step round (Philox, float64 scan, 64 alpha tests) on them.
// Here a wave carries FOUR rays, one per row of 16 lanes, in rounds of 16 steps: the same arithmetic per step (the
// partial sums of the step lengths are exact in float64, so neither the round width nor the scan order matters), the
// float64 scan stays inside a DPP row, the Philox cache covers 64 steps = 4 rounds per fill.  A row that is done idles
// (masked) until the other three are; the valid words of a ray are assembled from its 16-bit pieces.
__device__ __forceinline__ double row_incl_scan_dpp(double v) {
    v += dpp_f64<0x111, 0xf, 0xf>(v);
    v += dpp_f64<0x112, 0xf, 0xf>(v);
    v += dpp_f64<0x114, 0xf, 0xf>(v);
    v += dpp_f64<0x118, 0xf, 0xf>(v);
    return v;
}
// lane l16 of a row evaluates counter 16 g + l16 of the ray (steps [64 g, 64 g + 64))
__device__ __forceinline__ void jc16_fill(JitterCache& c, const Philox& rng, const nmf_march_params& p, int64_t r, int g) {
    uint32_t o[4];
    rng((uint64_t)r * 1024u + (uint64_t)(16 * g + (lane_id() & 15)), p.offset, o);
    c.o0 = o[0]; c.o1 = o[1]; c.o2 = o[2]; c.o3 = o[3];
}
// uniform of step 64 g + k_local (0..63) of this lane's row (all lanes of the wave call this together)
__device__ __forceinline__ float jc16_get(const JitterCache& c, int k_local) {
    const int src = (k_local >> 2) & 15, comp = k_local & 3;
    const uint32_t v0 = __shfl(c.o0, src, 16), v1 = __shfl(c.o1, src, 16);
    const uint32_t v2 = __shfl(c.o2, src, 16), v3 = __shfl(c.o3, src, 16);
    return u32_to_unit(comp == 0 ? v0 : (comp == 1 ? v1 : (comp == 2 ? v2 : v3)));
}
// one round of 16 steps of this lane's row: step k = 16 s + l16
__device__ __forceinline__ StepOut march_one16(const nmf_march_params& p, const RayCtx& c, const float* jitter,
                                               const Philox& rng, const uint32_t* bits, const uint32_t* coarse, int64_t r,
                                               int k, bool live, double& carry, double* cum_out, float* s_out,
                                               const JitterCache& jc, const JitterCache& jc_off) {
    const bool in_range = k < p.n_steps;
    float step;
    double cum = 0.0;
    float s = 0.f;
    if (p.is_train) {
        if (jc.on) s = fadd(fmul(jc16_get(jc, k & 63), p.stepsize), p.half_step);
        else s = step_len(p, jitter, rng, r, in_range ? k : 0, jc_off);
        s = in_range ? s : 0.f;
        const double incl = row_incl_scan_dpp((double)s);
        cum = carry + incl;
        carry += __shfl(incl, 15, 16);
        step = (float)cum;
    } else {
        step = fmul(p.stepsize, (float)k);   // alphagrid.py:190
    }
    if (cum_out) *cum_out = cum;
    if (s_out) *s_out = s;
    StepOut o;
    o.z = fadd(c.tmin, step);                                              // :192
    o.px = fadd(c.ox, fmul(c.dx, o.z));                                    // :194
    o.py = fadd(c.oy, fmul(c.dy, o.z));
    o.pz = fadd(c.oz, fmul(c.dz, o.z));
    bool outside = (p.aabb_min[0] > o.px) | (o.px > p.aabb_max[0]) | (p.aabb_min[1] > o.py) |
                   (o.py > p.aabb_max[1]) | (p.aabb_min[2] > o.pz) | (o.pz > p.aabb_max[2]);   // :195
    o.keep = live && in_range && !outside;
    if (o.keep && bits) o.keep = alpha_hit(p, bits, coarse, o.px, o.py, o.pz);    // :341-346
    return o;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_march_count16(nmf_march_params p, const float* __restrict__ rays, int64_t B, const float* __restrict__ jitter,
                const uint32_t* __restrict__ bits, const uint32_t* __restrict__ coarse, int coarse_words,
                uint64_t* __restrict__ valid, int32_t* __restrict__ counts) {
    extern __shared__ uint32_t s_coarse_buf[];
    const uint32_t* s_coarse = nullptr;
    if (coarse && bits) {
        for (int i = threadIdx.x; i < coarse_words; i += blockDim.x) s_coarse_buf[i] = coarse[i];
        __syncthreads();
        s_coarse = s_coarse_buf;
    }
    const int lane = lane_id(), l16 = lane & 15, row = lane >> 4;
    const int W = (p.n_steps + 63) >> 6;
    const int n_rounds = (p.n_steps + 15) >> 4;
    Philox rng(p.seed);
    JitterCache jc, jc_off;
    jc.o0 = jc.o1 = jc.o2 = jc.o3 = 0u;
    jc.on = p.is_train && !jitter;
    jc_off = jc;
    jc_off.on = false;
    const int64_t n_quads = (B + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); q < n_quads; q += (int64_t)gridDim.x * 4) {
        const int64_t r = q * 4 + row;
        const bool active = r < B;
        RayCtx c = load_ray(p, rays, active ? r : B - 1);
        double carry = 0.0;
        int total = 0, next_word = 0;
        uint64_t word = 0;
        float t_stop = c.tfar;
        bool live = active;
        if (bits) {
            float o0, o1;
            occ_interval(p, c, o0, o1);
            if (o1 < o0 || o1 < c.tmin - 1e-2f || o0 > c.tfar + 1e-2f) live = false;
            t_stop = fminf(t_stop, o1);
        }
        for (int s = 0; s < n_rounds; ++s) {
            if (__ballot(live) == 0ull) break;                               // every row of the wave is done
            if (jc.on && (s & 3) == 0) jc16_fill(jc, rng, p, active ? r : 0, s >> 2);
            StepOut o = march_one16(p, c, jitter, rng, bits, s_coarse, active ? r : 0, 16 * s + l16, live, carry, nullptr,
                                    nullptr, jc, jc_off);
            const uint64_t m = __ballot(o.keep);
            const uint64_t piece = (m >> (16 * row)) & 0xffffull;
            total += __popcll(piece);
            word |= piece << (16 * (s & 3));
            const float z_last = __shfl(o.z, 15, 16);
            const bool finishing = live && (z_last > t_stop + 1e-2f || s == n_rounds - 1);
            if (live && ((s & 3) == 3 || finishing)) {
                if (l16 == 0) valid[r * W + (s >> 2)] = word;
                next_word = (s >> 2) + 1;
                word = 0;
            }
            if (finishing) live = false;
        }
        if (active) {
            for (int jj = next_word + l16; jj < W; jj += 16) valid[r * W + jj] = 0ull;
            if (l16 == 0) counts[r] = total;
        }
    }
}

__global__ void __launch_bounds__(256) k_march_fill16(nmf_march_params p, const float* __restrict__ rays, int64_t b,
                                                      const float* __restrict__ jitter,
                                                      const uint64_t* __restrict__ valid,
                                                      const int64_t* __restrict__ offsets, float4* __restrict__ xyzt,
                                                      int32_t* __restrict__ ray_id, int32_t* __restrict__ step_id,
                                                      float* __restrict__ zout, float* __restrict__ dist) {
    const int lane = lane_id(), l16 = lane & 15, row = lane >> 4;
    const int64_t r = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + row;
    const bool active = r < b;
    const int W = (p.n_steps + 63) >> 6;
    const int n_rounds = (p.n_steps + 15) >> 4;
    int64_t base = active ? offsets[r] : 0;
    const int64_t end = active ? offsets[r + 1] : 0;
    bool live = active && end > base;
    if (__ballot(live) == 0ull) return;        // wave-uniform: nothing kept on these four rays
    RayCtx c = load_ray(p, rays, active ? r : b - 1);
    Philox rng(p.seed);
    double carry = 0.0;
    JitterCache jc, jc_off;
    jc.o0 = jc.o1 = jc.o2 = jc.o3 = 0u;
    jc.on = p.is_train && !jitter;
    jc_off = jc;
    jc_off.on = false;
    const int64_t rr = active ? r : 0;
    for (int s = 0; s < n_rounds; ++s) {
        if (__ballot(live) == 0ull) break;
        const int k = 16 * s + l16;
        double cum;
        if (jc.on && (s & 3) == 0) jc16_fill(jc, rng, p, rr, s >> 2);
        StepOut o = march_one16(p, c, jitter, rng, nullptr, nullptr, rr, k, live, carry, &cum, nullptr, jc, jc_off);
        // step length of the NEXT candidate (the sample's dist, :348-350), fetched while the wave is converged: it sits in
        // the row's jitter cache except for the last step of a 64-step group
        float s_next = 0.f;
        if (dist && p.is_train) {
            const int kn = min(k + 1, p.n_steps - 1);
            if (jc.on) {
                s_next = fadd(fmul(jc16_get(jc, kn & 63), p.stepsize), p.half_step);
                if ((k & 63) == 63) s_next = step_len(p, jitter, rng, rr, kn, jc_off);
            } else {
                s_next = step_len(p, jitter, rng, rr, kn, jc_off);
            }
        }
        const uint64_t mw = live ? valid[r * W + (s >> 2)] : 0ull;
        const uint32_t piece = (uint32_t)((mw >> (16 * (s & 3))) & 0xffffull);
        if ((piece >> l16) & 1u) {
            const int64_t idx = base + __popc(piece & ((1u << l16) - 1u));
            if (xyzt) xyzt[idx] = make_float4(o.px, o.py, o.pz, fdiv(o.z, p.focal));       // :200
            if (ray_id) ray_id[idx] = (int32_t)r;
            if (step_id) step_id[idx] = k;
            if (zout) zout[idx] = o.z;
            if (dist) {
                float d = 0.f;                                                              // :348-350
                if (k + 1 < p.n_steps) {
                    float znext;
                    if (p.is_train) znext = fadd(c.tmin, (float)(cum + (double)s_next));
                    else znext = fadd(c.tmin, fmul(p.stepsize, (float)(k + 1)));
                    d = fsub(znext, o.z);
                }
                dist[idx] = d;
            }
        }
        base += __popc(piece);
        if (base >= end) live = false;            // every kept sample of this ray has been written
    }
}

// coarse[c] = OR of the fine bits of the 9^3 voxels [8c, 8c+8]^3 (clamped): the union of the 8-corner footprints of all
// points whose floor coordinates fall into coarse cell c
__global__ void __launch_bounds__(256) k_alpha_coarse(const uint32_t* __restrict__ bits, int gx, int gy, int gz,
                                                      uint32_t* __restrict__ coarse) {
    const int cgx = (gx + CB - 1) / CB, cgy = (gy + CB - 1) / CB, cgz = (gz + CB - 1) / CB;
    const int n = cgx * cgy * cgz;
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    bool on = false;
    if (ci < n) {
        const int cx = ci % cgx, cy = (ci / cgx) % cgy, cz = ci / (cgx * cgy);
        for (int z = cz * CB; z <= min(cz * CB + CB, gz - 1) && !on; ++z)
            for (int y = cy * CB; y <= min(cy * CB + CB, gy - 1) && !on; ++y)
                for (int x = cx * CB; x <= min(cx * CB + CB, gx - 1); ++x) {
                    const int64_t idx = ((int64_t)z * gy + y) * gx + x;
                    if ((bits[idx >> 5] >> (idx & 31)) & 1u) { on = true; break; }
                }
    }
    const uint64_t m = __ballot(on);
    const int lane = lane_id();
    if ((lane & 31) == 0 && (ci >> 5) * 32 < n) coarse[ci >> 5] = lane == 0 ? (uint32_t)m : (uint32_t)(m >> 32);
}

__global__ void __launch_bounds__(256) k_march_fill(nmf_march_params p, const float* __restrict__ rays, int64_t b,
                                                    const float* __restrict__ jitter,
                                                    const uint64_t* __restrict__ valid,
                                                    const int64_t* __restrict__ offsets, float4* __restrict__ xyzt,
                                                    int32_t* __restrict__ ray_id, int32_t* __restrict__ step_id,
                                                    float* __restrict__ zout, float* __restrict__ dist) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= b) return;
    const int lane = lane_id();
    const int W = (p.n_steps + 63) >> 6;
    int64_t base = offsets[r];
    const int64_t end = offsets[r + 1];
    if (end == base) return;                  // wave-uniform: nothing kept on this ray
    RayCtx c = load_ray(p, rays, r);
    Philox rng(p.seed);
    double carry = 0.0;
    JitterCache jc, jc_off;
    jc.o0 = jc.o1 = jc.o2 = jc.o3 = 0u;
    jc.on = p.is_train && !jitter;
    jc_off = jc;
    jc_off.on = false;
    for (int j = 0; j < W; ++j) {
        const int k = j * 64 + lane;
        double cum;
        if (jc.on && (j & 3) == 0) jc_fill(jc, rng, p, r, j >> 2);
        // positions are recomputed (cheap) instead of being stored by pass 1
        StepOut o = march_one(p, c, jitter, rng, nullptr, nullptr, r, k, carry, &cum, jc);
        // step length of the NEXT candidate (the sample's dist, :348-350), fetched while the wave is still converged:
        // it sits in the jitter cache except for the last step of a 256-step group (next Philox counter group)
        float s_next = 0.f;
        if (dist && p.is_train) {
            const int kn = min(k + 1, p.n_steps - 1);
            if (jc.on) {
                s_next = fadd(fmul(jc_get(jc, kn & 255), p.stepsize), p.half_step);
                if ((k & 255) == 255) s_next = step_len(p, jitter, rng, r, kn, jc_off);
            } else {
                s_next = step_len(p, jitter, rng, r, kn, jc_off);
            }
        }
        const uint64_t m = valid[r * W + j];
        if ((m >> lane) & 1ull) {
            int64_t idx = base + __popcll(m & ((1ull << lane) - 1ull));
            if (xyzt) xyzt[idx] = make_float4(o.px, o.py, o.pz, fdiv(o.z, p.focal));       // :200
            if (ray_id) ray_id[idx] = (int32_t)r;
            if (step_id) step_id[idx] = k;
            if (zout) zout[idx] = o.z;
            if (dist) {
                float d = 0.f;                                                              // :348-350
                if (k + 1 < p.n_steps) {
                    float znext;
                    if (p.is_train) znext = fadd(c.tmin, (float)(cum + (double)s_next));
                    else znext = fadd(c.tmin, fmul(p.stepsize, (float)(k + 1)));
                    d = fsub(znext, o.z);
                }
                dist[idx] = d;
            }
        }
        base += __popcll(m);
        if (base >= end) break;               // every kept sample of this ray has been written
    }
}

__global__ void __launch_bounds__(256) k_march_dense(nmf_march_params p, const float* __restrict__ rays, int64_t b,
                                                     const float* __restrict__ jitter,
                                                     const uint64_t* __restrict__ valid,
                                                     uint8_t* __restrict__ ray_valid, float* __restrict__ z_vals) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= b) return;
    const int lane = lane_id();
    const int W = (p.n_steps + 63) >> 6;
    RayCtx c = load_ray(p, rays, r);
    Philox rng(p.seed);
    double carry = 0.0;
    JitterCache jc_off;
    jc_off.o0 = jc_off.o1 = jc_off.o2 = jc_off.o3 = 0u;
    jc_off.on = false;
    for (int j = 0; j < W; ++j) {
        const int k = j * 64 + lane;
        StepOut o = march_one(p, c, jitter, rng, nullptr, nullptr, r, k, carry, nullptr, jc_off);
        if (k < p.n_steps) {
            if (ray_valid) ray_valid[r * p.n_steps + k] = (uint8_t)((valid[r * W + j] >> lane) & 1ull);
            if (z_vals) z_vals[r * p.n_steps + k] = o.z;
        }
    }
}

__global__ void k_alpha_pack(const float* __restrict__ vol, int64_t n, uint32_t* __restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // grid covers ceil(n/256)*256
    const bool on = i < n && vol[i] > 0.f;
    const uint64_t m = __ballot(on);
    const int lane = lane_id();
    if ((lane & 31) == 0) {
        const int64_t w = i >> 5;
        if (w * 32 < n) bits[w] = lane == 0 ? (uint32_t)m : (uint32_t)(m >> 32);
    }
}

// Exclusive scan with the sample budget of alphagrid.py:353-364, three launches:
//   k_scan_partial : sum of each 1024-element chunk
//   k_scan_top     : one workgroup scans the chunk sums (<= 4096 chunks = 4 M elements), derives the total and
//                    whether the budget is active, and initialises totals = {0, 0}
//   k_scan_final   : every chunk scans itself, adds its base, applies the budget and finds (M, b)
constexpr int SCAN_CHUNK = 1024;

__global__ void __launch_bounds__(SCAN_CHUNK) k_scan_partial(const int32_t* __restrict__ counts, int64_t B,
                                                             int64_t* __restrict__ chunk_sum) {
    __shared__ int64_t ws[16];
    const int64_t i = (int64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x;
    int64_t v = i < B ? counts[i] : 0;
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
        for (int w = 0; w < 16; ++w) t += ws[w];
        chunk_sum[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(1024) k_scan_top(int64_t* __restrict__ chunk_sum, int n_chunks, int64_t B,
                                                   int64_t max_samples, int64_t* __restrict__ offsets,
                                                   int64_t* __restrict__ totals, int64_t* __restrict__ meta) {
    __shared__ int64_t ws[16];
    __shared__ int64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_chunks; base += 1024) {
        const int i = base + tid;
        const int64_t v = i < n_chunks ? chunk_sum[i] : 0;
        int64_t incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            int64_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) ws[wid] = incl;
        __syncthreads();
        int64_t woff = 0;
        for (int w = 0; w < wid; ++w) woff += ws[w];
        const int64_t excl = carry_s + woff + incl - v;
        if (i < n_chunks) chunk_sum[i] = excl;           // becomes the chunk base
        __syncthreads();
        if (tid == 1023) carry_s = excl + v;
        __syncthreads();
    }
    if (tid == 0) {
        const int64_t total = carry_s;
        const bool budget = max_samples > 0 && total > max_samples;
        meta[0] = total;
        meta[1] = budget ? 1 : 0;
        offsets[B] = total;                              // clamped by k_scan_final when the budget is active
        totals[0] = budget ? 0 : total;
        totals[1] = budget ? 0 : B;
    }
}

__global__ void __launch_bounds__(SCAN_CHUNK) k_scan_final(const int32_t* __restrict__ counts, int64_t B,
                                                           int64_t max_samples, const int64_t* __restrict__ chunk_base,
                                                           const int64_t* __restrict__ meta,
                                                           int64_t* __restrict__ offsets,
                                                           uint8_t* __restrict__ whole_valid,
                                                           int64_t* __restrict__ totals) {
    __shared__ int64_t ws[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * SCAN_CHUNK + tid;
    const int64_t v = i < B ? counts[i] : 0;
    int64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        int64_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) ws[wid] = incl;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += ws[w];
    const int64_t cum = chunk_base[blockIdx.x] + woff + incl;       // inclusive cumsum(counts)[i]
    const bool budget = meta[1] != 0;
    if (i < B) {
        if (!budget) {
            whole_valid[i] = 1;
            offsets[i] = cum - v;
        } else {
            const bool ok = cum < max_samples;                          // strict '<' (alphagrid.py:359)
            whole_valid[i] = ok ? 1 : 0;
            // valid rays form a prefix; the last valid one defines (M, b)
            const int64_t next = (i + 1 < B) ? cum + counts[i + 1] : max_samples;
            if (ok && !(next < max_samples)) { totals[0] = cum; totals[1] = i + 1; }
            // kept total M is the largest cum < max_samples; offsets of dropped rays are clamped to it below
            offsets[i] = ok ? cum - v : -1;
        }
    }
}

// budget active: offsets of dropped rays (marked -1) and offsets[B] become M
__global__ void k_scan_clamp(int64_t* __restrict__ offsets, int64_t B, const int64_t* __restrict__ meta,
                             const int64_t* __restrict__ totals) {
    if (meta[1] == 0) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > B) return;
    if (i == B || offsets[i] < 0) offsets[i] = totals[0];
}

// ---- the same scan in two launches (one for B <= 4096) ---------------------------------------------------------------
// k_scan_top / k_scan_final / k_scan_clamp above are three dependent launches of a few microseconds each around ~10 KB of
// data -- a fifth of the level-0 sampler's time.  With at most 4096 chunks every workgroup of the final pass can scan the
// chunk sums itself (4 per thread): its own base, the grand total, whether the budget binds and in which chunk c* the
// cumulative count crosses it.  Rays past the crossing take offsets = M directly (workgroups behind c* rescan chunk c*
// to learn M), so no clamp pass is needed.  With at most 4 chunks the chunk sums are computed here as well (no
// k_scan_partial).  Same outputs as the three-launch form (tests compare them).
__device__ __forceinline__ int64_t block_scan_incl(int64_t v, int64_t* ws, int tid, int64_t& block_total) {
    const int lane = tid & 63, wid = tid >> 6;
    int64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    __syncthreads();                 // ws may still be read by the previous use
    if (lane == 63) ws[wid] = incl;
    __syncthreads();
    int64_t woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
        const int64_t x = ws[w];
        if (w < wid) woff += x;
        tot += x;
    }
    block_total = tot;
    return woff + incl;
}

__global__ void __launch_bounds__(SCAN_CHUNK) k_scan_fused(const int32_t* __restrict__ counts, int64_t B, int64_t max_samples,
                                                           const int64_t* __restrict__ chunk_sum, int n_chunks,
                                                           int64_t* __restrict__ offsets, uint8_t* __restrict__ whole_valid,
                                                           int64_t* __restrict__ totals, int64_t* pub, int64_t pub_seq) {
    __shared__ int64_t ws[16];
    __shared__ int64_t s_base, s_cbase, s_direct[4];
    __shared__ int s_tot[4];
    __shared__ int s_cstar;
    const int tid = threadIdx.x;
    const int my = (int)blockIdx.x;
    // ---- phase A: the chunk sums, 4 per thread (chunk 4 tid + k)
    int64_t v4[4] = {0, 0, 0, 0};
    int64_t total, excl_t;
    int own = 0;                                     // this thread's count in this workgroup's chunk (direct path: already loaded)
    if (chunk_sum) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int c = 4 * tid + k; v4[k] = c < n_chunks ? chunk_sum[c] : 0; }
        const int64_t tsum = v4[0] + v4[1] + v4[2] + v4[3];
        excl_t = block_scan_incl(tsum, ws, tid, total) - tsum;
    } else {
        // n_chunks <= 4 (the primary rays of a chunk): every workgroup sums the chunks itself -- the counts of all four in flight
        // together and ONE reduction (R3 ran a block scan with its own dependent load per chunk, then a scan over the four sums:
        // six block scans and five memory round trips in a launch on the serial head of every step)
        int cv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t ic = (int64_t)c * SCAN_CHUNK + tid;
            cv[c] = (c < n_chunks && ic < B) ? counts[ic] : 0;
        }
        own = my == 0 ? cv[0] : (my == 1 ? cv[1] : (my == 2 ? cv[2] : cv[3]));
        if (tid < 4) s_tot[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int r = cv[c];
            for (int d = 32; d > 0; d >>= 1) r += __shfl_down(r, d, 64);
            if ((tid & 63) == 0 && r) atomicAdd(&s_tot[c], r);
        }
        __syncthreads();
        total = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) { if (tid == 0) v4[c] = s_tot[c]; total += s_tot[c]; }
        excl_t = 0;                                  // (thread 0 holds chunks 0-3)
    }
    const bool budget = max_samples > 0 && total > max_samples;
    {
        int64_t run = excl_t;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * tid + k;
            if (c == my) s_base = run;
            if (budget && c < n_chunks && run < max_samples && run + v4[k] >= max_samples) { s_cstar = c; s_cbase = run; }
            run += v4[k];
        }
    }
    __syncthreads();
    const int cstar = budget ? s_cstar : n_chunks;
    // ---- phase B: this chunk
    const int64_t i = (int64_t)my * SCAN_CHUNK + tid;
    const int64_t v = chunk_sum ? (i < B ? counts[i] : 0) : own;
    int64_t dummy;
    const int64_t cum = s_base + block_scan_incl(v, ws, tid, dummy);       // inclusive cumsum(counts)[i]
    const bool ok = !budget || cum < max_samples;                          // strict '<' (alphagrid.py:359)
    // ---- M (kept samples) for the chunks that hold dropped rays: the largest cumulative count below the budget
    int64_t M = 0;
    if (budget && my >= cstar) {
        int64_t cum_c = cum;
        bool in_c = i < B;
        if (my > cstar) {
            const int64_t ic = (int64_t)cstar * SCAN_CHUNK + tid;
            in_c = ic < B;
            cum_c = s_cbase + block_scan_incl(in_c ? counts[ic] : 0, ws, tid, dummy);
        }
        const bool ok_c = in_c && cum_c < max_samples;
        int64_t n_ok;
        block_scan_incl(ok_c ? 1 : 0, ws, tid, n_ok);                      // valid rays are a prefix: count them
        __syncthreads();
        if (tid == 0) s_direct[0] = s_cbase;                               // no valid ray in c*: M = its base
        __syncthreads();
        if (ok_c && tid + 1 == n_ok) s_direct[0] = cum_c;                  // the last valid ray
        __syncthreads();
        M = s_direct[0];
        if (my == cstar && tid == 0) {
            totals[0] = M;
            totals[1] = (int64_t)cstar * SCAN_CHUNK + n_ok;
            offsets[B] = M;
            publish_sizes(pub, M, (int64_t)cstar * SCAN_CHUNK + n_ok, pub_seq);
        }
    }
    if (!budget && my == 0 && tid == 0) {
        totals[0] = total; totals[1] = B; offsets[B] = total;
        publish_sizes(pub, total, B, pub_seq);
    }
    if (i < B) {
        whole_valid[i] = ok ? 1 : 0;
        offsets[i] = ok ? cum - v : M;
    }
}

}  // namespace

extern "C" int nmf_alpha_pack(const float* volume, int64_t n_voxels, uint32_t* bits, void* stream) {
    NMF_REQUIRE(volume && bits && n_voxels > 0, NMF_EINVAL, "nmf_alpha_pack: null/empty");
    NMF_LAUNCH(k_alpha_pack, dim3((unsigned)cdiv(n_voxels, 256)), dim3(256), 0, (hipStream_t)stream, volume,
                       n_voxels, bits);
    NMF_CHECK_LAUNCH("nmf_alpha_pack");
    return NMF_OK;
}

static int check_params(const nmf_march_params* p) {
    NMF_REQUIRE(p, NMF_EINVAL, "march: params null");
    NMF_REQUIRE(p->n_steps > 0 && p->n_steps <= 4096, NMF_ERANGE, "march: n_steps outside (0,4096]");
    NMF_REQUIRE((int64_t)p->grid[0] * p->grid[1] * p->grid[2] < (1ll << 31), NMF_ERANGE, "march: alpha volume >= 2^31 voxels");
    return NMF_OK;
}

extern "C" int64_t nmf_alpha_coarse_words(const int32_t grid[3]) {
    if (!grid) return 0;
    const int64_t n = (int64_t)((grid[0] + CB - 1) / CB) * ((grid[1] + CB - 1) / CB) * ((grid[2] + CB - 1) / CB);
    return (n + 31) / 32;
}

extern "C" int nmf_alpha_coarse(const uint32_t* bits, const int32_t grid[3], uint32_t* coarse, void* stream) {
    NMF_REQUIRE(bits && grid && coarse && grid[0] > 0 && grid[1] > 0 && grid[2] > 0, NMF_EINVAL, "nmf_alpha_coarse: null/size");
    const int64_t words = nmf_alpha_coarse_words(grid);
    NMF_LAUNCH(k_alpha_coarse, dim3((unsigned)cdiv(words * 32, 256)), dim3(256), 0, (hipStream_t)stream, bits,
                       grid[0], grid[1], grid[2], coarse);
    NMF_CHECK_LAUNCH("nmf_alpha_coarse");
    return NMF_OK;
}

// 64 lanes per ray for primary-ray batches (few rays, every round of 64 steps is needed), 16 for large batches (the
// re-traced secondary rays finish within their first rounds).  NMF_MARCH_LANES = 16 | 64 forces one (tests compare both).
static int lanes_per_ray(int64_t n_rays) {
    if (const char* ev = getenv("NMF_MARCH_LANES")) {
        const int v = atoi(ev);
        if (v == 16 || v == 64) return v;
    }
    return n_rays > 16384 ? 16 : 64;
}

extern "C" int nmf_march_count(const nmf_march_params* p, const float* rays, int64_t B, const float* jitter,
                               const uint32_t* alpha_bits, const uint32_t* alpha_coarse, uint64_t* valid_bits,
                               int32_t* counts, void* stream) {
    if (int e = check_params(p)) return e;
    NMF_REQUIRE(B >= 0 && (B == 0 || (rays && valid_bits && counts)), NMF_EINVAL, "nmf_march_count: null");
    if (B == 0) return NMF_OK;
    int64_t words = (alpha_bits && alpha_coarse) ? nmf_alpha_coarse_words(p->grid) : 0;
    if (words * 4 > 60 * 1024) { alpha_coarse = nullptr; words = 0; }        // mask larger than the LDS budget: skip it
    if (lanes_per_ray(B) == 16) {      // many rays (the re-traced secondary rays): four rays per wave, rounds of 16 steps
        int64_t blocks = cdiv(cdiv(B, 4), 4);
        if (blocks > 256 * 16) blocks = 256 * 16;
        NMF_LAUNCH(k_march_count16, dim3((unsigned)blocks), dim3(256), (size_t)words * 4, (hipStream_t)stream, *p,
                           rays, B, jitter, alpha_bits, alpha_coarse, (int)words, valid_bits, counts);
        NMF_CHECK_LAUNCH("nmf_march_count");
        return NMF_OK;
    }
    int64_t blocks = cdiv(B, 4);
    if (blocks > 256 * 16) blocks = 256 * 16;                                 // persistent: 16 workgroups per CU
    NMF_LAUNCH(k_march_count, dim3((unsigned)blocks), dim3(256), (size_t)words * 4, (hipStream_t)stream, *p, rays,
                       B, jitter, alpha_bits, alpha_coarse, (int)words, valid_bits, counts);
    NMF_CHECK_LAUNCH("nmf_march_count");
    return NMF_OK;
}

extern "C" int64_t nmf_march_scan_workspace_bytes(int64_t B) {
    return (cdiv(B, SCAN_CHUNK) + 2) * (int64_t)sizeof(int64_t);
}

extern "C" int nmf_march_scan(const int32_t* counts, int64_t B, int64_t max_samples, int64_t* offsets,
                              uint8_t* whole_valid, int64_t* totals, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    return nmf_march_scan_publish(counts, B, max_samples, offsets, whole_valid, totals, workspace, workspace_bytes, nullptr, 0,
                                  stream);
}

extern "C" int nmf_march_scan_publish(const int32_t* counts, int64_t B, int64_t max_samples, int64_t* offsets,
                                      uint8_t* whole_valid, int64_t* totals, void* workspace, int64_t workspace_bytes,
                                      void* publish_mapped_dev, int64_t publish_seq, void* stream) {
    int64_t* pub = static_cast<int64_t*>(publish_mapped_dev);
    NMF_REQUIRE(counts && offsets && whole_valid && totals && B > 0, NMF_EINVAL, "nmf_march_scan: null/empty");
    const int64_t n_chunks = cdiv(B, SCAN_CHUNK);
    NMF_REQUIRE(n_chunks <= (1 << 22), NMF_ERANGE, "nmf_march_scan: B too large");
    NMF_REQUIRE(workspace && workspace_bytes >= nmf_march_scan_workspace_bytes(B), NMF_EINVAL,
                "nmf_march_scan: workspace too small (see nmf_march_scan_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    int64_t* chunk = (int64_t*)workspace;
    int64_t* meta = chunk + n_chunks;
    const bool three_pass = getenv("NMF_SCAN_3PASS") != nullptr;             // the round-1 form (tests compare the two)
    if (n_chunks <= 4096 && !three_pass) {
        if (n_chunks > 4) NMF_LAUNCH(k_scan_partial, dim3((unsigned)n_chunks), dim3(SCAN_CHUNK), 0, st, counts, B, chunk);
        NMF_LAUNCH(k_scan_fused, dim3((unsigned)n_chunks), dim3(SCAN_CHUNK), 0, st, counts, B, max_samples,
                           n_chunks > 4 ? chunk : nullptr, (int)n_chunks, offsets, whole_valid, totals, pub, publish_seq);
        NMF_CHECK_LAUNCH("nmf_march_scan");
        return NMF_OK;
    }
    NMF_LAUNCH(k_scan_partial, dim3((unsigned)n_chunks), dim3(SCAN_CHUNK), 0, st, counts, B, chunk);
    NMF_LAUNCH(k_scan_top, dim3(1), dim3(1024), 0, st, chunk, (int)n_chunks, B, max_samples, offsets, totals,
                       meta);
    NMF_LAUNCH(k_scan_final, dim3((unsigned)n_chunks), dim3(SCAN_CHUNK), 0, st, counts, B, max_samples, chunk,
                       meta, offsets, whole_valid, totals);
    NMF_LAUNCH(k_scan_clamp, dim3((unsigned)cdiv(B + 1, 256)), dim3(256), 0, st, offsets, B, meta, totals);
    NMF_CHECK_LAUNCH("nmf_march_scan");
    if (pub) return nmf_publish_i64x2(totals, pub, publish_seq, stream);
    return NMF_OK;
}

extern "C" int nmf_march_fill(const nmf_march_params* p, const float* rays, int64_t b, const float* jitter,
                              const uint64_t* valid_bits, const int64_t* offsets, float* xyzt, int32_t* ray_id,
                              int32_t* step_id, float* z, float* dist, void* stream) {
    if (int e = check_params(p)) return e;
    NMF_REQUIRE(b >= 0 && (b == 0 || (rays && valid_bits && offsets)), NMF_EINVAL, "nmf_march_fill: null");
    if (b == 0) return NMF_OK;
    if (lanes_per_ray(b) == 16)
        NMF_LAUNCH(k_march_fill16, dim3((unsigned)cdiv(cdiv(b, 4), 4)), dim3(256), 0, (hipStream_t)stream, *p, rays, b,
                           jitter, valid_bits, offsets, (float4*)xyzt, ray_id, step_id, z, dist);
    else
        NMF_LAUNCH(k_march_fill, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, (hipStream_t)stream, *p, rays, b, jitter,
                           valid_bits, offsets, (float4*)xyzt, ray_id, step_id, z, dist);
    NMF_CHECK_LAUNCH("nmf_march_fill");
    return NMF_OK;
}

extern "C" int nmf_march_dense(const nmf_march_params* p, const float* rays, int64_t b, const float* jitter,
                               const uint64_t* valid_bits, uint8_t* ray_valid, float* z_vals, void* stream) {
    if (int e = check_params(p)) return e;
    NMF_REQUIRE(b >= 0 && (b == 0 || (rays && valid_bits)), NMF_EINVAL, "nmf_march_dense: null");
    if (b == 0) return NMF_OK;
    NMF_LAUNCH(k_march_dense, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, (hipStream_t)stream, *p, rays, b, jitter,
                       valid_bits, ray_valid, z_vals);
    NMF_CHECK_LAUNCH("nmf_march_dense");
    return NMF_OK;
}
