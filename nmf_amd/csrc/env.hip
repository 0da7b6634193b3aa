// Prefiltered equirectangular environment map for gfx950: summed-area-table build, box lookup with
// seam / pole wrapping, and the backward of both.
// Replaces IntegralEquirect.forward / sa2mip / integrate_area* of the reference
// (modules/integral_equirect.py:18-173, 373-504) and safemath.atan2 (modules/safemath.py:8-30).
//
// Numerics: the reference's fp32 SAT differences cancel catastrophically, so parity
// needs the oracle's exact rounding: prefix sums carry a float64 running value rounded to fp32 per
// element (H first, then W), the bilinear tap sum is ATen's fma chain nw,ne,sw,se and the four corner
// samples combine as (tr + bl - tl - br) / size.  This file is compiled with -ffp-contract=off and
// spells every fma explicitly.
//
// Backward: the table gradient is a scatter of +-weights into dSAT followed by two reverse prefix
// sums; the gradient wrt the lookup direction and mipbias is obtained by running the SAME templated
// forward on forward-mode dual numbers (4 tangents), which keeps the wrap/clip branch structure
// identical to the forward by construction.
#include "common.hpp"
#include "dual.hpp"

#pragma clang fp contract(off)

namespace {

constexpr float PI_F = 3.14159265358979323846f;
constexpr float TWO_PI_F = 6.28318530717958647692f;
constexpr float EPS_F = 1.1920929e-07f;

struct EnvTab {
    const float* sat;   // [3][H][W], or [H][W][4] when i4 is set
    int H, W;
    bool i4;            // channel-interleaved copy: one 16-byte load per tap instead of three 4-byte loads on three planes
};

// bilinear sample of the SAT at normalised (x, y) in [-1,1] (already clipped): returns 3 channels.
// ATen vectorised CPU kernel: ix = (x+1)*((W-1)/2); w = ix-floor; e = 1-w; nw=e*s ...;
// out = fma(se_v,se, fma(sw_v,sw, fma(ne_v,ne, nw_v*nw)))
// LAYOUT: -1 = t.i4 decides at run time, 0 / 1 = planar / interleaved known at compile time (k_env_lookup_fwd: without the
// run-time branch the four corners of a box are one basic block and their sixteen taps are in flight together)
template <class T, int LAYOUT = -1>
__device__ __forceinline__ void sat_sample(const EnvTab& t, const T& x, const T& y, T (&out)[3]) {
    const T ix = (x + 1.f) * ((float)(t.W - 1) * 0.5f);
    const T iy = (y + 1.f) * ((float)(t.H - 1) * 0.5f);
    const float fx = floorf(val(ix)), fy = floorf(val(iy));
    const T w = ix - fx, n = iy - fy;
    const T e = 1.f - w, s = 1.f - n;
    const T nw = e * s, ne = w * s, sw = e * n, se = w * n;
    const int x0 = (int)fx, y0 = (int)fy;
    const bool xi0 = x0 >= 0 && x0 < t.W, xi1 = x0 + 1 >= 0 && x0 + 1 < t.W;
    const bool yi0 = y0 >= 0 && y0 < t.H, yi1 = y0 + 1 >= 0 && y0 + 1 < t.H;
    float vnw[3] = {0.f, 0.f, 0.f}, vne[3] = {0.f, 0.f, 0.f}, vsw[3] = {0.f, 0.f, 0.f}, vse[3] = {0.f, 0.f, 0.f};
    if (LAYOUT < 0 ? t.i4 : LAYOUT == 1) {
        // the four taps are loaded UNCONDITIONALLY from clamped texels and zeroed afterwards: a load under a condition is a
        // branch, and four corners x four branches made a lookup a chain of dependent memory round trips (R4: 12 full waits on
        // the vector memory counter in this kernel; a launch of any size took 11 us)
        const float4* p = reinterpret_cast<const float4*>(t.sat);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const int xa = min(max(x0, 0), t.W - 1), xb = min(max(x0 + 1, 0), t.W - 1);
        const int ya = min(max(y0, 0), t.H - 1), yb = min(max(y0 + 1, 0), t.H - 1);
        const float4 la = p[ya * t.W + xa], lb = p[ya * t.W + xb], lc = p[yb * t.W + xa], ld = p[yb * t.W + xb];
        const float4 a = (xi0 && yi0) ? la : z;
        const float4 b = (xi1 && yi0) ? lb : z;
        const float4 c = (xi0 && yi1) ? lc : z;
        const float4 d = (xi1 && yi1) ? ld : z;
        vnw[0] = a.x; vnw[1] = a.y; vnw[2] = a.z;
        vne[0] = b.x; vne[1] = b.y; vne[2] = b.z;
        vsw[0] = c.x; vsw[1] = c.y; vsw[2] = c.z;
        vse[0] = d.x; vse[1] = d.y; vse[2] = d.z;
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* p = t.sat + (int64_t)c * t.H * t.W;
            vnw[c] = (xi0 && yi0) ? p[y0 * t.W + x0] : 0.f;
            vne[c] = (xi1 && yi0) ? p[y0 * t.W + x0 + 1] : 0.f;
            vsw[c] = (xi0 && yi1) ? p[(y0 + 1) * t.W + x0] : 0.f;
            vse[c] = (xi1 && yi1) ? p[(y0 + 1) * t.W + x0 + 1] : 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if constexpr (sizeof(T) == sizeof(float)) {
            float r = vnw[c] * nw;
            r = fmaf(vne[c], ne, r);
            r = fmaf(vsw[c], sw, r);
            r = fmaf(vse[c], se, r);
            out[c] = r;
        } else {
            out[c] = nw * vnw[c] + ne * vne[c] + sw * vsw[c] + se * vse[c];
        }
    }
}

// Table adjoint.  Device-scope float atomics cost ~one L2 operation per distinct cache line touched by an instruction
// (measured on MI355X: 21 G line-ops/s whatever the table size; 8 lanes on one 32-byte run = 156 G lane-ops/s, the same
// with workgroup-scope atomics on per-XCD copies of the table).  A lane-per-lookup scatter touches 64 lines per
// instruction; the adjoint therefore runs with 8 lanes per lookup on a CHANNEL-INTERLEAVED table dSAT4 [H][W][4]
// (Scatter8Acc below).
__device__ __forceinline__ void corner_taps(int H, int W, float x, float y, int& x0, int& y0, float& w, float& n) {
    const float ix = (x + 1.f) * ((float)(W - 1) * 0.5f);
    const float iy = (y + 1.f) * ((float)(H - 1) * 0.5f);
    const float fx = floorf(ix), fy = floorf(iy);
    w = ix - fx; n = iy - fy;
    x0 = (int)fx; y0 = (int)fy;
}

template <class T>
struct Rect {   // axis-aligned box in normalised coords
    T x0, x1, y0, y1;   // left/right (bl.x, tr.x), bottom/top (bl.y, tr.y)
};

// Visitor pattern: `Acc` receives every (corner, sign) of every box that integrate_area_wrap adds.
// integrate_area (:18-39): (S(tr) + S(bl) - S(tl) - S(br)) / size with corners clipped to [-1,1].
template <class T, class Acc>
__device__ __forceinline__ void box(const Rect<T>& r, Acc& acc) {
    const T xl = d_clip(r.x0, -1.f, 1.f), xr = d_clip(r.x1, -1.f, 1.f);
    const T yb = d_clip(r.y0, -1.f, 1.f), yt = d_clip(r.y1, -1.f, 1.f);
    acc.begin();
    acc.corner(xr, yt, 0);   // tr  (+)
    acc.corner(xl, yb, 1);   // bl  (+)
    acc.corner(xl, yt, 2);   // tl  (-)
    acc.corner(xr, yb, 3);   // br  (-)
    acc.end();
}

// integrate_area_wrap_lr (:42-93)
template <class T, class Acc>
__device__ __forceinline__ void box_lr(const Rect<T>& r, Acc& acc) {
    box(r, acc);
    if (val(r.x1) > 1.f) {
        Rect<T> q = r;
        q.x0 = set_val(r.x0, -1.f);
        q.x1 = r.x1 - 2.f;
        box(q, acc);
    }
    if (val(r.x0) < -1.f) {
        Rect<T> q = r;
        q.x0 = r.x0 + 2.f;
        q.x1 = set_val(r.x1, 1.f);
        box(q, acc);
    }
}

// integrate_area_wrap (:96-173)
template <class T, class Acc>
__device__ __forceinline__ void box_wrap(const Rect<T>& r, Acc& acc) {
    box_lr(r, acc);
    if (val(r.y1) > 1.f) {            // tl.y > 1
        const float rot = val(r.x0) > 0.f ? -1.f : 1.f;          // tl.x > 0
        const T over = d_clip(r.y1 - 1.f, 0.f, 0.5f);
        Rect<T> q;
        q.x0 = r.x0 + rot; q.x1 = r.x1 + rot;
        q.y1 = set_val(r.y1, 1.f);
        q.y0 = 1.f - over;
        box_lr(q, acc);
    }
    if (val(r.y0) < -1.f) {           // bl.y < -1
        const float rot = val(r.x0) > 0.f ? -1.f : 1.f;
        const T over = d_clip(-1.f - r.y0, 0.f, 0.5f);
        Rect<T> q;
        q.x0 = r.x0 + rot; q.x1 = r.x1 + rot;
        q.y0 = set_val(r.y0, -1.f);
        q.y1 = over - 1.f;            // -1 + over
        box_lr(q, acc);
    }
}

// ---- compacted walk (round 3) ---------------------------------------------------------------------------------------
// integrate_area_wrap adds up to nine boxes per lookup: (main | top pole wrap | bottom pole wrap) x (as is | right seam
// wrap | left seam wrap).  Nearly every lookup has the main box only, but among the 64 lanes of a wave some lane needs the
// seam wraps and some lane a pole wrap almost always, so a lane-per-lookup walk executes ~5 box bodies per wave with a
// handful of active lanes in four of them.  The kernels below evaluate the MAIN box on every lane and put the extra boxes
// of a workgroup on a queue in LDS that is then worked off densely (one queue entry per lane).
// combo = 3 * v + h, v in {main, top, bottom}, h in {as is, right wrap, left wrap}; the order 0..8 is the order in which
// box_wrap / box_lr add the boxes.
template <class T>
__device__ __forceinline__ void env_combo_rect(const Rect<T>& r, int combo, Rect<T>& out) {
    const int v = combo / 3, h = combo % 3;
    Rect<T> q = r;
    if (v != 0) {
        const float rot = val(r.x0) > 0.f ? -1.f : 1.f;
        q.x0 = r.x0 + rot; q.x1 = r.x1 + rot;
        if (v == 1) {
            const T over = d_clip(r.y1 - 1.f, 0.f, 0.5f);
            q.y1 = set_val(r.y1, 1.f);
            q.y0 = 1.f - over;
        } else {
            const T over = d_clip(-1.f - r.y0, 0.f, 0.5f);
            q.y0 = set_val(r.y0, -1.f);
            q.y1 = over - 1.f;
        }
    }
    out = q;
    if (h == 1) { out.x0 = set_val(q.x0, -1.f); out.x1 = q.x1 - 2.f; }
    else if (h == 2) { out.x0 = q.x0 + 2.f; out.x1 = set_val(q.x1, 1.f); }
}
// bit c of the result: combo c is added (c = 1..8; the main box always is)
__device__ __forceinline__ uint32_t env_combo_mask(float x0, float x1, float y0, float y1) {
    uint32_t m = 0;
    if (x1 > 1.f) m |= 1u << 1;
    if (x0 < -1.f) m |= 1u << 2;
    const float rot = x0 > 0.f ? -1.f : 1.f;
    const float qx0 = x0 + rot, qx1 = x1 + rot;
    const uint32_t lr = 1u | (qx1 > 1.f ? 2u : 0u) | (qx0 < -1.f ? 4u : 0u);
    if (y1 > 1.f) m |= lr << 3;
    if (y0 < -1.f) m |= lr << 6;
    return m;
}
struct EnvQueue {
    uint32_t n;
    uint16_t item[8 * 256];          // owner thread << 4 | combo
};
__device__ __forceinline__ void env_queue_push(EnvQueue& q, uint32_t mask) {
    while (mask) {
        const int c = __ffs(mask) - 1;
        mask &= mask - 1;
        q.item[atomicAdd(&q.n, 1u)] = (uint16_t)(threadIdx.x << 4 | c);
    }
}

template <class T>
struct Geometry {
    Rect<T> rect;
    T size;
    float cy;    // latitude coordinate (pole test)
};

// sa2mip (:373-397) + forward (:409-480) up to the box corners
template <class T>
__device__ __forceinline__ Geometry<T> env_geometry(int H, int W, const T& a, const T& b, const T& c, float sa,
                                                    const T& mipbias) {
    const float h = (float)H;
    const T cosv = d_sqrt(d_clipmin(1.f - c * c, EPS_F));
    // h*w / x is int.__truediv__(tensor) == reciprocal(x) * (h*w)   (h*w is a power of two here)
    const T den = d_clipmin(cosv * 19.739208802178716f, EPS_F);   // 2*math.pi**2 as fp32
    const T d = (set_val(den, 1.f) / den) * (float)(H * W);
    const T area = d_exp(d_log(d / 2.f) + sa);
    const T hh = d_clipmin(d_sqrt(d_clipmin(area, EPS_F)) * cosv, EPS_F);
    const T ww = area / hh;
    const T mw = d_clip(d_log(ww) / LN2_F + mipbias, 0.f, 7.f);
    const T mh = d_clip(d_log(hh) / LN2_F + mipbias, 0.f, 7.f);
    const T sw = d_pow2(mw) / h / 2.f;
    const T sh = d_pow2(mh) / h;
    Geometry<T> g;
    g.size = ((sw / 2.f) * (float)W) * ((sh / 2.f) * h);
    const T norm2d = d_sqrt(a * a + b * b);
    const T phi = d_atan2(b, a);
    const T theta = d_atan2(c, norm2d);
    const T cx = (d_rem(phi, TWO_PI_F) - PI_F) / PI_F;
    const T cy = ((-theta) / PI_F) * 2.f;
    g.cy = val(cy);
    g.rect.x0 = cx - sw / 2.f; g.rect.x1 = cx + sw / 2.f;
    g.rect.y0 = cy - sh / 2.f; g.rect.y1 = cy + sh / 2.f;
    return g;
}

// ---- accumulators ----------------------------------------------------------------------------
template <class T, int LAYOUT = -1>
struct SumAcc {   // forward value: sum of boxes, each divided by the ORIGINAL size
    EnvTab tab;
    T size;
    T total[3];
    T cur[3];
    __device__ void begin() {}
    __device__ void corner(const T& x, const T& y, int k) {
        T s[3];
        sat_sample<T, LAYOUT>(tab, x, y, s);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (k == 0) cur[c] = s[c];
            else if (k == 1) cur[c] = cur[c] + s[c];
            else cur[c] = cur[c] - s[c];
        }
    }
    __device__ void end() {
#pragma unroll
        for (int c = 0; c < 3; ++c) total[c] = total[c] + cur[c] / size;
    }
};

// ---- kernels -----------------------------------------------------------------------------------
// Forward lookup: main box on every lane, extra boxes through the queue; an owner adds its boxes in the reference's order
// (box_wrap), each computed with the same arithmetic as before: the same bits as the lane-per-lookup walk.
template <int LAYOUT>
__global__ void __launch_bounds__(256) k_env_lookup_fwd(EnvTab tab, const float* __restrict__ dirs, int ld,
                                                        const float* __restrict__ sa, int64_t R, float mipbias,
                                                        const float* __restrict__ sc,
                                                        const float* __restrict__ pole_rows /*[2][3] top,bot*/,
                                                        float* __restrict__ out) {
    __shared__ EnvQueue Q;
    __shared__ float geo[256][5];            // rect + size of every lookup of the workgroup
    __shared__ float part[256][8][3];        // value of the extra boxes, by owner and combo - 1
    if (threadIdx.x == 0) Q.n = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sc) mipbias = sc[0];
    float total[3] = {0.f, 0.f, 0.f};
    uint32_t mask = 0;
    float cy = 0.f;
    if (r < R) {
        const float* q = dirs + r * ld + (ld - 3);            // ld = 6: [origin | direction] ray rows
        const Geometry<float> g = env_geometry<float>(tab.H, tab.W, q[0], q[1], q[2], sa[r], mipbias);
        cy = g.cy;
        geo[threadIdx.x][0] = g.rect.x0; geo[threadIdx.x][1] = g.rect.x1; geo[threadIdx.x][2] = g.rect.y0;
        geo[threadIdx.x][3] = g.rect.y1; geo[threadIdx.x][4] = g.size;
        SumAcc<float, LAYOUT> acc;
        acc.tab = tab;
        acc.size = g.size;
        acc.total[0] = acc.total[1] = acc.total[2] = 0.f;
        box(g.rect, acc);
        total[0] = acc.total[0]; total[1] = acc.total[1]; total[2] = acc.total[2];
        mask = env_combo_mask(g.rect.x0, g.rect.x1, g.rect.y0, g.rect.y1);
        env_queue_push(Q, mask);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < Q.n; i += blockDim.x) {
        const int owner = Q.item[i] >> 4, combo = Q.item[i] & 15;
        Rect<float> rr{geo[owner][0], geo[owner][1], geo[owner][2], geo[owner][3]}, p;
        env_combo_rect(rr, combo, p);
        SumAcc<float, LAYOUT> acc;
        acc.tab = tab;
        acc.size = geo[owner][4];
        acc.total[0] = acc.total[1] = acc.total[2] = 0.f;
        box(p, acc);
        part[owner][combo - 1][0] = acc.total[0]; part[owner][combo - 1][1] = acc.total[1]; part[owner][combo - 1][2] = acc.total[2];
    }
    __syncthreads();
    if (r >= R) return;
    while (mask) {                              // in combo order = the order of integrate_area_wrap
        const int c = __ffs(mask) - 1;
        mask &= mask - 1;
        total[0] = total[0] + part[threadIdx.x][c - 1][0];
        total[1] = total[1] + part[threadIdx.x][c - 1][1];
        total[2] = total[2] + part[threadIdx.x][c - 1][2];
    }
    const float cutoff = 1.f - 2.f / (float)tab.H * 3.f;
    float v[3] = {total[0] * 1000.f, total[1] * 1000.f, total[2] * 1000.f};
    if (cy > cutoff) { v[0] = pole_rows[3]; v[1] = pole_rows[4]; v[2] = pole_rows[5]; }
    if (cy < -cutoff) { v[0] = pole_rows[0]; v[1] = pole_rows[1]; v[2] = pole_rows[2]; }
    out[r * 3] = v[0]; out[r * 3 + 1] = v[1]; out[r * 3 + 2] = v[2];
}

// Table adjoint, 8 lanes per lookup: lane t of a group owns (texel x0 + (t >> 2), channel t & 3) of every corner of the
// lookup's boxes, so one atomic instruction covers 8 lookups with one 32-byte run each (two instructions per corner:
// rows y0, y0 + 1).  The 8 lanes walk the same boxes (the float geometry is recomputed per lane: ~200 instructions).
struct Scatter8Acc {
    float* dsat4;
    int H, W, dx, ch;
    float g;          // d_vals[ch] * 1000 / size (0 on the padding channel)
    __device__ void begin() {}
    __device__ void corner(float x, float y, int k) {
        int x0, y0;
        float w, n;
        corner_taps(H, W, x, y, x0, y0, w, n);
        const float sgn = k < 2 ? 1.f : -1.f;
        const int X = x0 + dx;
        const float v = g * sgn * (dx ? w : 1.f - w);
        if (X >= 0 && X < W && v != 0.f) {
            float* p = dsat4 + ((int64_t)y0 * W + X) * 4 + ch;
            const float v0 = v * (1.f - n), v1 = v * n;
            if (y0 >= 0 && y0 < H && v0 != 0.f) atomicAdd(p, v0);
            if (y0 + 1 >= 0 && y0 + 1 < H && v1 != 0.f) atomicAdd(p + (int64_t)W * 4, v1);
        }
    }
    __device__ void end() {}
};

// Backward of the lookup, two roles in one launch (blocks b % 9 == 0: role A, the others: role B), so that the
// instruction-bound role A overlaps the atomic-bound role B:
//   A  one lane per lookup: the forward on dual numbers (4 tangents) -> d_dirs, d_mipbias; pole rows -> d_pole
//   B  eight lanes per lookup: the table adjoint (Scatter8Acc)
// Round 1 did both in one 64-thread workgroup with the corners parked in 18.5 KB of LDS (8 waves per CU resident:
// 192 us for 242 k lookups, profiles/r02_c); neither role needs LDS now.
constexpr int ENV_BWD_THREADS = 256;
template <int LAYOUT>
__global__ void __launch_bounds__(ENV_BWD_THREADS) k_env_lookup_bwd(EnvTab tab, const float* __restrict__ dirs, int ld,
                                                                    const float* __restrict__ sa, int64_t R, float mipbias,
                                                                    const float* __restrict__ sc,
                                                                    const float* __restrict__ d_out,
                                                                    float* __restrict__ d_sat4,
                                                                    float* __restrict__ d_pole /*[2][3]*/,
                                                                    float* __restrict__ d_dirs,
                                                                    float* __restrict__ d_mipbias) {
    if (sc) mipbias = sc[0];
    const float cutoff = 1.f - 2.f / (float)tab.H * 3.f;
    const int64_t period = blockIdx.x / 9;
    const int slot = blockIdx.x % 9;
    if (slot != 0) {
        // ---- role B -------------------------------------------------------------------------------------------------
        if (!d_sat4) return;
        const int64_t r = (period * 8 + (slot - 1)) * (ENV_BWD_THREADS / 8) + (threadIdx.x >> 3);
        if (r >= R) return;
        const int t = threadIdx.x & 7;
        const float* q = dirs + r * ld + (ld - 3);
        const Geometry<float> g = env_geometry<float>(tab.H, tab.W, q[0], q[1], q[2], sa[r], mipbias);
        if (g.cy > cutoff || g.cy < -cutoff) return;          // pole rows: role A
        Scatter8Acc acc;
        acc.dsat4 = d_sat4; acc.H = tab.H; acc.W = tab.W; acc.dx = t >> 2; acc.ch = t & 3;
        acc.g = acc.ch < 3 ? d_out[r * 3 + acc.ch] * (1000.f / g.size) : 0.f;
        box_wrap(g.rect, acc);
        return;
    }
    // ---- role A -----------------------------------------------------------------------------------------------------
    const int64_t r = period * ENV_BWD_THREADS + threadIdx.x;
    float dm = 0.f;
    if (r < R) {
        const float* q = dirs + r * ld + (ld - 3);
        const float a = q[0], b = q[1], c = q[2];
        float* dq = d_dirs ? d_dirs + r * ld + (ld - 3) : nullptr;
        if (dq && ld == 6) { dq[-3] = 0.f; dq[-2] = 0.f; dq[-1] = 0.f; }          // no dependence on the ray origin
        const float go[3] = {d_out[r * 3], d_out[r * 3 + 1], d_out[r * 3 + 2]};
        typedef Dual<4> D;
        D da = mk_const<4>(a), db = mk_const<4>(b), dc = mk_const<4>(c), dmb = mk_const<4>(mipbias);
        da.d[0] = 1.f; db.d[1] = 1.f; dc.d[2] = 1.f; dmb.d[3] = 1.f;
        const Geometry<float> g = env_geometry<float>(tab.H, tab.W, a, b, c, sa[r], mipbias);     // the branch role B takes
        const bool bot = g.cy > cutoff, top = g.cy < -cutoff;
        if (top || bot) {
            // value = mean of a pole row of the activated map: no dependence on dirs / mipbias
            float* qp = d_pole + (top ? 0 : 3);
            atomicAdd(qp, go[0]); atomicAdd(qp + 1, go[1]); atomicAdd(qp + 2, go[2]);
            if (dq) { dq[0] = 0.f; dq[1] = 0.f; dq[2] = 0.f; }
        } else if (d_dirs || d_mipbias) {
            const Geometry<D> gd = env_geometry<D>(tab.H, tab.W, da, db, dc, sa[r], dmb);
            SumAcc<D, LAYOUT> acc;
            acc.tab = tab;
            acc.size = gd.size;
            acc.total[0] = acc.total[1] = acc.total[2] = mk_const<4>(0.f);
            box_wrap(gd.rect, acc);
            float gsum[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                gsum[i] = 1000.f * (go[0] * acc.total[0].d[i] + go[1] * acc.total[1].d[i] + go[2] * acc.total[2].d[i]);
            if (dq) { dq[0] = gsum[0]; dq[1] = gsum[1]; dq[2] = gsum[2]; }
            dm = gsum[3];
        }
    }
    if (d_mipbias) {
        // float atomics onto ONE address retire one after the other (13 ns each, tools/ub/same_addr_atomic.hip: a wave's worth per
        // lookup block of a 0.24 M launch = 50 us): one per workgroup
        __shared__ float s_dm[ENV_BWD_THREADS / 64];
        for (int d = 32; d > 0; d >>= 1) dm += __shfl_down(dm, d, 64);
        if (lane_id() == 0) s_dm[threadIdx.x >> 6] = dm;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < ENV_BWD_THREADS / 64; ++w) t += s_dm[w];
            if (t != 0.f) atomicAdd(d_mipbias, t);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Binned table adjoint (round 3).  The scatter above sits on the rate of the memory-side float atomics (156 G lane-ops/s,
// tools/ub/atom2.hip) and LDS float atomics are no faster (ds_add_f32: 200 G lane-ops/s chip-wide, tools/ub/lds_atom.hip),
// but INTEGER LDS atomics are: ds_add_u64 1.1 - 1.8 T lane-ops/s (tools/ub/lds_atom_int.hip).  So:
//   1  count    lane per lookup walks its boxes and counts corners per SAT tile (32 x 64 texels; LDS histogram, one
//               global add per (workgroup, tile)); also the largest |d_out| * 1000 / size of the launch (fixed-point scale)
//   2  scatter  the same walk writes one 24-byte record per corner into the tile's slice of a record pool
//               (exclusive scan of the tile counts in every workgroup, ranges reserved per (workgroup, tile))
//   3  accum    work items of <= ENV_ITEM records of one tile: the 2 x 2 x 3 taps of every record are added with
//               ds_add_u64 into a (33 x 65 x 3) window of 2^e-scaled 64-bit fixed-point accumulators (e from the launch's
//               largest contribution: >= 2^-49 of it is resolved, fp32 atomics resolve 2^-24 of the running sum, and integer
//               sums do not depend on the order), the non-zero entries are flushed to dSAT with one float atomic each.
// The dual-number role (d_dirs, d_mipbias, pole rows) is instruction-bound and independent of all three: its workgroups
// are appended to the three launches in shares, so that they fill the CUs next to the LDS / memory-bound passes.
// The texel values enter the dual numbers contracted with d_out first (one dual number per tap instead of three).
constexpr int ENV_TILE_H = 32, ENV_TILE_W = 64, ENV_MAX_TILES = 1024, ENV_ITEM = 4096;
constexpr int ENV_WIN_W = ENV_TILE_W + 1, ENV_WIN_H = ENV_TILE_H + 1, ENV_WIN = ENV_WIN_W * ENV_WIN_H * 3;

struct EnvBinHeader {
    uint32_t counts[ENV_MAX_TILES];     // corners per tile
    uint32_t cursor[ENV_MAX_TILES];     // (unused since pass 1 hands out the ranges: kept so that the words behind it stay where they were)
    uint32_t gmax_bits;                 // bits of the largest |d_out[c]| * 1000 / size (a non-negative float)
    uint32_t overflow;                  // corners that did not fit the pool (they took the direct float atomics)
    uint32_t pad[2];
    float mip_slots[64];                // d_mipbias of the dual-number workgroups, spread over 64 addresses (pass 3 adds them up)
    uint32_t gmax_slots[64];            // gmax_bits spread the same way (one atomicMax per workgroup of pass 1; pass 3 takes the max)
};
struct CornerRec {
    float w, n, g[3];                   // bilinear fractions, signed contribution per channel
    uint32_t xy;                        // (y0 - tile y) << 8 | (x0 - tile x)
};

struct EnvBwdArgs {
    EnvTab tab;
    const float* dirs; int ld;
    const float* sa; int64_t R; float mipbias;
    const float* sc;
    const float* d_out;
    float* d_sat4; float* d_pole; float* d_dirs; float* d_mipbias;
    EnvBinHeader* hdr; CornerRec* recs; int64_t cap;
    int ntx, nt;
    uint32_t* blockres;                 // [lookup workgroups][nt]: where in its tile's range a workgroup's records start (pass 1 -> pass 2)
};

// where the dual-number role leaves its mip-bias adjoint: one atomic per wave on d_mipbias / in the 64 slots of the header (riders of
// passes 1 and 2) / one atomic per workgroup on d_mipbias
constexpr int ENV_MIP_DIRECT = 0, ENV_MIP_SLOTS = 1, ENV_MIP_WORKGROUP = 2;

// dual-number role with the channels contracted: q(x, y) = sum_c go[c] * S_c(x, y)
template <int LAYOUT>
struct SumAccQ {
    EnvTab tab;
    float go[3];
    Dual<4> size, total, cur;
    __device__ void begin() {}
    __device__ void corner(const Dual<4>& x, const Dual<4>& y, int k) {
        typedef Dual<4> D;
        const D ix = (x + 1.f) * ((float)(tab.W - 1) * 0.5f);
        const D iy = (y + 1.f) * ((float)(tab.H - 1) * 0.5f);
        const float fx = floorf(ix.v), fy = floorf(iy.v);
        const D w = ix - fx, n = iy - fy;
        const D e = 1.f - w, s = 1.f - n;
        const int x0 = (int)fx, y0 = (int)fy;
        const bool xi0 = x0 >= 0 && x0 < tab.W, xi1 = x0 + 1 >= 0 && x0 + 1 < tab.W;
        const bool yi0 = y0 >= 0 && y0 < tab.H, yi1 = y0 + 1 >= 0 && y0 + 1 < tab.H;
        float q[4] = {0.f, 0.f, 0.f, 0.f};
        if (LAYOUT < 0 ? tab.i4 : LAYOUT == 1) {
            const float4* p = reinterpret_cast<const float4*>(tab.sat);
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const int xa = min(max(x0, 0), tab.W - 1), xb = min(max(x0 + 1, 0), tab.W - 1);      // (unconditional loads: sat_sample)
            const int ya = min(max(y0, 0), tab.H - 1), yb = min(max(y0 + 1, 0), tab.H - 1);
            const float4 la = p[ya * tab.W + xa], lb = p[ya * tab.W + xb], lc = p[yb * tab.W + xa], ld = p[yb * tab.W + xb];
            const float4 a = (xi0 && yi0) ? la : z;
            const float4 b = (xi1 && yi0) ? lb : z;
            const float4 c = (xi0 && yi1) ? lc : z;
            const float4 d = (xi1 && yi1) ? ld : z;
            q[0] = go[0] * a.x + go[1] * a.y + go[2] * a.z;
            q[1] = go[0] * b.x + go[1] * b.y + go[2] * b.z;
            q[2] = go[0] * c.x + go[1] * c.y + go[2] * c.z;
            q[3] = go[0] * d.x + go[1] * d.y + go[2] * d.z;
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* p = tab.sat + (int64_t)c * tab.H * tab.W;
                q[0] += go[c] * ((xi0 && yi0) ? p[y0 * tab.W + x0] : 0.f);
                q[1] += go[c] * ((xi1 && yi0) ? p[y0 * tab.W + x0 + 1] : 0.f);
                q[2] += go[c] * ((xi0 && yi1) ? p[(y0 + 1) * tab.W + x0] : 0.f);
                q[3] += go[c] * ((xi1 && yi1) ? p[(y0 + 1) * tab.W + x0 + 1] : 0.f);
            }
        }
        const D v = (e * s) * q[0] + (w * s) * q[1] + (e * n) * q[2] + (w * n) * q[3];
        if (k == 0) cur = v;
        else if (k == 1) cur = cur + v;
        else cur = cur - v;
    }
    __device__ void end() { total = total + cur / size; }
};

// LDS of the dual-number role: queue | rect + size as dual numbers [256][25] | d_out [256][3] | extra-box tangents [256][4]
constexpr int ENV_DIRS_LDS = (int)sizeof(EnvQueue) + 8 + 256 * (25 + 3 + 4) * 4;

template <int LAYOUT>
__device__ __forceinline__ void env_role_dirs(const EnvBwdArgs& A, int64_t block, unsigned char* smem, int mip_mode) {
    typedef Dual<4> D;
    EnvQueue& Q = *reinterpret_cast<EnvQueue*>(smem);
    float (*geo)[25] = reinterpret_cast<float (*)[25]>(smem + ((sizeof(EnvQueue) + 15) & ~15));
    float (*gos)[3] = reinterpret_cast<float (*)[3]>(geo + 256);
    float (*extra)[4] = reinterpret_cast<float (*)[4]>(gos + 256);
    if (threadIdx.x == 0) Q.n = 0;
    extra[threadIdx.x][0] = extra[threadIdx.x][1] = extra[threadIdx.x][2] = extra[threadIdx.x][3] = 0.f;
    __syncthreads();
    const float mipbias = A.sc ? A.sc[0] : A.mipbias;
    const float cutoff = 1.f - 2.f / (float)A.tab.H * 3.f;
    const int64_t r = block * ENV_BWD_THREADS + threadIdx.x;
    float* dq = nullptr;
    D total = mk_const<4>(0.f);
    bool walked = false;
    if (r < A.R) {
        const float* q = A.dirs + r * A.ld + (A.ld - 3);
        const float a = q[0], b = q[1], c = q[2];
        dq = A.d_dirs ? A.d_dirs + r * A.ld + (A.ld - 3) : nullptr;
        if (dq && A.ld == 6) { dq[-3] = 0.f; dq[-2] = 0.f; dq[-1] = 0.f; }
        const float go[3] = {A.d_out[r * 3], A.d_out[r * 3 + 1], A.d_out[r * 3 + 2]};
        const Geometry<float> g = env_geometry<float>(A.tab.H, A.tab.W, a, b, c, A.sa[r], mipbias);
        const bool bot = g.cy > cutoff, top = g.cy < -cutoff;
        if (top || bot) {
            float* qp = A.d_pole + (top ? 0 : 3);
            atomicAdd(qp, go[0]); atomicAdd(qp + 1, go[1]); atomicAdd(qp + 2, go[2]);
        } else if (A.d_dirs || A.d_mipbias) {
            D da = mk_const<4>(a), db = mk_const<4>(b), dc = mk_const<4>(c), dmb = mk_const<4>(mipbias);
            da.d[0] = 1.f; db.d[1] = 1.f; dc.d[2] = 1.f; dmb.d[3] = 1.f;
            const Geometry<D> gd = env_geometry<D>(A.tab.H, A.tab.W, da, db, dc, A.sa[r], dmb);
            const D* src[5] = {&gd.rect.x0, &gd.rect.x1, &gd.rect.y0, &gd.rect.y1, &gd.size};
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                geo[threadIdx.x][5 * k] = src[k]->v;
#pragma unroll
                for (int i = 0; i < 4; ++i) geo[threadIdx.x][5 * k + 1 + i] = src[k]->d[i];
            }
            gos[threadIdx.x][0] = go[0]; gos[threadIdx.x][1] = go[1]; gos[threadIdx.x][2] = go[2];
            SumAccQ<LAYOUT> acc;
            acc.tab = A.tab;
            acc.go[0] = go[0]; acc.go[1] = go[1]; acc.go[2] = go[2];
            acc.size = gd.size;
            acc.total = mk_const<4>(0.f);
            acc.cur = mk_const<4>(0.f);
            box(gd.rect, acc);
            total = acc.total;
            walked = true;
            env_queue_push(Q, env_combo_mask(gd.rect.x0.v, gd.rect.x1.v, gd.rect.y0.v, gd.rect.y1.v));
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < Q.n; i += ENV_BWD_THREADS) {
        const int owner = Q.item[i] >> 4, combo = Q.item[i] & 15;
        D v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            v[k].v = geo[owner][5 * k];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[k].d[j] = geo[owner][5 * k + 1 + j];
        }
        Rect<D> rr{v[0], v[1], v[2], v[3]}, p;
        env_combo_rect(rr, combo, p);
        SumAccQ<LAYOUT> acc;
        acc.tab = A.tab;
        acc.go[0] = gos[owner][0]; acc.go[1] = gos[owner][1]; acc.go[2] = gos[owner][2];
        acc.size = v[4];
        acc.total = mk_const<4>(0.f);
        acc.cur = mk_const<4>(0.f);
        box(p, acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&extra[owner][j], acc.total.d[j]);
    }
    __syncthreads();
    float dm = 0.f;
    if (r < A.R) {
        if (walked) {
            if (dq) {
                dq[0] = 1000.f * (total.d[0] + extra[threadIdx.x][0]);
                dq[1] = 1000.f * (total.d[1] + extra[threadIdx.x][1]);
                dq[2] = 1000.f * (total.d[2] + extra[threadIdx.x][2]);
            }
            dm = 1000.f * (total.d[3] + extra[threadIdx.x][3]);
        } else if (dq) {
            dq[0] = 0.f; dq[1] = 0.f; dq[2] = 0.f;
        }
    }
    if (A.d_mipbias) {
        // one float atomic per wave onto the single address of d_mipbias would retire one after the other (13 ns each: 50 us for
        // the 3.8 k waves of a 0.24 M-lookup launch, all of what this role used to cost next to the passes): spread over 64
        // slots of the header here, summed by pass 3 (riders of pass 3 itself -- none with today's shares -- go direct)
        for (int d = 32; d > 0; d >>= 1) dm += __shfl_down(dm, d, 64);
        if (mip_mode == ENV_MIP_WORKGROUP) {      // the role as a launch of its own (k_env_dirs): one atomic per workgroup
            __shared__ float s_dm[ENV_BWD_THREADS / 64];
            if (lane_id() == 0) s_dm[threadIdx.x >> 6] = dm;
            __syncthreads();
            if (threadIdx.x == 0) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < ENV_BWD_THREADS / 64; ++w) t += s_dm[w];
                if (t != 0.f) atomicAdd(A.d_mipbias, t);
            }
        } else if (lane_id() == 0 && dm != 0.f) {
            if (mip_mode == ENV_MIP_SLOTS) atomicAdd(&A.hdr->mip_slots[(block * (ENV_BWD_THREADS / 64) + (threadIdx.x >> 6)) & 63], dm);
            else atomicAdd(A.d_mipbias, dm);
        }
    }
}

// Role of a workgroup when D dual-number workgroups ride along with O workgroups of a pass: every k-th one (k = (O + D) / D),
// so that both kinds are resident from the start of the launch (appended behind the pass they would only run after it).
// -> true: dual-number role, idx = its block; false: idx = block of the pass
__device__ __forceinline__ bool env_role(int own_blocks, int dirs_blocks, int64_t& idx) {
    const int b = blockIdx.x;
    if (dirs_blocks <= 0) { idx = b; return false; }
    const int k = (own_blocks + dirs_blocks) / dirs_blocks;
    const int m = b / k;
    if (b % k == k - 1 && m < dirs_blocks) { idx = m; return true; }
    idx = b - min(dirs_blocks, m);
    return false;
}

__device__ __forceinline__ int env_tile_of(int x0, int y0, int ntx) { return (y0 / ENV_TILE_H) * ntx + (x0 / ENV_TILE_W); }

struct CountAcc {
    uint32_t* hist; int H, W, ntx;
    __device__ void begin() {}
    __device__ void corner(float x, float y, int) {
        int x0, y0; float w, n;
        corner_taps(H, W, x, y, x0, y0, w, n);
        atomicAdd(&hist[env_tile_of(x0, y0, ntx)], 1u);
    }
    __device__ void end() {}
};

// the float walk of passes 1 and 2: geometry of this thread's lookup into LDS, main box on `acc`, extra boxes queued
// -> live (a lookup off the pole rows)
template <class Acc>
__device__ __forceinline__ bool env_walk_main(const EnvBwdArgs& A, int64_t r, EnvQueue& Q, float (*geo)[5], Acc& acc, float& inv_size) {
    const float mipbias = A.sc ? A.sc[0] : A.mipbias;
    const float cutoff = 1.f - 2.f / (float)A.tab.H * 3.f;
    if (r >= A.R) return false;
    const float* q = A.dirs + r * A.ld + (A.ld - 3);
    const Geometry<float> g = env_geometry<float>(A.tab.H, A.tab.W, q[0], q[1], q[2], A.sa[r], mipbias);
    if (g.cy > cutoff || g.cy < -cutoff) return false;
    geo[threadIdx.x][0] = g.rect.x0; geo[threadIdx.x][1] = g.rect.x1; geo[threadIdx.x][2] = g.rect.y0;
    geo[threadIdx.x][3] = g.rect.y1; geo[threadIdx.x][4] = g.size;
    inv_size = 1000.f / g.size;
    box(g.rect, acc);
    env_queue_push(Q, env_combo_mask(g.rect.x0, g.rect.x1, g.rect.y0, g.rect.y1));
    return true;
}

// the dual-number role as a launch of its own (nmf_sat_lookup_bwd_dirs): a caller whose dependency chain needs d_dirs only runs this
// on the chain and the table role (the three passes, no riders) on another stream
template <int LAYOUT>
__global__ void __launch_bounds__(ENV_BWD_THREADS) k_env_dirs(EnvBwdArgs A) {
    __shared__ __align__(16) unsigned char smem[ENV_DIRS_LDS];
    env_role_dirs<LAYOUT>(A, blockIdx.x, smem, ENV_MIP_WORKGROUP);
}

// pass 1
template <int LAYOUT>
__global__ void __launch_bounds__(ENV_BWD_THREADS) k_env_bin_count(EnvBwdArgs A, int own_blocks, int dirs_blocks, int64_t dirs_block0) {
    __shared__ __align__(16) unsigned char smem[ENV_DIRS_LDS];
    int64_t blk;
    if (env_role(own_blocks, dirs_blocks, blk)) { env_role_dirs<LAYOUT>(A, dirs_block0 + blk, smem, ENV_MIP_SLOTS); return; }
    EnvQueue& Q = *reinterpret_cast<EnvQueue*>(smem);
    float (*geo)[5] = reinterpret_cast<float (*)[5]>(smem + ((sizeof(EnvQueue) + 15) & ~15));
    uint32_t* hist = reinterpret_cast<uint32_t*>(geo + 256);
    uint32_t* gmax_s = hist + ENV_MAX_TILES;
    for (int i = threadIdx.x; i < A.nt; i += ENV_BWD_THREADS) hist[i] = 0;
    if (threadIdx.x == 0) { *gmax_s = 0; Q.n = 0; }
    __syncthreads();
    const int64_t r = blk * ENV_BWD_THREADS + threadIdx.x;
    CountAcc acc{hist, A.tab.H, A.tab.W, A.ntx};
    float inv_size = 0.f;
    if (env_walk_main(A, r, Q, geo, acc, inv_size)) {
        const float g0 = A.d_out[r * 3], g1 = A.d_out[r * 3 + 1], g2 = A.d_out[r * 3 + 2];
        const float m = fmaxf(fmaxf(fabsf(g0), fabsf(g1)), fabsf(g2)) * inv_size;
        // a NaN / infinite adjoint (fmaxf drops NaNs): its bit pattern, above every finite float's, tells pass 3 to poison the
        // table gradient
        if (!(fabsf(g0) <= 3.0e38f) || !(fabsf(g1) <= 3.0e38f) || !(fabsf(g2) <= 3.0e38f) || !(m <= 3.0e38f))
            atomicMax(gmax_s, 0x7fc00000u);
        else if (m > 0.f) atomicMax(gmax_s, __float_as_uint(m));
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < Q.n; i += ENV_BWD_THREADS) {
        const int owner = Q.item[i] >> 4, combo = Q.item[i] & 15;
        Rect<float> rr{geo[owner][0], geo[owner][1], geo[owner][2], geo[owner][3]}, p;
        env_combo_rect(rr, combo, p);
        box(p, acc);
    }
    __syncthreads();
    // counting IS reserving: what the counter held before this workgroup's corners is where its records start inside the tile's
    // range (pass 2 used to walk every footprint twice, once to find this out)
    for (int i = threadIdx.x; i < A.nt; i += ENV_BWD_THREADS)
        if (hist[i]) A.blockres[blk * A.nt + i] = atomicAdd(&A.hdr->counts[i], hist[i]);
    if (threadIdx.x == 0 && *gmax_s) atomicMax(&A.hdr->gmax_slots[blockIdx.x & 63], *gmax_s);
}

// exclusive scans over the tiles (every workgroup of passes 2 and 3 does them for itself): base[i] = first record of tile i,
// base[nt] = all records; items[i] = first work item of tile i when a tile has ceil(count / ENV_ITEM) of them (may be null)
__device__ __forceinline__ uint32_t env_block_excl(uint32_t sum, uint32_t* tmp /*[8] LDS*/) {
    uint32_t incl = sum;
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    __syncthreads();
    if (lane == 63) tmp[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += tmp[w];
    return off + incl - sum;
}
__device__ __forceinline__ void env_scan_counts(const EnvBinHeader* hdr, int nt, uint32_t* base, uint32_t* items, uint32_t* tmp) {
    const int per = (nt + ENV_BWD_THREADS - 1) / ENV_BWD_THREADS;          // <= 4 consecutive tiles per thread
    uint32_t loc[4] = {0, 0, 0, 0}, sum = 0, isum = 0;
    for (int j = 0; j < per; ++j) {
        const int i = threadIdx.x * per + j;
        loc[j] = i < nt ? hdr->counts[i] : 0;
        sum += loc[j];
        isum += (loc[j] + ENV_ITEM - 1) / ENV_ITEM;
    }
    uint32_t run = env_block_excl(sum, tmp);
    for (int j = 0; j < per; ++j) {
        const int i = threadIdx.x * per + j;
        if (i < nt) base[i] = run;
        run += loc[j];
    }
    if (threadIdx.x == ENV_BWD_THREADS - 1) base[nt] = run;
    if (items) {
        uint32_t irun = env_block_excl(isum, tmp);
        for (int j = 0; j < per; ++j) {
            const int i = threadIdx.x * per + j;
            if (i < nt) items[i] = irun;
            irun += (loc[j] + ENV_ITEM - 1) / ENV_ITEM;
        }
        if (threadIdx.x == ENV_BWD_THREADS - 1) items[nt] = irun;
    }
    __syncthreads();
}

struct EmitAcc {
    uint32_t* rank;         // LDS: running rank of this workgroup inside each tile
    const uint32_t* resv;   // LDS: first record of this workgroup's range in each tile
    CornerRec* recs; int64_t cap;
    float* dsat4; uint32_t* overflow;
    int H, W, ntx;
    float g[3];             // d_out[c] * 1000 / size
    __device__ void begin() {}
    __device__ void corner(float x, float y, int k) {
        int x0, y0; float w, n;
        corner_taps(H, W, x, y, x0, y0, w, n);
        const int t = env_tile_of(x0, y0, ntx);
        const int64_t pos = (int64_t)resv[t] + atomicAdd(&rank[t], 1u);
        const float sgn = k < 2 ? 1.f : -1.f;
        if (pos < cap) {
            CornerRec rec;
            rec.w = w; rec.n = n;
            rec.g[0] = g[0] * sgn; rec.g[1] = g[1] * sgn; rec.g[2] = g[2] * sgn;
            rec.xy = (uint32_t)((y0 % ENV_TILE_H) << 8 | (x0 % ENV_TILE_W));
            recs[pos] = rec;
        } else {            // pool exhausted: the direct float atomics (correct, slower)
            if (overflow) atomicAdd(overflow, 1u);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int X = x0 + dx, Y = y0 + dy;
                    if (X < W && Y < H) {
                        const float wt = (dx ? w : 1.f - w) * (dy ? n : 1.f - n) * sgn;
                        float* p = dsat4 + ((int64_t)Y * W + X) * 4;
                        atomicAdd(p, g[0] * wt); atomicAdd(p + 1, g[1] * wt); atomicAdd(p + 2, g[2] * wt);
                    }
                }
        }
    }
    __device__ void end() {}
};

// pass 2
constexpr int ENV_SCATTER_LDS = (int)sizeof(EnvQueue) + 8 + 256 * (5 + 3) * 4 + (3 * ENV_MAX_TILES + 1 + 8) * 4;
template <int LAYOUT>
__global__ void __launch_bounds__(ENV_BWD_THREADS) k_env_bin_scatter(EnvBwdArgs A, int own_blocks, int dirs_blocks, int64_t dirs_block0) {
    __shared__ __align__(16) unsigned char smem[ENV_DIRS_LDS > ENV_SCATTER_LDS ? ENV_DIRS_LDS : ENV_SCATTER_LDS];
    int64_t blk;
    if (env_role(own_blocks, dirs_blocks, blk)) { env_role_dirs<LAYOUT>(A, dirs_block0 + blk, smem, ENV_MIP_SLOTS); return; }
    EnvQueue& Q = *reinterpret_cast<EnvQueue*>(smem);
    float (*geo)[5] = reinterpret_cast<float (*)[5]>(smem + ((sizeof(EnvQueue) + 15) & ~15));
    float (*gs)[3] = reinterpret_cast<float (*)[3]>(geo + 256);
    uint32_t* base = reinterpret_cast<uint32_t*>(gs + 256);          // [nt + 1]
    uint32_t* hist = base + ENV_MAX_TILES + 1;
    uint32_t* resv = hist + ENV_MAX_TILES;
    uint32_t* tmp = resv + ENV_MAX_TILES;
    for (int i = threadIdx.x; i < A.nt; i += ENV_BWD_THREADS) hist[i] = 0;
    if (threadIdx.x == 0) Q.n = 0;
    const uint32_t* mine = A.blockres + blk * A.nt;          // (only the tiles this workgroup has corners in were written, and only
    uint32_t own[ENV_MAX_TILES / ENV_BWD_THREADS];           //  those are looked up below)
#pragma unroll
    for (int j = 0; j < ENV_MAX_TILES / ENV_BWD_THREADS; ++j) {
        const int i = threadIdx.x + j * ENV_BWD_THREADS;
        own[j] = i < A.nt ? mine[i] : 0;
    }
    env_scan_counts(A.hdr, A.nt, base, nullptr, tmp);
#pragma unroll
    for (int j = 0; j < ENV_MAX_TILES / ENV_BWD_THREADS; ++j) {
        const int i = threadIdx.x + j * ENV_BWD_THREADS;
        if (i < A.nt) resv[i] = base[i] + own[j];
    }
    __syncthreads();
    // one walk: the records (same lookups per workgroup as in pass 1, whose counters handed out the ranges)
    const int64_t r = blk * ENV_BWD_THREADS + threadIdx.x;
    EmitAcc acc;
    acc.rank = hist; acc.resv = resv; acc.recs = A.recs; acc.cap = A.cap; acc.dsat4 = A.d_sat4; acc.overflow = &A.hdr->overflow;
    acc.H = A.tab.H; acc.W = A.tab.W; acc.ntx = A.ntx;
    {
        const float mipbias = A.sc ? A.sc[0] : A.mipbias;
        const float cutoff = 1.f - 2.f / (float)A.tab.H * 3.f;
        if (r < A.R) {
            const float* q = A.dirs + r * A.ld + (A.ld - 3);
            const float go0 = A.d_out[r * 3], go1 = A.d_out[r * 3 + 1], go2 = A.d_out[r * 3 + 2];
            const Geometry<float> g = env_geometry<float>(A.tab.H, A.tab.W, q[0], q[1], q[2], A.sa[r], mipbias);
            if (!(g.cy > cutoff || g.cy < -cutoff)) {
                geo[threadIdx.x][0] = g.rect.x0; geo[threadIdx.x][1] = g.rect.x1; geo[threadIdx.x][2] = g.rect.y0;
                geo[threadIdx.x][3] = g.rect.y1; geo[threadIdx.x][4] = g.size;
                const float inv_size = 1000.f / g.size;
                acc.g[0] = go0 * inv_size; acc.g[1] = go1 * inv_size; acc.g[2] = go2 * inv_size;
                gs[threadIdx.x][0] = acc.g[0]; gs[threadIdx.x][1] = acc.g[1]; gs[threadIdx.x][2] = acc.g[2];
                box(g.rect, acc);
                env_queue_push(Q, env_combo_mask(g.rect.x0, g.rect.x1, g.rect.y0, g.rect.y1));
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < Q.n; i += ENV_BWD_THREADS) {
        const int owner = Q.item[i] >> 4, combo = Q.item[i] & 15;
        Rect<float> rr{geo[owner][0], geo[owner][1], geo[owner][2], geo[owner][3]}, p;
        env_combo_rect(rr, combo, p);
        acc.g[0] = gs[owner][0]; acc.g[1] = gs[owner][1]; acc.g[2] = gs[owner][2];
        box(p, acc);
    }
}

// pass 3
template <int LAYOUT>
__global__ void __launch_bounds__(ENV_BWD_THREADS) k_env_bin_accum(EnvBwdArgs A, int own_blocks, int dirs_blocks, int64_t dirs_block0) {
    constexpr int OWN_LDS = ENV_WIN * 8 + (2 * ENV_MAX_TILES + 2 + 8) * 4;
    __shared__ __align__(16) unsigned char smem[ENV_DIRS_LDS > OWN_LDS ? ENV_DIRS_LDS : OWN_LDS];
    int64_t blk;
    if (env_role(own_blocks, dirs_blocks, blk)) { env_role_dirs<LAYOUT>(A, dirs_block0 + blk, smem, ENV_MIP_DIRECT); return; }
    if (blk == 0 && A.d_mipbias && threadIdx.x < 64) {      // the mip-bias adjoint the riders of passes 1 and 2 left in 64 slots
        float v = A.hdr->mip_slots[threadIdx.x];
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
        if (threadIdx.x == 0 && v != 0.f) atomicAdd(A.d_mipbias, v);
    }
    unsigned long long* win = reinterpret_cast<unsigned long long*>(smem);
    uint32_t* base = reinterpret_cast<uint32_t*>(win + ENV_WIN);
    uint32_t* items = base + ENV_MAX_TILES + 1;
    uint32_t* tmp = items + ENV_MAX_TILES + 1;
    env_scan_counts(A.hdr, A.nt, base, items, tmp);
    const uint32_t n_items = items[A.nt];
    uint32_t gbits = 0;                                      // (every lane reads all 64 slots: the same cache line, no shuffle)
#pragma unroll 8
    for (int i = 0; i < 64; ++i) gbits = max(gbits, A.hdr->gmax_slots[i]);
    const float gmax = __uint_as_float(gbits);
    if (!(gmax <= 3.0e38f)) {
        // a NaN / infinite record adjoint (its bit pattern wins the unsigned maximum): the fixed-point accumulation cannot carry
        // it, so it goes into the table gradient as it is -- the reverse prefix sums spread it and the optimizer's per-element
        // isfinite test then sees it, as on the direct (float atomic) path
        if (blk == 0 && threadIdx.x == 0) atomicAdd(A.d_sat4, gmax != gmax ? gmax : gmax - gmax);
        return;
    }
    if (n_items == 0 || !(gmax > 0.f)) return;
    // |sum| <= ENV_ITEM * gmax < 2^(12 + ilogb(gmax) + 1): scaled by 2^e it stays below 2^62
    const int e = 49 - ilogbf(gmax);
    const double scale = ldexp(1.0, e), inv_scale = ldexp(1.0, -e);
    for (uint32_t item = (uint32_t)blk; item < n_items; item += own_blocks) {
        int lo = 0, hi = A.nt - 1;                       // the tile whose item range holds `item`
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (items[mid] <= item) lo = mid; else hi = mid - 1;
        }
        // (the LAST tile with items[tile] <= item: tiles without records share their prefix value with the next one)
        const int tile = lo;
        const uint32_t first = base[tile] + (item - items[tile]) * ENV_ITEM;
        const uint32_t last = min(first + (uint32_t)ENV_ITEM, base[tile + 1]);
        for (int i = threadIdx.x; i < ENV_WIN; i += ENV_BWD_THREADS) win[i] = 0ull;
        __syncthreads();
        for (uint32_t i = first + threadIdx.x; i < last; i += ENV_BWD_THREADS) {
            if ((int64_t)i >= A.cap) break;               // (went the direct way in pass 2)
            const CornerRec rec = A.recs[i];
            const int lx = rec.xy & 255, ly = rec.xy >> 8;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float wt = (dx ? rec.w : 1.f - rec.w) * (dy ? rec.n : 1.f - rec.n);
                    if (wt != 0.f) {
                        unsigned long long* p = win + ((ly + dy) * ENV_WIN_W + lx + dx) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            atomicAdd(p + c, (unsigned long long)__double2ll_rn((double)(rec.g[c] * wt) * scale));
                    }
                }
        }
        __syncthreads();
        const int ty = tile / A.ntx, tx = tile % A.ntx;
        // flush: a lane per (texel, channel slot), so that 64 lanes cover 16 consecutive texels = two 128-byte lines
        for (int i = threadIdx.x; i < ENV_WIN_W * ENV_WIN_H * 4; i += ENV_BWD_THREADS) {
            const int c = i & 3, t = i >> 2;
            const int ly = t / ENV_WIN_W, lx = t % ENV_WIN_W;
            const int Y = ty * ENV_TILE_H + ly, X = tx * ENV_TILE_W + lx;
            if (c < 3 && Y < A.tab.H && X < A.tab.W) {
                const long long v = (long long)win[t * 3 + c];
                if (v != 0) atomicAdd(A.d_sat4 + ((int64_t)Y * A.tab.W + X) * 4 + c, (float)((double)v * inv_scale));
            }
        }
        __syncthreads();
    }
}

// activation + prefix sum down H: one WAVE per (channel, column), float64 wave scan carried over the 64-row chunks
// (a lane-per-column walk is a 512-deep dependent chain on only 3072 lanes: 0.2-0.5 ms; the map is L2-resident, so
// the stride-W accesses of this layout are cheap)
__global__ void __launch_bounds__(256) k_sat_cols(const float* __restrict__ bg, int H, int W, float brightness,
                                                  float mul, const float* __restrict__ sc, float* __restrict__ act,
                                                  float* __restrict__ sat) {
    if (sc) { brightness = sc[1]; mul = sc[2]; }
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= 3 * W) return;
    const int c = col / W, x = col % W, lane = lane_id();
    double carry = 0.0;
    for (int y0 = 0; y0 < H; y0 += 64) {
        const int y = y0 + lane;
        double v = 0.0;
        int64_t i = 0;
        if (y < H) {
            i = ((int64_t)c * H + y) * W + x;
            const float a = expf(fminf(brightness + mul * bg[i], 20.f));     // activation_fn, :263-273
            act[i] = a;
            v = (double)(a / 1000.f);                                        // :432-433
        }
        const double incl = wave_incl_scan(v);
        if (y < H) sat[i] = (float)(carry + incl);
        carry += __shfl(incl, 63, 64);
    }
}

// prefix sum across W, in place: one wave per (channel, row)
__global__ void __launch_bounds__(256) k_sat_rows(float* __restrict__ sat, int H, int W, float* __restrict__ sat_i4) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= 3 * H) return;
    const int lane = lane_id();
    float* p = sat + (int64_t)row * W;
    float* q = sat_i4 ? sat_i4 + (int64_t)(row % H) * W * 4 + row / H : nullptr;
    double carry = 0.0;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const double v = x < W ? (double)p[x] : 0.0;
        const double incl = wave_incl_scan(v);
        if (x < W) {
            const float v32 = (float)(carry + incl);
            p[x] = v32;
            if (q) q[(int64_t)x * 4] = v32;
        }
        carry += __shfl(incl, 63, 64);
    }
}

// backward of the build: d_act = reverse-cumsum_H(reverse-cumsum_W(dSAT)) / 1000 (+ pole-row means);
// d_bg = d_act * act * mul where the exp argument is not clipped.
// dsat4 is the channel-interleaved adjoint table [H][W][4] filled by k_env_lookup_bwd
__global__ void __launch_bounds__(256) k_sat_rows_rev(float* __restrict__ dsat4, int H, int W) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= 3 * H) return;
    const int lane = lane_id();
    const int c = row / H, y = row % H;
    float* p = dsat4 + (int64_t)y * W * 4 + c;
    double carry = 0.0;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = W - 1 - (x0 + lane);
        const double v = x >= 0 ? (double)p[(int64_t)x * 4] : 0.0;
        const double incl = wave_incl_scan(v);
        if (x >= 0) p[(int64_t)x * 4] = (float)(carry + incl);
        carry += __shfl(incl, 63, 64);
    }
}

__global__ void __launch_bounds__(256) k_sat_cols_rev(const float* __restrict__ dsat, const float* __restrict__ bg,
                                                      const float* __restrict__ act, int H, int W, float brightness,
                                                      float mul, const float* __restrict__ sc,
                                                      const float* __restrict__ d_pole, float* __restrict__ d_bg) {
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= 3 * W) return;
    if (sc) { brightness = sc[1]; mul = sc[2]; }
    const int c = col / W, x = col % W, lane = lane_id();
    double carry = 0.0;
    for (int y0 = 0; y0 < H; y0 += 64) {
        const int y = H - 1 - (y0 + lane);
        const int64_t i = y >= 0 ? ((int64_t)c * H + y) * W + x : 0;
        const double v = y >= 0 ? (double)dsat[((int64_t)y * W + x) * 4 + c] : 0.0;
        const double incl = wave_incl_scan(v);
        if (y >= 0) {
            float da = (float)(carry + incl) / 1000.f;
            if (d_pole && y == 0) da += d_pole[c] / (float)W;
            if (d_pole && y == H - 1) da += d_pole[3 + c] / (float)W;
            const bool clipped = (brightness + mul * bg[i]) > 20.f;
            d_bg[i] = clipped ? 0.f : da * act[i] * mul;
        }
        carry += __shfl(incl, 63, 64);
    }
}

}  // namespace

// SH irradiance coefficients (modules/integral_equirect.py:324-360): coeffs[k][c] = sum_i wq[i][k] * vals[i][c] over the
// quadrature lattice, conv[k][c] = A[k] * coeffs[k][c] / pi.  One workgroup per (k, c), float64 partial sums.
__global__ void __launch_bounds__(256) k_sh_project(const float* __restrict__ vals, const float* __restrict__ wq, int64_t n,
                                                    int K, const float* __restrict__ A, float* __restrict__ coeffs,
                                                    float* __restrict__ conv) {
    __shared__ double ws[4];
    const int k = blockIdx.x / 3, c = blockIdx.x % 3;
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) a += (double)wq[i * K + k] * (double)vals[i * 3 + c];
    for (int d = 32; d > 0; d >>= 1) a += __shfl_down(a, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(ws[0] + ws[1] + ws[2] + ws[3]);
        coeffs[k * 3 + c] = v;
        if (conv) conv[k * 3 + c] = (A[k] * v) / 3.14159265358979323846f;
    }
}

// pole[0][c] = mean of the first row of activated[c], pole[1][c] = mean of its last row (:499-502): one wave per mean
__global__ void __launch_bounds__(64) k_pole_rows(const float* __restrict__ act, int H, int W, float* __restrict__ pole) {
    const int which = blockIdx.x / 3, c = blockIdx.x % 3, lane = lane_id();
    const float* row = act + ((int64_t)c * H + (which ? H - 1 : 0)) * W;
    double a = 0.0;
    for (int x = lane; x < W; x += 64) a += (double)row[x];
    for (int d = 32; d > 0; d >>= 1) a += __shfl_down(a, d, 64);
    if (lane == 0) pole[which * 3 + c] = (float)(a / (double)W);
}

extern "C" int nmf_sat_build(const float* bg_mat, int32_t H, int32_t W, float brightness, float mul,
                             const float* scalars_dev, float* activated, float* sat, float* pole_rows, float* sat_i4,
                             void* stream) {
    NMF_REQUIRE(bg_mat && activated && sat && H > 1 && W > 1, NMF_EINVAL, "nmf_sat_build: null/size");
    hipStream_t st = (hipStream_t)stream;
    NMF_LAUNCH(k_sat_cols, dim3((unsigned)cdiv(3 * W, 4)), dim3(256), 0, st, bg_mat, H, W, brightness, mul,
                       scalars_dev, activated, sat);
    NMF_LAUNCH(k_sat_rows, dim3((unsigned)cdiv(3 * H, 4)), dim3(256), 0, st, sat, H, W, sat_i4);
    if (pole_rows) NMF_LAUNCH(k_pole_rows, dim3(6), dim3(64), 0, st, activated, H, W, pole_rows);
    NMF_CHECK_LAUNCH("nmf_sat_build");
    return NMF_OK;
}

extern "C" int nmf_sh_project(const float* vals, const float* wq, int64_t n, int32_t K, const float* sh_A, float* coeffs,
                              float* conv, void* stream) {
    NMF_REQUIRE(vals && wq && coeffs && n > 0 && K > 0 && K <= 64 && (!conv || sh_A), NMF_EINVAL, "nmf_sh_project: args");
    NMF_LAUNCH(k_sh_project, dim3((unsigned)(3 * K)), dim3(256), 0, (hipStream_t)stream, vals, wq, n, (int)K, sh_A,
                       coeffs, conv);
    NMF_CHECK_LAUNCH("nmf_sh_project");
    return NMF_OK;
}

extern "C" int nmf_sat_build_bwd(float* d_sat, const float* bg_mat, const float* activated, int32_t H, int32_t W,
                                 float brightness, float mul, const float* scalars_dev, const float* d_pole, float* d_bg,
                                 void* stream) {
    NMF_REQUIRE(d_sat && bg_mat && activated && d_bg && H > 1 && W > 1, NMF_EINVAL, "nmf_sat_build_bwd: null/size");
    hipStream_t st = (hipStream_t)stream;
    NMF_LAUNCH(k_sat_rows_rev, dim3((unsigned)cdiv(3 * H, 4)), dim3(256), 0, st, d_sat, H, W);
    NMF_LAUNCH(k_sat_cols_rev, dim3((unsigned)cdiv(3 * W, 4)), dim3(256), 0, st, d_sat, bg_mat, activated, H,
                       W, brightness, mul, scalars_dev, d_pole, d_bg);
    NMF_CHECK_LAUNCH("nmf_sat_build_bwd");
    return NMF_OK;
}

extern "C" int nmf_sat_lookup_fwd(const float* sat, int32_t H, int32_t W, const float* dirs, int32_t dirs_ld,
                                  const float* sa, int64_t R, float mipbias, const float* scalars_dev,
                                  const float* pole_rows, int32_t layout, float* out, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_sat_lookup_fwd: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(sat && dirs && sa && pole_rows && out, NMF_EINVAL, "nmf_sat_lookup_fwd: null");
    NMF_REQUIRE(dirs_ld == 3 || dirs_ld == 6, NMF_EINVAL, "nmf_sat_lookup_fwd: dirs_ld must be 3 or 6");
    EnvTab tab{sat, H, W, layout == 1};
    if (layout == 1)
        NMF_LAUNCH(k_env_lookup_fwd<1>, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, tab, dirs,
                           (int)dirs_ld, sa, R, mipbias, scalars_dev, pole_rows, out);
    else
        NMF_LAUNCH(k_env_lookup_fwd<0>, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, tab, dirs,
                           (int)dirs_ld, sa, R, mipbias, scalars_dev, pole_rows, out);
    NMF_CHECK_LAUNCH("nmf_sat_lookup_fwd");
    return NMF_OK;
}

extern "C" int64_t nmf_sat_lookup_bwd_workspace_bytes(int64_t R) {
    const int64_t r = R < 0 ? 0 : R;      // header | one slot per (workgroup of 256 lookups, tile) | 12 corner records per lookup
    return (int64_t)sizeof(EnvBinHeader) + cdiv(r, ENV_BWD_THREADS) * ENV_MAX_TILES * 4 + r * 12 * (int64_t)sizeof(CornerRec);
}

// (Round 5, tools/env_bwd_bench.py: pass 3 takes 43 us at 1.1 M records and 31 us at 0.2 M, with items of 512 .. 4096 records and with
// four record loads of a lane in flight alike -- its time is the flush: every touched tile adds its whole 33 x 65 x 3 window to dSAT
// with memory-side float atomics, 1.6 M lane-atomics for the 256 tiles of a 512 x 1024 map at the 156 G/s of tools/ub/atom2.hip, and
// smaller items only flush more windows.)
extern "C" int nmf_sat_lookup_bwd_binned(const float* sat, int32_t H, int32_t W, const float* dirs, int32_t dirs_ld,
                                         const float* sa, int64_t R, float mipbias, const float* scalars_dev, int32_t layout,
                                         const float* d_out, float* d_sat, float* d_pole, float* d_dirs, float* d_mipbias,
                                         void* workspace, int64_t workspace_bytes, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_sat_lookup_bwd_binned: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(sat && dirs && sa && d_out && d_sat && workspace, NMF_EINVAL, "nmf_sat_lookup_bwd_binned: null");
    // d_pole null: the table role alone (its other half is nmf_sat_lookup_bwd_dirs: pole rows, d_dirs, d_mipbias)
    NMF_REQUIRE(d_pole || (!d_dirs && !d_mipbias), NMF_EINVAL, "nmf_sat_lookup_bwd_binned: d_dirs / d_mipbias need d_pole");
    NMF_REQUIRE(dirs_ld == 3 || dirs_ld == 6, NMF_EINVAL, "nmf_sat_lookup_bwd_binned: dirs_ld must be 3 or 6");
    const int ntx = (int)cdiv(W, ENV_TILE_W), nty = (int)cdiv(H, ENV_TILE_H);
    NMF_REQUIRE(ntx * nty <= ENV_MAX_TILES, NMF_EINVAL, "nmf_sat_lookup_bwd_binned: map larger than 1024 tiles of 32 x 64");
    const int64_t nb = cdiv(R, ENV_BWD_THREADS);             // lookups: 256 per workgroup in every role
    const int64_t res_bytes = (nb * (int64_t)(ntx * nty) * 4 + 15) & ~(int64_t)15;
    NMF_REQUIRE(workspace_bytes >= (int64_t)sizeof(EnvBinHeader) + res_bytes + (int64_t)sizeof(CornerRec) && ((uintptr_t)workspace & 15) == 0,
                NMF_EINVAL, "nmf_sat_lookup_bwd_binned: workspace too small or not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    EnvBwdArgs A;
    A.tab = EnvTab{sat, H, W, layout == 1};
    A.dirs = dirs; A.ld = dirs_ld; A.sa = sa; A.R = R; A.mipbias = mipbias; A.sc = scalars_dev; A.d_out = d_out;
    A.d_sat4 = d_sat; A.d_pole = d_pole; A.d_dirs = d_dirs; A.d_mipbias = d_mipbias;
    A.hdr = reinterpret_cast<EnvBinHeader*>(workspace);
    A.blockres = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + sizeof(EnvBinHeader));
    A.recs = reinterpret_cast<CornerRec*>(reinterpret_cast<char*>(workspace) + sizeof(EnvBinHeader) + res_bytes);
    A.cap = (workspace_bytes - (int64_t)sizeof(EnvBinHeader) - res_bytes) / (int64_t)sizeof(CornerRec);
    A.ntx = ntx; A.nt = ntx * nty;
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(EnvBinHeader), st);
    if (e != hipSuccess) return nmf_fail((int)e, "nmf_sat_lookup_bwd_binned: hipMemsetAsync");
    // The dual-number workgroups (d_dirs, d_mipbias) ride next to the COUNTING pass: that pass is LDS counting only, the riders'
    // arithmetic fills it.  Measured on the 247 k lookups of a steady-state step (tools/env_bwd_bench.py, us): all next to pass 1
    // 98, 50 / 50 next to passes 1 / 2 101, 80 / 20 104, all next to pass 2 111, all behind pass 3 155 (direct scatter: 176).
    // (Round 4 read the split from the environment on every call; the measurement is in, the split is fixed.)
    const int split1 = d_pole ? 100 : 0, split2 = 0;
    const int64_t d1 = nb * split1 / 100, d2 = std::min(nb - d1, nb * split2 / 100), d3 = d_pole ? nb - d1 - d2 : 0;
    const int accum_blocks = (int)(d3 > 768 ? d3 : 768);      // (env_role needs at least as many pass workgroups as riders)
    if (layout == 1) {
        NMF_LAUNCH(k_env_bin_count<1>, dim3((unsigned)(nb + d1)), dim3(ENV_BWD_THREADS), 0, st, A, (int)nb, (int)d1, (int64_t)0);
        NMF_LAUNCH(k_env_bin_scatter<1>, dim3((unsigned)(nb + d2)), dim3(ENV_BWD_THREADS), 0, st, A, (int)nb, (int)d2, d1);
        NMF_LAUNCH(k_env_bin_accum<1>, dim3((unsigned)(accum_blocks + d3)), dim3(ENV_BWD_THREADS), 0, st, A, accum_blocks, (int)d3, d1 + d2);
    } else {
        NMF_LAUNCH(k_env_bin_count<0>, dim3((unsigned)(nb + d1)), dim3(ENV_BWD_THREADS), 0, st, A, (int)nb, (int)d1, (int64_t)0);
        NMF_LAUNCH(k_env_bin_scatter<0>, dim3((unsigned)(nb + d2)), dim3(ENV_BWD_THREADS), 0, st, A, (int)nb, (int)d2, d1);
        NMF_LAUNCH(k_env_bin_accum<0>, dim3((unsigned)(accum_blocks + d3)), dim3(ENV_BWD_THREADS), 0, st, A, accum_blocks, (int)d3, d1 + d2);
    }
    NMF_CHECK_LAUNCH("nmf_sat_lookup_bwd_binned");
    return NMF_OK;
}

extern "C" int nmf_sat_lookup_bwd_dirs(const float* sat, int32_t H, int32_t W, const float* dirs, int32_t dirs_ld, const float* sa,
                                       int64_t R, float mipbias, const float* scalars_dev, int32_t layout, const float* d_out,
                                       float* d_pole, float* d_dirs, float* d_mipbias, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_sat_lookup_bwd_dirs: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(sat && dirs && sa && d_out && d_pole, NMF_EINVAL, "nmf_sat_lookup_bwd_dirs: null");
    NMF_REQUIRE(dirs_ld == 3 || dirs_ld == 6, NMF_EINVAL, "nmf_sat_lookup_bwd_dirs: dirs_ld must be 3 or 6");
    EnvBwdArgs A{};
    A.tab = EnvTab{sat, H, W, layout == 1};
    A.dirs = dirs; A.ld = dirs_ld; A.sa = sa; A.R = R; A.mipbias = mipbias; A.sc = scalars_dev; A.d_out = d_out;
    A.d_pole = d_pole; A.d_dirs = d_dirs; A.d_mipbias = d_mipbias;
    const dim3 grid((unsigned)cdiv(R, ENV_BWD_THREADS));
    if (layout == 1) NMF_LAUNCH(k_env_dirs<1>, grid, dim3(ENV_BWD_THREADS), 0, (hipStream_t)stream, A);
    else NMF_LAUNCH(k_env_dirs<0>, grid, dim3(ENV_BWD_THREADS), 0, (hipStream_t)stream, A);
    NMF_CHECK_LAUNCH("nmf_sat_lookup_bwd_dirs");
    return NMF_OK;
}

extern "C" int nmf_sat_lookup_bwd(const float* sat, int32_t H, int32_t W, const float* dirs, int32_t dirs_ld,
                                  const float* sa, int64_t R, float mipbias, const float* scalars_dev, int32_t layout,
                                  const float* d_out, float* d_sat, float* d_pole, float* d_dirs, float* d_mipbias,
                                  void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_sat_lookup_bwd: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(sat && dirs && sa && d_out && d_pole, NMF_EINVAL, "nmf_sat_lookup_bwd: null");
    NMF_REQUIRE(dirs_ld == 3 || dirs_ld == 6, NMF_EINVAL, "nmf_sat_lookup_bwd: dirs_ld must be 3 or 6");
    EnvTab tab{sat, H, W, layout == 1};
    if (layout == 1)
        NMF_LAUNCH(k_env_lookup_bwd<1>, dim3((unsigned)(9 * cdiv(R, ENV_BWD_THREADS))), dim3(ENV_BWD_THREADS), 0,
                           (hipStream_t)stream, tab, dirs, (int)dirs_ld, sa, R, mipbias, scalars_dev, d_out, d_sat, d_pole,
                           d_dirs, d_mipbias);
    else
        NMF_LAUNCH(k_env_lookup_bwd<0>, dim3((unsigned)(9 * cdiv(R, ENV_BWD_THREADS))), dim3(ENV_BWD_THREADS), 0,
                           (hipStream_t)stream, tab, dirs, (int)dirs_ld, sa, R, mipbias, scalars_dev, d_out, d_sat, d_pole,
                           d_dirs, d_mipbias);
    NMF_CHECK_LAUNCH("nmf_sat_lookup_bwd");
    return NMF_OK;
}
