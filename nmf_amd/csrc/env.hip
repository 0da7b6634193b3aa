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
template <class T>
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
    if (t.i4) {
        const float4* p = reinterpret_cast<const float4*>(t.sat);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 a = (xi0 && yi0) ? p[y0 * t.W + x0] : z;
        const float4 b = (xi1 && yi0) ? p[y0 * t.W + x0 + 1] : z;
        const float4 c = (xi0 && yi1) ? p[(y0 + 1) * t.W + x0] : z;
        const float4 d = (xi1 && yi1) ? p[(y0 + 1) * t.W + x0 + 1] : z;
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
template <class T>
struct SumAcc {   // forward value: sum of boxes, each divided by the ORIGINAL size
    EnvTab tab;
    T size;
    T total[3];
    T cur[3];
    __device__ void begin() {}
    __device__ void corner(const T& x, const T& y, int k) {
        T s[3];
        sat_sample(tab, x, y, s);
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
__global__ void __launch_bounds__(256) k_env_lookup_fwd(EnvTab tab, const float* __restrict__ dirs, int ld,
                                                        const float* __restrict__ sa, int64_t R, float mipbias,
                                                        const float* __restrict__ sc,
                                                        const float* __restrict__ pole_rows /*[2][3] top,bot*/,
                                                        float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (sc) mipbias = sc[0];
    const float* q = dirs + r * ld + (ld - 3);            // ld = 6: [origin | direction] ray rows
    const float a = q[0], b = q[1], c = q[2];
    Geometry<float> g = env_geometry<float>(tab.H, tab.W, a, b, c, sa[r], mipbias);
    SumAcc<float> acc;
    acc.tab = tab;
    acc.size = g.size;
    acc.total[0] = acc.total[1] = acc.total[2] = 0.f;
    box_wrap(g.rect, acc);
    const float cutoff = 1.f - 2.f / (float)tab.H * 3.f;
    float v[3] = {acc.total[0] * 1000.f, acc.total[1] * 1000.f, acc.total[2] * 1000.f};
    if (g.cy > cutoff) { v[0] = pole_rows[3]; v[1] = pole_rows[4]; v[2] = pole_rows[5]; }
    if (g.cy < -cutoff) { v[0] = pole_rows[0]; v[1] = pole_rows[1]; v[2] = pole_rows[2]; }
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
            SumAcc<D> acc;
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
    if (d_mipbias) {   // wave reduction, one atomic per wave
        for (int d = 32; d > 0; d >>= 1) dm += __shfl_down(dm, d, 64);
        if (lane_id() == 0 && dm != 0.f) atomicAdd(d_mipbias, dm);
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
    hipLaunchKernelGGL(k_sat_cols, dim3((unsigned)cdiv(3 * W, 4)), dim3(256), 0, st, bg_mat, H, W, brightness, mul,
                       scalars_dev, activated, sat);
    hipLaunchKernelGGL(k_sat_rows, dim3((unsigned)cdiv(3 * H, 4)), dim3(256), 0, st, sat, H, W, sat_i4);
    if (pole_rows) hipLaunchKernelGGL(k_pole_rows, dim3(6), dim3(64), 0, st, activated, H, W, pole_rows);
    NMF_CHECK_LAUNCH("nmf_sat_build");
    return NMF_OK;
}

extern "C" int nmf_sh_project(const float* vals, const float* wq, int64_t n, int32_t K, const float* sh_A, float* coeffs,
                              float* conv, void* stream) {
    NMF_REQUIRE(vals && wq && coeffs && n > 0 && K > 0 && K <= 64 && (!conv || sh_A), NMF_EINVAL, "nmf_sh_project: args");
    hipLaunchKernelGGL(k_sh_project, dim3((unsigned)(3 * K)), dim3(256), 0, (hipStream_t)stream, vals, wq, n, (int)K, sh_A,
                       coeffs, conv);
    NMF_CHECK_LAUNCH("nmf_sh_project");
    return NMF_OK;
}

extern "C" int nmf_sat_build_bwd(float* d_sat, const float* bg_mat, const float* activated, int32_t H, int32_t W,
                                 float brightness, float mul, const float* scalars_dev, const float* d_pole, float* d_bg,
                                 void* stream) {
    NMF_REQUIRE(d_sat && bg_mat && activated && d_bg && H > 1 && W > 1, NMF_EINVAL, "nmf_sat_build_bwd: null/size");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sat_rows_rev, dim3((unsigned)cdiv(3 * H, 4)), dim3(256), 0, st, d_sat, H, W);
    hipLaunchKernelGGL(k_sat_cols_rev, dim3((unsigned)cdiv(3 * W, 4)), dim3(256), 0, st, d_sat, bg_mat, activated, H,
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
    hipLaunchKernelGGL(k_env_lookup_fwd, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, tab, dirs,
                       (int)dirs_ld, sa, R, mipbias, scalars_dev, pole_rows, out);
    NMF_CHECK_LAUNCH("nmf_sat_lookup_fwd");
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
    hipLaunchKernelGGL(k_env_lookup_bwd, dim3((unsigned)(9 * cdiv(R, ENV_BWD_THREADS))), dim3(ENV_BWD_THREADS), 0,
                       (hipStream_t)stream, tab, dirs, (int)dirs_ld, sa, R, mipbias, scalars_dev, d_out, d_sat, d_pole,
                       d_dirs, d_mipbias);
    NMF_CHECK_LAUNCH("nmf_sat_lookup_bwd");
    return NMF_OK;
}
