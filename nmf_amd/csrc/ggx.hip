// Secondary-ray generation and shading mix for gfx950.
//
// nmf_ggx_rays_fwd/bwd replace, per secondary ray, PseudoRandomSampler.draw + GGXSampler.sample + compute_prob
// (reference: brdf_samplers/base.py:11-20, brdf_samplers/ggx.py:61-268) and the per-ray glue of
// Microfacet.forward that turns the sample into a bounce ray and MLP inputs (models/microfacet.py:377-456):
//   u = (Sobol[j] + 0.25 U[row]) mod 1;  L = VNDF sample;  H = normalize((V+L)/2);
//   half_local = B H, diff_local = B L (B = rows tangent, bitangent, normal);  mipval = -log(count) - log pdf;
//   ray = (x + 5e-3 L, L)
// nmf_shade_mix_fwd/bwd replace the Fresnel-Schlick mix (models/microfacet.py:595-613):
//   F = f0 + (1-f0)(1-|V.H|)^5;  contrib = (F * Li * brdf + (1-F) * diffuse) / count
//
// One lane per ray on the compact ray list (row_of_ray, j_of_ray) -- the reference's padded
// [bounce points x m] tensors and their ~20 [R,*] temporaries are never formed.  The backward of the ray
// generator (d L / d normal, d L / d roughness: the second-order path through the GGX frame) runs the same
// templated sampler on 4-tangent dual numbers.
#include "common.hpp"
#include "dual.hpp"

namespace {

constexpr float EPS_F = 1.1920929e-07f;
constexpr float PI_F = 3.14159265358979323846f;

template <class T> struct V3 { T x, y, z; };

template <class T> __device__ __forceinline__ V3<T> cross(const V3<T>& a, const V3<T>& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class T> __device__ __forceinline__ T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// mutils.normalize: v / sqrt(clip(sum v^2, eps))
template <class T> __device__ __forceinline__ V3<T> nrm(const V3<T>& v) {
    const T inv = set_val(v.x, 1.f) / d_sqrt(d_clipmin(dot(v, v), EPS_F));
    return {v.x * inv, v.y * inv, v.z * inv};
}
template <class T> __device__ __forceinline__ V3<T> lift(const V3<float>& v, const T& like) {
    return {set_val(like, v.x), set_val(like, v.y), set_val(like, v.z)};
}
__device__ __forceinline__ float safe_mod(float x) {   // safemath.safe_trig_helper: x % (100 pi), python modulo
    const float t = 100.f * PI_F;
    float r = fmodf(x, t);
    if (r != 0.f && r < 0.f) r += t;
    return r;
}

template <class T>
struct GgxOut {
    V3<T> L;
    V3<T> tangent, bitangent;   // rows of the basis (with the normal)
    V3<T> Hl;                   // sampled half vector in the local frame
};

// GGXSampler.sample for ONE ray (ggx.py:61-226); V, N and r carry tangents when T is a dual number (V only for the
// rays of recursion level >= 1, whose view direction is the sampled direction of the level above)
template <class T>
__device__ __forceinline__ GgxOut<T> ggx_sample(const V3<T>& V, const V3<T>& N, const T& r, float u1, float u2) {
    GgxOut<T> o;
    const V3<float> z_up = {0.f, 0.f, 1.f}, x_up = {-1.f, 0.f, 0.f};
    const V3<T> up = lift(fabsf(val(N.z)) < 0.999f ? z_up : x_up, r);
    o.tangent = nrm(cross(up, N));
    o.bitangent = nrm(cross(N, o.tangent));
    const V3<T> Vl = {dot(o.tangent, V), dot(o.bitangent, V), dot(N, V)};
    const V3<T> Vs = nrm(V3<T>{r * Vl.x, r * Vl.y, Vl.z});
    const V3<T> T1 = val(Vs.z) < 0.999f ? nrm(cross(Vs, lift(z_up, r))) : lift(x_up, r);
    const V3<T> T2 = nrm(cross(T1, Vs));
    const float a = fminf(1.f / fmaxf(1.f + val(Vs.z), 1e-8f), 1e4f);              // detached (:116)
    const float rr = sqrtf(u1);
    const bool lower = u2 < a;
    const float phi = lower ? u2 / a * PI_F : (u2 - a) / (1.f - a) * PI_F + PI_F;
    const float pm = safe_mod(phi);
    const T P1 = set_val(r, rr * cosf(pm));
    const T P2 = lower ? set_val(r, rr * sinf(pm)) : Vs.z * (rr * sinf(pm));
    const T w3 = d_sqrt(d_clipmin(1.f - P1 * P1 - P2 * P2, EPS_F));
    const V3<T> Ns = {P1 * T1.x + P2 * T2.x + w3 * Vs.x, P1 * T1.y + P2 * T2.y + w3 * Vs.y,
                      P1 * T1.z + P2 * T2.z + w3 * Vs.z};
    o.Hl = nrm(V3<T>{Ns.x * r, Ns.y * r, Ns.z});
    const V3<T> H = {o.tangent.x * o.Hl.x + o.bitangent.x * o.Hl.y + N.x * o.Hl.z,
                     o.tangent.y * o.Hl.x + o.bitangent.y * o.Hl.y + N.y * o.Hl.z,
                     o.tangent.z * o.Hl.x + o.bitangent.z * o.Hl.y + N.z * o.Hl.z};
    const T vh2 = dot(V, H) * 2.f;
    V3<T> wi = nrm(V3<T>{vh2 * H.x - V.x, vh2 * H.y - V.y, vh2 * H.z - V.z});
    const float sgn = val(dot(wi, N)) > 0.f ? 1.f : -1.f;
    o.L = {wi.x * sgn, wi.y * sgn, wi.z * sgn};
    return o;
}

// GGXSampler.compute_prob (ggx.py:228-268), isotropic
__device__ __forceinline__ float ggx_prob(const V3<float>& li, const V3<float>& lo, const V3<float>& h, float r) {
    const float r2 = fmaxf(r, EPS_F);
    const float r1 = fmaxf(r + r2, EPS_F) / 2.f;
    const float lam = (-1.f + sqrtf(fmaxf(1.f + ((li.x * r1) * (li.x * r1) + (li.y * r2) * (li.y * r2)) /
                                                    fmaxf(li.z * li.z, 1e-6f), EPS_F))) / 2.f;
    const float q = h.x * h.x / (r1 * r1) + h.y * h.y / (r2 * r2) + h.z * h.z;
    const float invD = PI_F * r1 * r2 * q * q;
    const float logD = -logf(fmaxf((1.f + lam) * invD, EPS_F)) - logf(fmaxf(4.f * lo.z, EPS_F));
    return li.z > 0.f ? expf(logD) : 0.f;
}

struct RowIn {
    const float* V;      // [Mb][3]  direction towards the viewer
    const float* N;      // [Mb][3]  normal, already flipped towards the viewer
    const float* r;      // [Mb]
    const float* x;      // [Mb][3]  bounce point
    const float* off;    // [Mb][2]  Sobol row offsets (uniform draws, NOT yet scaled by 0.25)
    const int32_t* cnt;  // [Mb]     rays of the row
};

__device__ __forceinline__ void load_ray(const RowIn& in, const float* __restrict__ sobol, int32_t row, int32_t jj,
                                         V3<float>& V, V3<float>& N, float& r, float& u1, float& u2) {
    V = {in.V[row * 3], in.V[row * 3 + 1], in.V[row * 3 + 2]};
    N = {in.N[row * 3], in.N[row * 3 + 1], in.N[row * 3 + 2]};
    r = in.r[row];
    // (angs + offset) % 1.0 with offset = U * 0.25 (base.py:18-19); fp32, python modulo of non-negative values
    u1 = fmodf(sobol[jj * 2] + in.off[row * 2] * 0.25f, 1.f);
    u2 = fmodf(sobol[jj * 2 + 1] + in.off[row * 2 + 1] * 0.25f, 1.f);
}

__global__ void __launch_bounds__(256) k_ggx_rays_fwd(RowIn in, const float* __restrict__ sobol,
                                                      const int32_t* __restrict__ row_of_ray,
                                                      const int32_t* __restrict__ j_of_ray, int64_t R,
                                                      float* __restrict__ L_out, float* __restrict__ half_l,
                                                      float* __restrict__ diff_l, float* __restrict__ lpdf,
                                                      float* __restrict__ mipval, float* __restrict__ rays) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int32_t row = row_of_ray[i];
    V3<float> V, N;
    float r, u1, u2;
    load_ray(in, sobol, row, j_of_ray[i], V, N, r, u1, u2);
    const GgxOut<float> o = ggx_sample<float>(V, N, r, u1, u2);
    const V3<float> L = o.L;
    // local-frame vectors: basis rows . v   (microfacet.py:406-428)
    const V3<float> Ll = {dot(o.tangent, L), dot(o.bitangent, L), dot(N, L)};
    const V3<float> Vl = {dot(o.tangent, V), dot(o.bitangent, V), dot(N, V)};
    const float p = ggx_prob(Ll, Vl, o.Hl, r);
    const float lp = logf(fmaxf(p, EPS_F));
    const V3<float> H = nrm(V3<float>{(V.x + L.x) / 2.f, (V.y + L.y) / 2.f, (V.z + L.z) / 2.f});
    L_out[i * 3] = L.x; L_out[i * 3 + 1] = L.y; L_out[i * 3 + 2] = L.z;
    half_l[i * 3] = dot(o.tangent, H); half_l[i * 3 + 1] = dot(o.bitangent, H); half_l[i * 3 + 2] = dot(N, H);
    diff_l[i * 3] = Ll.x; diff_l[i * 3 + 1] = Ll.y; diff_l[i * 3 + 2] = Ll.z;
    lpdf[i] = lp;
    mipval[i] = -logf(fmaxf((float)in.cnt[row], 1.f)) - lp;                                     // :448
    const float* x = in.x + row * 3;
    float* q = rays + i * 6;
    q[0] = x[0] + L.x * 5e-3f; q[1] = x[1] + L.y * 5e-3f; q[2] = x[2] + L.z * 5e-3f;           // :450-456
    q[3] = L.x; q[4] = L.y; q[5] = L.z;
}

// adjoint: dL [R][3] -> per-ray (dN, dr) [R][4] (NT = 4) or (dN, dr, dV) [R][7] (NT = 7); the caller reduces them per row
// with nmf_segment_sum
template <int NT>
__global__ void __launch_bounds__(256) k_ggx_rays_bwd(RowIn in, const float* __restrict__ sobol,
                                                      const int32_t* __restrict__ row_of_ray,
                                                      const int32_t* __restrict__ j_of_ray, int64_t R,
                                                      const float* __restrict__ dL, const float* __restrict__ d_rays,
                                                      float* __restrict__ d_nr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int32_t row = row_of_ray[i];
    V3<float> V, Nf;
    float rf, u1, u2;
    load_ray(in, sobol, row, j_of_ray[i], V, Nf, rf, u1, u2);
    typedef Dual<NT> D;
    V3<D> N = {mk_const<NT>(Nf.x), mk_const<NT>(Nf.y), mk_const<NT>(Nf.z)};
    V3<D> Vd = {mk_const<NT>(V.x), mk_const<NT>(V.y), mk_const<NT>(V.z)};
    D r = mk_const<NT>(rf);
    N.x.d[0] = 1.f; N.y.d[1] = 1.f; N.z.d[2] = 1.f; r.d[3] = 1.f;
    if (NT == 7) { Vd.x.d[NT - 3] = 1.f; Vd.y.d[NT - 2] = 1.f; Vd.z.d[NT - 1] = 1.f; }
    const GgxOut<D> o = ggx_sample<D>(Vd, N, r, u1, u2);
    // adjoint of L itself plus of the bounce ray (origin x + 5e-3 L | direction L), either may be absent
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (dL) { g0 = dL[i * 3]; g1 = dL[i * 3 + 1]; g2 = dL[i * 3 + 2]; }
    if (d_rays) {
        const float* q = d_rays + i * 6;
        g0 += q[3] + 5e-3f * q[0]; g1 += q[4] + 5e-3f * q[1]; g2 += q[5] + 5e-3f * q[2];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) d_nr[i * NT + t] = g0 * o.L.x.d[t] + g1 * o.L.y.d[t] + g2 * o.L.z.d[t];
}

// ---- Fresnel mix ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_shade_mix_fwd(const float* __restrict__ Vrow, const float* __restrict__ f0row,
                                                       const float* __restrict__ diffrow,
                                                       const int32_t* __restrict__ cnt,
                                                       const int32_t* __restrict__ row_of_ray, int64_t R,
                                                       const float* __restrict__ L, const float* __restrict__ inc,
                                                       const float* __restrict__ brdf, float* __restrict__ contrib) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int32_t row = row_of_ray[i];
    const V3<float> V = {Vrow[row * 3], Vrow[row * 3 + 1], Vrow[row * 3 + 2]};
    const V3<float> Lv = {L[i * 3], L[i * 3 + 1], L[i * 3 + 2]};
    const V3<float> H = nrm(V3<float>{(V.x + Lv.x) / 2.f, (V.y + Lv.y) / 2.f, (V.z + Lv.z) / 2.f});
    const float c = fabsf(-(V.x * H.x + V.y * H.y + V.z * H.z));
    const float om = fminf(fmaxf(1.f - c, 0.f), 1.f);
    const float p5 = om * om * om * om * om;
    const float ec = fmaxf((float)cnt[row], 1.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float R0 = f0row[row * 3 + k];
        const float Fr = R0 + (1.f - R0) * p5;
        contrib[i * 3 + k] = (Fr * inc[i * 3 + k] * brdf[i * 3 + k] + (1.f - Fr) * diffrow[row * 3 + k]) / ec;
    }
}

// adjoints per ray: d_inc [R][3], d_brdf [R][3], dL [R][3], d_f0diff [R][6] = (d f0 | d diffuse), all overwritten,
// to be reduced per row by the caller
__global__ void __launch_bounds__(256) k_shade_mix_bwd(const float* __restrict__ Vrow, const float* __restrict__ f0row,
                                                       const float* __restrict__ diffrow,
                                                       const int32_t* __restrict__ cnt,
                                                       const int32_t* __restrict__ row_of_ray, int64_t R,
                                                       const float* __restrict__ L, const float* __restrict__ inc,
                                                       const float* __restrict__ brdf,
                                                       const float* __restrict__ d_rows /*[Mb][3]*/,
                                                       float* __restrict__ d_inc, float* __restrict__ d_brdf,
                                                       float* __restrict__ dL, float* __restrict__ d_f0diff,
                                                       float* __restrict__ dV /*[R][3] or null*/) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int32_t row = row_of_ray[i];
    const V3<float> V = {Vrow[row * 3], Vrow[row * 3 + 1], Vrow[row * 3 + 2]};
    const V3<float> Lv = {L[i * 3], L[i * 3 + 1], L[i * 3 + 2]};
    const V3<float> h = {(V.x + Lv.x) / 2.f, (V.y + Lv.y) / 2.f, (V.z + Lv.z) / 2.f};
    const float n2 = h.x * h.x + h.y * h.y + h.z * h.z;
    const float inv = 1.f / sqrtf(fmaxf(n2, EPS_F));
    const V3<float> H = {h.x * inv, h.y * inv, h.z * inv};
    const float d = -(V.x * H.x + V.y * H.y + V.z * H.z);
    const float c = fabsf(d);
    const float om_raw = 1.f - c;
    const float om = fminf(fmaxf(om_raw, 0.f), 1.f);
    const float p4 = om * om * om * om, p5 = p4 * om;
    const float ec = fmaxf((float)cnt[row], 1.f);
    float dp5 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g = d_rows[row * 3 + k] / ec;
        const float R0 = f0row[row * 3 + k], Li = inc[i * 3 + k], bw = brdf[i * 3 + k], df = diffrow[row * 3 + k];
        const float Fr = R0 + (1.f - R0) * p5;
        d_inc[i * 3 + k] = g * Fr * bw;
        d_brdf[i * 3 + k] = g * Fr * Li;
        const float dFr = g * (Li * bw - df);
        d_f0diff[i * 6 + k] = dFr * (1.f - p5);
        d_f0diff[i * 6 + 3 + k] = g * (1.f - Fr);
        dp5 += dFr * (1.f - R0);
    }
    // p5 = clip(1-c,0,1)^5 ; c = |d| ; d = -V.H ; H = h/|h| ; h = (V+L)/2
    const float dom = (om_raw >= 0.f && om_raw <= 1.f) ? 5.f * p4 * dp5 : 0.f;
    const float dd = -dom * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    const V3<float> dH = {-dd * V.x, -dd * V.y, -dd * V.z};
    const float hd = dH.x * H.x + dH.y * H.y + dH.z * H.z;
    const float s = n2 > EPS_F ? 1.f : 0.f;     // clip: below eps H = h / sqrt(eps), no projection term
    const float k2 = 0.5f * inv;
    const V3<float> gl = {k2 * (dH.x - s * hd * H.x), k2 * (dH.y - s * hd * H.y), k2 * (dH.z - s * hd * H.z)};
    dL[i * 3] = gl.x; dL[i * 3 + 1] = gl.y; dL[i * 3 + 2] = gl.z;
    if (dV) {      // V enters d = -V.H directly and through h = (V+L)/2 like L does
        dV[i * 3] = gl.x - dd * H.x; dV[i * 3 + 1] = gl.y - dd * H.y; dV[i * 3 + 2] = gl.z - dd * H.z;
    }
}

}  // namespace

static RowIn mk_rows(const float* V, const float* N, const float* r, const float* x, const float* off,
                     const int32_t* cnt) {
    RowIn in{V, N, r, x, off, cnt};
    return in;
}

extern "C" int nmf_ggx_rays_fwd(const float* V_rows, const float* N_rows, const float* r_rows, const float* x_rows,
                                const float* off_rows, const int32_t* cnt_rows, const float* sobol,
                                const int32_t* row_of_ray, const int32_t* j_of_ray, int64_t R, float* L,
                                float* half_local, float* diff_local, float* lpdf, float* mipval, float* rays,
                                void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_ggx_rays_fwd: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(V_rows && N_rows && r_rows && x_rows && off_rows && cnt_rows && sobol && row_of_ray && j_of_ray && L &&
                    half_local && diff_local && lpdf && mipval && rays,
                NMF_EINVAL, "nmf_ggx_rays_fwd: null");
    NMF_LAUNCH(k_ggx_rays_fwd, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream,
                       mk_rows(V_rows, N_rows, r_rows, x_rows, off_rows, cnt_rows), sobol, row_of_ray, j_of_ray, R, L,
                       half_local, diff_local, lpdf, mipval, rays);
    NMF_CHECK_LAUNCH("nmf_ggx_rays_fwd");
    return NMF_OK;
}

// GGXSampler.compute_prob as an operator of its own (the reference exposes it: brdf_samplers/ggx.py:228-268)
__global__ void __launch_bounds__(256) k_ggx_prob(const float* __restrict__ li, const float* __restrict__ lo,
                                                  const float* __restrict__ h, const float* __restrict__ r, int64_t R,
                                                  float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    out[i] = ggx_prob(V3<float>{li[i * 3], li[i * 3 + 1], li[i * 3 + 2]}, V3<float>{lo[i * 3], lo[i * 3 + 1], lo[i * 3 + 2]},
                      V3<float>{h[i * 3], h[i * 3 + 1], h[i * 3 + 2]}, r[i]);
}

extern "C" int nmf_ggx_prob(const float* dir_in_local, const float* dir_out_local, const float* half_local,
                            const float* rough, int64_t R, float* prob, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_ggx_prob: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(dir_in_local && dir_out_local && half_local && rough && prob, NMF_EINVAL, "nmf_ggx_prob: null");
    NMF_LAUNCH(k_ggx_prob, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, dir_in_local,
                       dir_out_local, half_local, rough, R, prob);
    NMF_CHECK_LAUNCH("nmf_ggx_prob");
    return NMF_OK;
}

extern "C" int nmf_ggx_rays_bwd(const float* V_rows, const float* N_rows, const float* r_rows, const float* off_rows,
                                const float* sobol, const int32_t* row_of_ray, const int32_t* j_of_ray, int64_t R,
                                const float* dL, const float* d_rays, float* d_nr, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_ggx_rays_bwd: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(V_rows && N_rows && r_rows && off_rows && sobol && row_of_ray && j_of_ray && d_nr, NMF_EINVAL,
                "nmf_ggx_rays_bwd: null");
    NMF_LAUNCH(k_ggx_rays_bwd<4>, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream,
                       mk_rows(V_rows, N_rows, r_rows, nullptr, off_rows, nullptr), sobol, row_of_ray, j_of_ray, R, dL, d_rays,
                       d_nr);
    NMF_CHECK_LAUNCH("nmf_ggx_rays_bwd");
    return NMF_OK;
}

extern "C" int nmf_ggx_rays_bwd_view(const float* V_rows, const float* N_rows, const float* r_rows, const float* off_rows,
                                     const float* sobol, const int32_t* row_of_ray, const int32_t* j_of_ray, int64_t R,
                                     const float* dL, const float* d_rays, float* d_nrv, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_ggx_rays_bwd_view: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(V_rows && N_rows && r_rows && off_rows && sobol && row_of_ray && j_of_ray && d_nrv, NMF_EINVAL,
                "nmf_ggx_rays_bwd_view: null");
    NMF_LAUNCH(k_ggx_rays_bwd<7>, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream,
                       mk_rows(V_rows, N_rows, r_rows, nullptr, off_rows, nullptr), sobol, row_of_ray, j_of_ray, R, dL, d_rays,
                       d_nrv);
    NMF_CHECK_LAUNCH("nmf_ggx_rays_bwd_view");
    return NMF_OK;
}

extern "C" int nmf_shade_mix_fwd(const float* V_rows, const float* f0_rows, const float* diffuse_rows,
                                 const int32_t* cnt_rows, const int32_t* row_of_ray, int64_t R, const float* L,
                                 const float* incoming, const float* brdf, float* contrib, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_shade_mix_fwd: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(V_rows && f0_rows && diffuse_rows && cnt_rows && row_of_ray && L && incoming && brdf && contrib,
                NMF_EINVAL, "nmf_shade_mix_fwd: null");
    NMF_LAUNCH(k_shade_mix_fwd, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, V_rows, f0_rows,
                       diffuse_rows, cnt_rows, row_of_ray, R, L, incoming, brdf, contrib);
    NMF_CHECK_LAUNCH("nmf_shade_mix_fwd");
    return NMF_OK;
}

extern "C" int nmf_shade_mix_bwd(const float* V_rows, const float* f0_rows, const float* diffuse_rows,
                                 const int32_t* cnt_rows, const int32_t* row_of_ray, int64_t R, const float* L,
                                 const float* incoming, const float* brdf, const float* d_rows, float* d_incoming,
                                 float* d_brdf, float* dL, float* d_f0diff, void* stream) {
    return nmf_shade_mix_bwd_view(V_rows, f0_rows, diffuse_rows, cnt_rows, row_of_ray, R, L, incoming, brdf, d_rows,
                                  d_incoming, d_brdf, dL, d_f0diff, nullptr, stream);
}

extern "C" int nmf_shade_mix_bwd_view(const float* V_rows, const float* f0_rows, const float* diffuse_rows,
                                      const int32_t* cnt_rows, const int32_t* row_of_ray, int64_t R, const float* L,
                                      const float* incoming, const float* brdf, const float* d_rows, float* d_incoming,
                                      float* d_brdf, float* dL, float* d_f0diff, float* dV, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_shade_mix_bwd: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(V_rows && f0_rows && diffuse_rows && cnt_rows && row_of_ray && L && incoming && brdf && d_rows &&
                    d_incoming && d_brdf && dL && d_f0diff,
                NMF_EINVAL, "nmf_shade_mix_bwd: null");
    NMF_LAUNCH(k_shade_mix_bwd, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, V_rows, f0_rows,
                       diffuse_rows, cnt_rows, row_of_ray, R, L, incoming, brdf, d_rows, d_incoming, d_brdf, dL, d_f0diff, dV);
    NMF_CHECK_LAUNCH("nmf_shade_mix_bwd");
    return NMF_OK;
}
