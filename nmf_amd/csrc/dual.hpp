// Forward-mode dual numbers for the gfx950 kernels: a value plus N tangents.  The SAME templated device function is
// instantiated on float (forward pass, exact reference op order) and on Dual<N> (backward pass: exact derivatives
// with the forward's branch structure by construction).  Used by env.hip (d/d direction, d/d mipbias) and ggx.hip
// (d/d normal, d/d roughness).
#pragma once
#include <hip/hip_runtime.h>

constexpr float LN2_F = 0.69314718055994530942f;

// ---- forward-mode dual number with N tangents -------------------------------------------------
template <int N>
struct Dual {
    float v;
    float d[N];
};
template <int N> __device__ __forceinline__ Dual<N> mk_const(float v) { Dual<N> r; r.v = v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = 0.f; return r; }

__device__ __forceinline__ float val(float a) { return a; }
template <int N> __device__ __forceinline__ float val(const Dual<N>& a) { return a.v; }

#define DUAL_BIN(op, expr_v, expr_d)                                                                       \
    template <int N> __device__ __forceinline__ Dual<N> op(const Dual<N>& a, const Dual<N>& b) {           \
        Dual<N> r; r.v = expr_v;                                                                           \
        _Pragma("unroll") for (int i = 0; i < N; ++i) r.d[i] = expr_d; return r; }
DUAL_BIN(operator+, a.v + b.v, a.d[i] + b.d[i])
DUAL_BIN(operator-, a.v - b.v, a.d[i] - b.d[i])
DUAL_BIN(operator*, a.v * b.v, a.d[i] * b.v + a.v * b.d[i])
DUAL_BIN(operator/, a.v / b.v, (a.d[i] - (a.v / b.v) * b.d[i]) / b.v)
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N>& a, float b) { Dual<N> r = a; r.v = a.v + b; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a, float b) { Dual<N> r = a; r.v = a.v - b; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(float a, const Dual<N>& b) { Dual<N> r; r.v = a - b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = -b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& b) { return 0.f - b; }
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& a, float b) { Dual<N> r; r.v = a.v * b;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(float b, const Dual<N>& a) { return a * b; }
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& a, float b) { Dual<N> r; r.v = a.v / b;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / b; return r; }

#define DUAL_UN(name, fv, fd)                                                       \
    __device__ __forceinline__ float name(float a) { return fv; }                   \
    template <int N> __device__ __forceinline__ Dual<N> name(const Dual<N>& A) {    \
        const float a = A.v; Dual<N> r; r.v = fv; const float dv = fd;             \
        _Pragma("unroll") for (int i = 0; i < N; ++i) r.d[i] = A.d[i] * dv; return r; }
DUAL_UN(d_sqrt, sqrtf(a), 0.5f / sqrtf(a))
DUAL_UN(d_log, logf(a), 1.f / a)
DUAL_UN(d_exp, expf(a), expf(a))
DUAL_UN(d_pow2, powf(2.f, a), powf(2.f, a) * LN2_F)
// clip: torch.clamp backward passes the gradient where min <= x <= max
__device__ __forceinline__ float d_clipmin(float a, float lo) { return fmaxf(a, lo); }
template <int N> __device__ __forceinline__ Dual<N> d_clipmin(const Dual<N>& a, float lo) { return a.v >= lo ? a : mk_const<N>(lo); }
__device__ __forceinline__ float d_clip(float a, float lo, float hi) { return fminf(fmaxf(a, lo), hi); }
template <int N> __device__ __forceinline__ Dual<N> d_clip(const Dual<N>& a, float lo, float hi) {
    return a.v < lo ? mk_const<N>(lo) : (a.v > hi ? mk_const<N>(hi) : a); }
// safemath.atan2: forward atan2(x, y); backward dx = g*y/(x^2+y^2+1e-5), dy = -g*x/(...)
__device__ __forceinline__ float d_atan2(float x, float y) { return atan2f(x, y); }
template <int N> __device__ __forceinline__ Dual<N> d_atan2(const Dual<N>& x, const Dual<N>& y) {
    Dual<N> r; r.v = atan2f(x.v, y.v);
    const float den = x.v * x.v + y.v * y.v + 1e-5f;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (x.d[i] * y.v - y.d[i] * x.v) / den;
    return r; }
// torch.remainder(x, m) for m > 0 (derivative 1)
__device__ __forceinline__ float d_rem(float a, float m) { float r = fmodf(a, m); if (r != 0.f && r < 0.f) r += m; return r; }
template <int N> __device__ __forceinline__ Dual<N> d_rem(const Dual<N>& a, float m) { Dual<N> r = a; r.v = d_rem(a.v, m); return r; }
template <class T> __device__ __forceinline__ T set_val(const T& like, float v);
template <> __device__ __forceinline__ float set_val<float>(const float&, float v) { return v; }
template <int N> __device__ __forceinline__ Dual<N> set_val(const Dual<N>&, float v) { return mk_const<N>(v); }

