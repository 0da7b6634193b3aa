// Shared helpers for the gfx950 kernels of libnmf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/nmf_hip.h"

#define NMF_WAVE 64

extern thread_local char nmf_err_buf[256];

static inline int nmf_fail(int code, const char* what) {
    snprintf(nmf_err_buf, sizeof(nmf_err_buf), "%s (code %d)", what, code);
    return code;
}

#define NMF_REQUIRE(cond, code, what) \
    do {                              \
        if (!(cond)) return nmf_fail((code), (what)); \
    } while (0)

// Launch check: records a readable message and returns the hipError_t (positive) on failure.
#define NMF_CHECK_LAUNCH(name)                                            \
    do {                                                                  \
        hipError_t e_ = hipGetLastError();                                \
        if (e_ != hipSuccess) {                                           \
            snprintf(nmf_err_buf, sizeof(nmf_err_buf), "%s: %s", (name), hipGetErrorString(e_)); \
            return (int)e_;                                               \
        }                                                                 \
    } while (0)

// Every kernel of the library is launched through this: with a probe installed (nmf_set_launch_probe) the launch is bracketed by
// calls the measuring host turns into HIP events on the launching stream; without one it is two pointer tests.
extern nmf_launch_probe_fn nmf_launch_probe;
#define NMF_LAUNCH_NAMED(name, kernel, grid, block, lds, stream, ...)                       \
    do {                                                                                    \
        nmf_launch_probe_fn probe_ = nmf_launch_probe;                                      \
        const char* name_ = (name);                                                         \
        if (probe_) probe_(name_, (void*)(stream), 0);                                      \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                  \
        if (probe_) probe_(name_, (void*)(stream), 1);                                      \
    } while (0)
#define NMF_LAUNCH(kernel, grid, block, lds, stream, ...) NMF_LAUNCH_NAMED(#kernel, kernel, grid, block, lds, stream, __VA_ARGS__)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Non-contracted fp32 arithmetic: bookkeeping that must be bit-exact against the CPU oracle
// (sample positions, in-box / occupancy tests, bounce counts) uses these so the compiler can
// never fuse a*b+c into an fma.
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }

// ---- Philox4x32-10 (counter-based RNG for in-kernel jitter) -------------------------------
struct Philox {
    uint32_t k0, k1;
    __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ static void round_(uint32_t (&c)[4], uint32_t a, uint32_t b) {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ a, n1 = lo1, n2 = hi0 ^ c[3] ^ b, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __device__ void operator()(uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) const {
        uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round_(c, a, b);
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};
__device__ __forceinline__ float u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// ---- wave-level helpers (64 lanes) ------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// The same inclusive sum with DPP lane moves instead of ds_bpermute (six dependent LDS-crossbar round trips, ~100 cycles
// each, were a third of a marcher round): Kogge-Stone inside the rows of 16 lanes (row_shr:1,2,4,8, zeros shifted in),
// then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into lanes 32-63 (row_bcast:31).  A different
// association than wave_incl_scan: use it where the partial sums are EXACT in float64 (the marcher's step lengths), so
// that the result does not depend on the order.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_incl_scan_dpp(double v) {
    v += dpp_f64<0x111, 0xf, 0xf>(v);      // row_shr:1
    v += dpp_f64<0x112, 0xf, 0xf>(v);      // row_shr:2
    v += dpp_f64<0x114, 0xf, 0xf>(v);      // row_shr:4
    v += dpp_f64<0x118, 0xf, 0xf>(v);      // row_shr:8
    v += dpp_f64<0x142, 0xa, 0xf>(v);      // row_bcast:15 into rows 1 and 3
    v += dpp_f64<0x143, 0xc, 0xf>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ double wave_incl_scan(double v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// two sizes + a sequence number into mapped, coherent host memory (nmf_host_alloc_mapped): the producing kernel itself
// publishes what the host is waiting for (nmf_wait_seq) -- no extra one-thread launch on the dependency chain
__device__ __forceinline__ void publish_sizes(int64_t* dst, int64_t a, int64_t b, int64_t seq) {
    if (!dst) return;
    volatile int64_t* d = dst;
    d[0] = a;
    d[1] = b;
    __threadfence_system();
    d[2] = seq;
}
