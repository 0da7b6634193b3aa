// The four material heads of a feature row (modules/render_modules.py:519-574) as ONE piece of code for every kernel that
// evaluates them (k_heads_fwd, and k_bounce_prep_fwd when it evaluates the heads itself): explicit fmas in a fixed order, no
// contraction left to the optimiser -- two inlined copies must give the same bits (tests compare the module path, which runs the
// heads as a launch of their own, with the training pass, which does not).
#pragma once
#include "common.hpp"

namespace nmf_heads {

constexpr int F = NMF_APP_DIM;   // 24
constexpr int O = 11;

struct HeadP {
    float diffuse_mul, diffuse_bias, tint_bias, f0_bias, rough_bias;
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// W [11][24] row-major (rows: diffuse 0-2, tint 3-5, f0 6-8, roughness 9-10), b [11]; uniform addresses (scalar loads or LDS)
__device__ __forceinline__ void heads_eval(const float (&f)[F], const float* __restrict__ W, const float* __restrict__ b, const HeadP& hp,
                                           float (&o)[O]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int j = 0; j < O; ++j) {
        float a = b[j];
#pragma unroll
        for (int k = 0; k < F; ++k) a = fmaf(W[j * F + k], f[k], a);
        float v;
        if (j < 3) v = fminf(fmaxf(sigm(fmaf(hp.diffuse_mul, a, hp.diffuse_bias)), 0.f), 1.f);
        else if (j < 6) v = sigm(a + hp.tint_bias);
        else if (j < 9) v = sigm(a + hp.f0_bias);
        else v = fminf(fmaxf(sigm(a + hp.rough_bias) * 0.5f, 1e-2f), 1.f);
        o[j] = v;
    }
}

}  // namespace nmf_heads
