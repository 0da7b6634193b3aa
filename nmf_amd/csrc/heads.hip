// Material heads for gfx950: the four single-layer heads of RandHydraMLPDiffuse
// (reference: modules/render_modules.py:519-574 with pospe=-1, feape=0, num_layers=1, std=0) evaluated
// as ONE 24 -> 11 product per kept sample with the activations fused:
//   albedo = clip(sigmoid(diffuse_mul * a[0:3] + diffuse_bias), 0, 1)      tint = sigmoid(a[3:6] + tint_bias)
//   f0     = sigmoid(a[6:9] + f0_bias)                                     r = clip(sigmoid(a[9:11] + rough_bias)/2, 0.01, 1)
// rocBLAS runs these N=3 GEMMs over ~1 M samples at ~1.6 ms each (12 calls per level, profiles/r01_b);
// here the forward is one streaming pass and the weight gradient is reduced per workgroup through LDS.
#include "heads_eval.hpp"
#include "rows_bwd.hpp"

namespace {

using nmf_heads::F;
using nmf_heads::O;
using nmf_heads::HeadP;
using nmf_heads::sigm;

__device__ __forceinline__ void load_feat(const float* __restrict__ feat, int64_t m, float (&f)[F]) {
    const float4* q = reinterpret_cast<const float4*>(feat + m * F);
#pragma unroll
    for (int i = 0; i < F / 4; ++i) {
        const float4 v = q[i];
        f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
    }
}

// W [11][24] row-major (rows: diffuse 0-2, tint 3-5, f0 6-8, roughness 9-10), b [11]
__global__ void __launch_bounds__(256) k_heads_fwd(const float* __restrict__ feat, int64_t M, const float* __restrict__ W,
                                                   const float* __restrict__ b, HeadP hp, float* __restrict__ out) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float f[F], v[O];
    load_feat(feat, m, f);
    nmf_heads::heads_eval(f, W, b, hp, v);
    float* o = out + m * O;
#pragma unroll
    for (int j = 0; j < O; ++j) o[j] = v[j];
}

__global__ void __launch_bounds__(256) k_heads_bwd(const float* __restrict__ feat, int64_t M, const float* __restrict__ W,
                                                   const float* __restrict__ b, HeadP hp,
                                                   const float* __restrict__ d_out,
                                                   const float* d_feat_add, float* d_feat,   // may be the same buffer
                                                   float* __restrict__ gW, float* __restrict__ gb, nmf_rows::RowsBwdIn rin) {
    // rin.bidx != NULL (nmf_bounce_prep_heads_bwd): the adjoint of the heads' outputs and the feature-row adjoint that is added are
    // not read -- they are what nmf_bounce_prep_bwd would have written for this row (row_inputs = 2), computed here from its inputs
    // (nmf_rows::prep_bwd_row, which also writes d_normals): one launch less per level on the backward's main chain
    __shared__ float s_da[256 * (O + 1)];
    __shared__ float s_f[256 * (F + 1)];
    const int t = threadIdx.x;
    // Weight gradient gW [11][24] += da^T [11 x rows] f [rows x 24] (and gb = column sums of da: a 25th column of ones) on the
    // matrix cores: wave w takes rows 64 w .. 64 w + 63 of a 256-row block as sixteen K = 4 steps of v_mfma_f32_16x16x4_f32
    // (M = 16 >= 11 outputs, N = 2 x 16 >= 25 columns).  R3 gave every thread one of the 264 entries and walked the 256 rows one
    // by one, then 8 threads a second entry, then 11 threads the bias sums: three serial passes of 256 LDS round trips, 14 us for
    // ANY number of rows (tools/fixed_cost.py) on the serial tail of the step.
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    const int lane = t & 63, wv = t >> 6, ci = lane & 15, kg = lane >> 4;
    floatx4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
    // W and b through LDS (broadcast reads): as uniform global addresses they became ~200 scalar loads that the 100 scalar
    // registers cannot hold at once -- 130 waits on the scalar cache per row block and 450 lane moves of spilled scalars, most
    // of the 14 us this launch took for any number of rows
    __shared__ float s_W[O * F];
    __shared__ float s_b[O];
    for (int i = t; i < O * F; i += 256) s_W[i] = W[i];
    if (t < O) s_b[t] = b[t];
    __syncthreads();
    const int64_t n_it = (M + 255) / 256;
    for (int64_t it = blockIdx.x; it < n_it; it += gridDim.x) {
        const int64_t m = it * 256 + t;
        float f[F], da[O];
        if (m < M) {
            load_feat(feat, m, f);
            float gq[O];
            float4 addv[F / 4];
            if (rin.bidx) nmf_rows::prep_bwd_row(rin, m, gq, addv);
#pragma unroll
            for (int j = 0; j < O; ++j) {
                float a = s_b[j];
#pragma unroll
                for (int k = 0; k < F; ++k) a += s_W[j * F + k] * f[k];
                const float g = rin.bidx ? gq[j] : d_out[m * O + j];
                float d;
                if (j < 3) {
                    const float s = sigm(hp.diffuse_mul * a + hp.diffuse_bias);
                    d = g * s * (1.f - s) * hp.diffuse_mul;             // clip(0,1) never binds on a sigmoid
                } else if (j < 9) {
                    const float s = sigm(a + (j < 6 ? hp.tint_bias : hp.f0_bias));
                    d = g * s * (1.f - s);
                } else {
                    const float s = sigm(a + hp.rough_bias);
                    const float r = 0.5f * s;
                    d = (r >= 1e-2f && r <= 1.f) ? g * 0.5f * s * (1.f - s) : 0.f;
                }
                da[j] = d;
            }
            float df[F];
#pragma unroll
            for (int k = 0; k < F; ++k) df[k] = 0.f;
#pragma unroll
            for (int j = 0; j < O; ++j)
#pragma unroll
                for (int k = 0; k < F; ++k) df[k] += da[j] * s_W[j * F + k];
            float4* q = reinterpret_cast<float4*>(d_feat + m * F);
            if (rin.bidx) {
#pragma unroll
                for (int i = 0; i < F / 4; ++i) {
                    df[4 * i] = addv[i].x + df[4 * i]; df[4 * i + 1] = addv[i].y + df[4 * i + 1];
                    df[4 * i + 2] = addv[i].z + df[4 * i + 2]; df[4 * i + 3] = addv[i].w + df[4 * i + 3];
                }
            } else if (d_feat_add) {                                    // another adjoint of the same rows, added here (may alias d_feat)
                const float4* qa = reinterpret_cast<const float4*>(d_feat_add + m * F);
#pragma unroll
                for (int i = 0; i < F / 4; ++i) {
                    const float4 a = qa[i];
                    df[4 * i] = a.x + df[4 * i]; df[4 * i + 1] = a.y + df[4 * i + 1];
                    df[4 * i + 2] = a.z + df[4 * i + 2]; df[4 * i + 3] = a.w + df[4 * i + 3];
                }
            }
#pragma unroll
            for (int i = 0; i < F / 4; ++i) q[i] = make_float4(df[4 * i], df[4 * i + 1], df[4 * i + 2], df[4 * i + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < F; ++k) f[k] = 0.f;
#pragma unroll
            for (int j = 0; j < O; ++j) da[j] = 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < O; ++j) s_da[t * (O + 1) + j] = da[j];
#pragma unroll
        for (int k = 0; k < F; ++k) s_f[t * (F + 1) + k] = f[k];
        __syncthreads();
#pragma unroll 4
        for (int r0 = 64 * wv; r0 < 64 * wv + 64; r0 += 4) {
            const int row = r0 + kg;
            const float a = ci < O ? s_da[row * (O + 1) + ci] : 0.f;
            const float b0 = s_f[row * (F + 1) + ci];
            const float b1 = ci < F - 16 ? s_f[row * (F + 1) + 16 + ci] : (ci == F - 16 ? 1.f : 0.f);
            g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, g1, 0, 0, 0);
        }
    }
    // the four waves' tiles meet in LDS (register r of a lane: output 4 (lane >> 4) + r, column lane & 15 of its block), one atomic
    // per entry and workgroup
    __syncthreads();
    float* red = s_f;                                   // [4 waves][2 blocks][4 registers][64 lanes]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wv * 512 + r * 64 + lane] = g0[r];
        red[wv * 512 + 256 + r * 64 + lane] = g1[r];
    }
    __syncthreads();
#pragma unroll
    for (int e = t; e < 512; e += 256) {
        const float v = (red[e] + red[512 + e]) + (red[1024 + e] + red[1536 + e]);
        const int blk = e >> 8, r = (e >> 6) & 3, ln = e & 63;
        const int oi = 4 * (ln >> 4) + r, c = ln & 15;
        if (oi < O) {
            if (blk == 0) atomicAdd(gW + oi * F + c, v);
            else if (c < F - 16) atomicAdd(gW + oi * F + 16 + c, v);
            else if (c == F - 16) atomicAdd(gb + oi, v);
        }
    }
}

}  // namespace

extern "C" int nmf_heads_fwd(const float* feat, int64_t M, const float* W, const float* b, float diffuse_mul,
                             float diffuse_bias, float tint_bias, float f0_bias, float rough_bias, float* out,
                             void* stream) {
    NMF_REQUIRE(M >= 0, NMF_EINVAL, "nmf_heads_fwd: M < 0");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(feat && W && b && out, NMF_EINVAL, "nmf_heads_fwd: null");
    HeadP hp{diffuse_mul, diffuse_bias, tint_bias, f0_bias, rough_bias};
    NMF_LAUNCH(k_heads_fwd, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, feat, M, W, b, hp, out);
    NMF_CHECK_LAUNCH("nmf_heads_fwd");
    return NMF_OK;
}

extern "C" int nmf_heads_bwd(const float* feat, int64_t M, const float* W, const float* b, float diffuse_mul,
                             float diffuse_bias, float tint_bias, float f0_bias, float rough_bias, const float* d_out,
                             const float* d_feat_add, float* d_feat, float* gW, float* gb, void* stream) {
    NMF_REQUIRE(M >= 0, NMF_EINVAL, "nmf_heads_bwd: M < 0");
    if (M == 0) return NMF_OK;
    NMF_REQUIRE(feat && W && b && d_out && d_feat && gW && gb, NMF_EINVAL, "nmf_heads_bwd: null");
    HeadP hp{diffuse_mul, diffuse_bias, tint_bias, f0_bias, rough_bias};
    const int64_t n_it = cdiv(M, 256);
    const unsigned grid = (unsigned)(n_it < 1024 ? n_it : 1024);
    NMF_LAUNCH(k_heads_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, feat, M, W, b, hp, d_out, d_feat_add, d_feat, gW, gb,
                       nmf_rows::RowsBwdIn{});
    NMF_CHECK_LAUNCH("nmf_heads_bwd");
    return NMF_OK;
}

extern "C" int nmf_bounce_prep_heads_bwd(const int32_t* bidx, int64_t Mb, const float* normals, const float* heads,
                                         const int32_t* ray_id, const float* rays, const float* conv, float min_rough,
                                         int32_t detach_normals, const float* dN, const float* dr1, const float* df0,
                                         const float* ddiffuse, const int32_t row_strides[4], const float* dfeat, const float* app,
                                         const float* head_W, const float* head_b, float diffuse_mul, float diffuse_bias,
                                         float tint_bias, float f0_bias, float rough_bias, float* d_normals, float* d_app,
                                         float* g_head_W, float* g_head_b, void* stream) {
    NMF_REQUIRE(Mb >= 0, NMF_EINVAL, "nmf_bounce_prep_heads_bwd: Mb < 0");
    if (Mb == 0) return NMF_OK;
    NMF_REQUIRE(bidx && normals && heads && ray_id && rays && conv && app && head_W && head_b && d_normals && d_app && g_head_W &&
                    g_head_b, NMF_EINVAL, "nmf_bounce_prep_heads_bwd: null");
    HeadP hp{diffuse_mul, diffuse_bias, tint_bias, f0_bias, rough_bias};
    nmf_rows::RowsBwdIn rin{bidx, normals, heads, ray_id, rays, nmf_rows::Conv{conv}, min_rough, (int)detach_normals, dN, dr1, df0, ddiffuse,
                            row_strides ? row_strides[0] : 3, row_strides ? row_strides[1] : 1, row_strides ? row_strides[2] : 3,
                            row_strides ? row_strides[3] : 3, dfeat, d_normals};
    const int64_t n_it = cdiv(Mb, 256);
    const unsigned grid = (unsigned)(n_it < 1024 ? n_it : 1024);
    NMF_LAUNCH(k_heads_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, app, Mb, head_W, head_b, hp, nullptr, nullptr, d_app,
                       g_head_W, g_head_b, rin);
    NMF_CHECK_LAUNCH("nmf_bounce_prep_heads_bwd");
    return NMF_OK;
}
