// Fused BRDF MLP for gfx950: feature build (gather + ISH encodings) -> 66->64->64->4 MLP -> sigmoid,
// forward and backward, with the dense layers on the matrix cores.
// Replaces MLPBRDF.forward (reference: modules/brdf.py:177-261, modules/ish.py:94-105,
// modules/sh.py:251-308) and its autograd.
//
// Why a kernel: the only dense contraction of the hot path has N = 64 and K <= 66 -- rocBLAS runs these
// skinny GEMMs at ~2 TFLOP/s and they cost 38 % of the first end-to-end step (profiles/r01_a).  Here a
// workgroup owns a tile of 64 secondary rays that never leaves LDS: X (64x66) -> H1 -> H2 -> out.
// fp32 parity (1e-4 on radiance) rules out bf16 inputs, so the exact-fp32 MFMA
// v_mfma_f32_32x32x2_f32 is used: each of the 4 waves owns one 32x32 quadrant of a 64x64 layer output,
// its weight slice stays in VGPRs for the whole (persistent) kernel, the activation operand is read from
// LDS with an odd row stride (67) so the 32 rows of an operand column hit 32 different banks.
// The backward re-computes the forward per tile, forms dH2/dH1 with two more MFMA passes, accumulates
// dW2 = dH2^T H1 and dW0 = dH1^T X over ALL its tiles in MFMA accumulators (K = rays) and flushes them
// once per workgroup.
//
// Measured on MI355X (round 2, scratch micro-benchmark of dependent v_mfma_f32_32x32x2_f32 chains): an fp32 MFMA occupies
// its SIMD for 64 cycles and NO other VALU instruction of any wave of that SIMD issues meanwhile
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0; "MFMA wave + FMA wave" on one SIMD takes longer than the two one after the other) --
// the fp32 matrix rate equals the packed-FMA rate, the instructions seem to share the lanes.  So the floor of this kernel
// is MFMA cycles + VALU cycles, not their maximum; a second workgroup per CU only hides LDS / barrier / global latency.
#include "common.hpp"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int IN = NMF_MLP_IN;     // 66
constexpr int HID = NMF_MLP_HID;   // 64
constexpr int TR = 64;             // rays per tile
constexpr int LS = 67;             // LDS row stride (odd -> conflict-free column reads)
constexpr int K1 = IN / 2;         // 33 mfma k-steps for layer 1
constexpr int K2 = HID / 2;        // 32

__device__ __forceinline__ void ish18(float x, float y, float z, float kappa, float* o) {
    const float k = kappa + 1e-8f;
    const float a1 = expf(-1.f / k), a2 = expf(-3.f / k);
    const float xx = x * x, yy = y * y, zz = z * z;
    o[0] = 0.28209479177387814f;
    o[1] = -a1 * 0.488603f * x;
    o[2] = a1 * 0.488603f * z;
    o[3] = -a1 * 0.488603f * y;
    o[4] = a2 * 1.092548f * y * x;
    o[5] = -a2 * 1.092548f * y * z;
    o[6] = a2 * 0.315392f * (3.f * zz - 1.f);
    o[7] = -a2 * 1.092548f * x * y;
    o[8] = a2 * 0.546274f * (xx - yy);
    o[9] = 2.50334f * x * y * (xx - yy);
    o[10] = -1.77013f * y * z * (-3.f * xx + yy);
    o[11] = 0.946175f * x * y * (7.f * zz - 1.f);
    o[12] = 0.669047f * y * z * (7.f * zz - 3.f);
    o[13] = 3.70251f * (zz * zz) - 3.17358f * zz + 0.317358f;
    o[14] = 0.669047f * x * z * (7.f * zz - 3.f);
    o[15] = (0.473087f * xx - 0.473087f * yy) * (7.f * zz - 1.f);
    o[16] = 1.77013f * x * z * (xx - 3.f * yy);
    o[17] = 0.625836f * (xx * xx) - 3.755016f * xx * yy + 0.625836f * (yy * yy);
}

struct MlpW {
    const float *W0, *b0, *W2, *b2, *W4, *b4;   // [64][66],[64],[64][64],[64],[4][64],[4]
};

// fills rows of the X tile: thread t -> ray t>>2, part t&3 (0: features, 1: half, 2: diff, 3: idle)
__device__ __forceinline__ void build_x_tile(float* Xs, int64_t r0, int64_t R, const float* __restrict__ half_v,
                                             const float* __restrict__ diff_v, const float* __restrict__ feat_src,
                                             const float* __restrict__ rough_src, const int32_t* __restrict__ src_idx) {
    const int t = threadIdx.x, ray = t >> 2, part = t & 3;
    const int64_t r = r0 + ray;
    float* x = Xs + ray * LS;
    if (r >= R) {
        if (part == 0) for (int i = 0; i < 24; ++i) x[i] = 0.f;
        if (part == 1) for (int i = 24; i < 45; ++i) x[i] = 0.f;
        if (part == 2) for (int i = 45; i < 66; ++i) x[i] = 0.f;
        return;
    }
    const int64_t b = src_idx ? src_idx[r] : r;
    if (part == 0) {
        const float4* f = reinterpret_cast<const float4*>(feat_src + b * NMF_APP_DIM);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float4 v = f[i];
            x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
        }
    } else if (part < 3) {
        const float* v = (part == 1 ? half_v : diff_v) + r * 3;
        const float kappa = 1.f / (rough_src[b] + 1e-3f);
        float o[18];
        ish18(v[0], v[1], v[2], kappa, o);
        float* q = x + (part == 1 ? 24 : 45);
#pragma unroll
        for (int i = 0; i < 18; ++i) q[i] = o[i];
        q[18] = v[0]; q[19] = v[1]; q[20] = v[2];
    }
}

// A chain of N dependent v_mfma_f32_32x32x2_f32 on one accumulator runs at 64 cycles per instruction at best.  Left to
// itself the compiler issues each pair's ds_read right after the previous pair and waits for it (lgkmcnt(0)) in front of
// the next one, which exposes the LDS latency once per pair (~170 instead of 128 cycles, ISA of round 2).  Here the
// operands of chunk c+1 are requested before the instructions of chunk c issue; the empty asm keeps the loads above it.
constexpr int CH = 8;
template <int N, class FA, class FB>
__device__ __forceinline__ void mfma_chain(floatx16& acc, FA load_a, FB load_b) {
    constexpr int NC = (N + CH - 1) / CH;
    float a[2][CH], b[2][CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        a[0][i] = load_a(i);
        b[0][i] = load_b(i);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c + 1 < NC) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if ((c + 1) * CH + i < N) {
                    a[(c + 1) & 1][i] = load_a((c + 1) * CH + i);
                    b[(c + 1) & 1][i] = load_b((c + 1) * CH + i);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (c * CH + i < N) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][i], b[c & 1][i], acc, 0, 0, 0);
    }
}

// one 32x32 quadrant of  act_out = relu(act_in[64 x K] * W^T + bias)  on the matrix core.
// wreg[kk] = W[32*wc + (lane&31)][2*kk + (lane>>5)]
template <int KS>
__device__ __forceinline__ void layer_quadrant(const float* in_s, float* out_s, const float (&wreg)[KS],
                                               const float* __restrict__ bias, int wr, int wc, int lane, bool relu) {
    floatx16 acc = {0};
    const float* arow = in_s + (32 * wr + (lane & 31)) * LS + (lane >> 5);
    float a[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) a[kk] = arow[2 * kk];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], wreg[kk], acc, 0, 0, 0);
    const int col = 32 * wc + (lane & 31);
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[r] + bv;
        out_s[row * LS + col] = relu ? fmaxf(v, 0.f) : v;
    }
}

template <int KS>
__device__ __forceinline__ void load_wreg(float (&wreg)[KS], const float* __restrict__ W, int ldw, int kmax, int wc,
                                          int lane) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int k = 2 * kk + (lane >> 5);
        wreg[kk] = k < kmax ? W[(32 * wc + (lane & 31)) * ldw + k] : 0.f;
    }
}

__global__ void __launch_bounds__(256) k_brdf_mlp_fwd(MlpW w, const float* __restrict__ half_v,
                                                      const float* __restrict__ diff_v,
                                                      const float* __restrict__ feat_src,
                                                      const float* __restrict__ rough_src,
                                                      const int32_t* __restrict__ src_idx, int64_t R, float out_bias,
                                                      float* __restrict__ out) {
    __shared__ float Xs[TR * LS], H1s[TR * LS], H2s[TR * LS];
    __shared__ float W4s[4 * HID], b4s[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
    float w0reg[K1], w2reg[K2];
    load_wreg<K1>(w0reg, w.W0, IN, IN, wc, lane);
    load_wreg<K2>(w2reg, w.W2, HID, HID, wc, lane);
    W4s[t] = w.W4[t];
    if (t < 4) b4s[t] = w.b4[t];
    const int64_t n_tiles = (R + TR - 1) / TR;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r0 = tile * TR;
        __syncthreads();
        build_x_tile(Xs, r0, R, half_v, diff_v, feat_src, rough_src, src_idx);
        __syncthreads();
        layer_quadrant<K1>(Xs, H1s, w0reg, w.b0, wr, wc, lane, true);
        __syncthreads();
        layer_quadrant<K2>(H1s, H2s, w2reg, w.b2, wr, wc, lane, true);
        __syncthreads();
        const int ray = t >> 2, j = t & 3;
        if (j < 3 && r0 + ray < R) {
            float o = b4s[j];
            const float* h = H2s + ray * LS;
#pragma unroll 8
            for (int k = 0; k < HID; ++k) o += h[k] * W4s[j * HID + k];
            out[(r0 + ray) * 3 + j] = 1.f / (1.f + expf(-(o + out_bias)));      // modules/brdf.py:131
        }
    }
}

// dW (64 x 64 block) += A^T B over the 64 rays of the tile:  A[ray][i], B[ray][j] both LDS tiles (stride LS)
__device__ __forceinline__ void outer_quadrant(floatx16& acc, const float* a_s, const float* b_s, int wr, int wc,
                                               int lane) {
    const float* ap = a_s + (lane >> 5) * LS + 32 * wr + (lane & 31);
    const float* bp = b_s + (lane >> 5) * LS + 32 * wc + (lane & 31);
    mfma_chain<TR / 2>(acc, [&](int kk) { return ap[2 * kk * LS]; }, [&](int kk) { return bp[2 * kk * LS]; });
}

__global__ void __launch_bounds__(256, 2) k_brdf_mlp_bwd(MlpW w, const float* __restrict__ half_v,
                                                      const float* __restrict__ diff_v,
                                                      const float* __restrict__ feat_src,
                                                      const float* __restrict__ rough_src,
                                                      const int32_t* __restrict__ src_idx, int64_t R, float out_bias,
                                                      const float* __restrict__ d_out, float* __restrict__ d_xfeat,
                                                      float* __restrict__ gW0, float* __restrict__ gb0,
                                                      float* __restrict__ gW2, float* __restrict__ gb2,
                                                      float* __restrict__ gW4, float* __restrict__ gb4) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                    // [64][67]
    float* H1s = Xs + TR * LS;           // H1, later dH1
    float* H2s = H1s + TR * LS;          // H2, later dH2
    float* W2s = H2s + TR * LS;          // [64][64]  (row = unit of layer 2)
    float* W0f = W2s + HID * HID;        // [64][24]  feature columns of W0
    float* W4s = W0f + HID * 24;         // [4][64]
    float* dOs = W4s + 4 * HID;          // [64][4]
    float* b4s = dOs + TR * 4;           // [4]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
    float w0reg[K1], w2reg[K2];
    load_wreg<K1>(w0reg, w.W0, IN, IN, wc, lane);
    load_wreg<K2>(w2reg, w.W2, HID, HID, wc, lane);
    for (int i = t; i < HID * HID; i += 256) W2s[i] = w.W2[i];
    for (int i = t; i < HID * 24; i += 256) W0f[i] = w.W0[(i / 24) * IN + (i % 24)];
    W4s[t] = w.W4[t];
    if (t < 4) b4s[t] = w.b4[t];
    // persistent accumulators
    floatx16 accW2 = {0}, accW0 = {0};
    float accW4 = 0.f;                   // thread (j = t>>6, k = t&63)
    float accb = 0.f;                    // 64<=t<128: db2[t-64]; 128<=t<132: db4
    float accW0tail = 0.f;               // waves 1, 3: dW0[lane][64 + (wave >> 1)]
    float accb0 = 0.f;                   // wave 1: db0[lane]
    const int64_t n_tiles = (R + TR - 1) / TR;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r0 = tile * TR;
        __syncthreads();
        build_x_tile(Xs, r0, R, half_v, diff_v, feat_src, rough_src, src_idx);
        __syncthreads();
        layer_quadrant<K1>(Xs, H1s, w0reg, w.b0, wr, wc, lane, true);
        __syncthreads();
        layer_quadrant<K2>(H1s, H2s, w2reg, w.b2, wr, wc, lane, true);
        __syncthreads();
        {   // output layer + adjoint of the sigmoid
            const int ray = t >> 2, j = t & 3;
            float g = 0.f;
            if (j < 3 && r0 + ray < R) {
                float o = b4s[j];
                const float* h = H2s + ray * LS;
#pragma unroll 8
                for (int k = 0; k < HID; ++k) o += h[k] * W4s[j * HID + k];
                const float s = 1.f / (1.f + expf(-(o + out_bias)));
                g = d_out[(r0 + ray) * 3 + j] * s * (1.f - s);
            }
            dOs[ray * 4 + j] = g;
        }
        __syncthreads();
        {   // dW4[j][k] += sum_ray dO[ray][j] H2[ray][k]
            const int j = t >> 6, k = t & 63;
            float a = 0.f;
#pragma unroll 8
            for (int ray = 0; ray < TR; ++ray) a += dOs[ray * 4 + j] * H2s[ray * LS + k];
            accW4 += a;
            if (t >= 128 && t < 132) {
                float b = 0.f;
                for (int ray = 0; ray < TR; ++ray) b += dOs[ray * 4 + (t - 128)];
                accb += b;
            }
        }
        __syncthreads();
        {   // dH2 = (dO W4) * [H2 > 0]   (in place)
            const int ray = t >> 2, k0 = (t & 3) * 16;
            const float d0 = dOs[ray * 4], d1 = dOs[ray * 4 + 1], d2 = dOs[ray * 4 + 2];
#pragma unroll
            for (int k = k0; k < k0 + 16; ++k) {
                float* h = H2s + ray * LS + k;
                const float v = d0 * W4s[k] + d1 * W4s[HID + k] + d2 * W4s[2 * HID + k];
                *h = *h > 0.f ? v : 0.f;
            }
        }
        __syncthreads();
        outer_quadrant(accW2, H2s, H1s, wr, wc, lane);                 // dW2 += dH2^T H1
        if (t >= 64 && t < 128) {                                      // db2
            float b = 0.f;
            for (int ray = 0; ray < TR; ++ray) b += H2s[ray * LS + (t - 64)];
            accb += b;
        }
        floatx16 acc = {0};                                            // dH1 = dH2 W2 (pre-mask), quadrant (wr, wc)
        {
            const float* arow = H2s + (32 * wr + (lane & 31)) * LS + (lane >> 5);
            const float* brow = W2s + (lane >> 5) * HID + 32 * wc + (lane & 31);
            mfma_chain<K2>(acc, [&](int kk) { return arow[2 * kk]; }, [&](int kk) { return brow[2 * kk * HID]; });
        }
        __syncthreads();                                               // everyone is done reading H1
        {
            const int col = 32 * wc + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float* h = H1s + row * LS + col;
                *h = *h > 0.f ? acc[r] : 0.f;                          // relu'
            }
        }
        __syncthreads();
        outer_quadrant(accW0, H1s, Xs, wr, wc, lane);                  // dW0[:, 0:64] += dH1^T X[:, 0:64]
        if (wc == 0) {
            // dX[:, 0:24] = dH1 W0[:, 0:24] -> adjoint of the gathered feature row, on the matrix core: waves 0 and 2 own
            // the ray blocks 0-31 / 32-63 of a 64 x 32 product (feature columns 24..31 are padding), K = 64 units
            floatx16 dx = {0};
            const float* arow = H1s + (32 * wr + (lane & 31)) * LS + (lane >> 5);
            const int col = lane & 31;
            const float* brow = W0f + (lane >> 5) * 24 + (col < 24 ? col : 0);
            const float bmask = col < 24 ? 1.f : 0.f;
            mfma_chain<K2>(dx, [&](int kk) { return arow[2 * kk]; }, [&](int kk) { return brow[2 * kk * 24] * bmask; });
            if (col < 24) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = r0 + 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < R) d_xfeat[row * 24 + col] = dx[r];
                }
            }
        } else {
            // meanwhile waves 1 and 3: the two trailing input columns 64, 65 of dW0, then db0 on wave 1
            const int i = lane, c = 64 + wr;
            float a = 0.f;
#pragma unroll 8
            for (int ray = 0; ray < TR; ++ray) a += H1s[ray * LS + i] * Xs[ray * LS + c];
            accW0tail += a;
            if (wr == 0) {
                float b = 0.f;
#pragma unroll 8
                for (int ray = 0; ray < TR; ++ray) b += H1s[ray * LS + lane];
                accb0 += b;
            }
        }
    }
    // flush the per-workgroup weight gradients
    {
        const int col = 32 * wc + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            atomicAdd(gW2 + row * HID + col, accW2[r]);
            atomicAdd(gW0 + row * IN + col, accW0[r]);
        }
        if (wc == 1) atomicAdd(gW0 + lane * IN + 64 + wr, accW0tail);
        atomicAdd(gW4 + (t >> 6) * HID + (t & 63), accW4);
        if (wave == 1) atomicAdd(gb0 + lane, accb0);
        if (t >= 64 && t < 128) atomicAdd(gb2 + (t - 64), accb);
        else if (t >= 128 && t < 132) atomicAdd(gb4 + (t - 128), accb);
    }
}

constexpr int BWD_LDS = (3 * TR * LS + HID * HID + HID * 24 + 4 * HID + TR * 4 + 4) * sizeof(float);

}  // namespace

static int check_w(const float* const* p) {
    for (int i = 0; i < 6; ++i)
        if (!p[i]) return 0;
    return 1;
}

extern "C" int nmf_brdf_mlp_fwd(const float* W0, const float* b0, const float* W2, const float* b2, const float* W4,
                                const float* b4, const float* half_vec, const float* diff_vec, const float* feat_src,
                                const float* rough_src, const int32_t* src_idx, int64_t R, float out_bias, float* out,
                                int32_t max_workgroups, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_brdf_mlp_fwd: R < 0");
    if (R == 0) return NMF_OK;
    const float* ws[6] = {W0, b0, W2, b2, W4, b4};
    NMF_REQUIRE(check_w(ws) && half_vec && diff_vec && feat_src && rough_src && out, NMF_EINVAL, "nmf_brdf_mlp_fwd: null");
    MlpW w{W0, b0, W2, b2, W4, b4};
    const int64_t tiles = cdiv(R, TR);
    int64_t cap = 1024;
    if (max_workgroups > 0 && max_workgroups < cap) cap = max_workgroups;
    const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
    hipLaunchKernelGGL(k_brdf_mlp_fwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, half_vec, diff_vec, feat_src,
                       rough_src, src_idx, R, out_bias, out);
    NMF_CHECK_LAUNCH("nmf_brdf_mlp_fwd");
    return NMF_OK;
}

extern "C" int nmf_brdf_mlp_bwd(const float* W0, const float* b0, const float* W2, const float* b2, const float* W4,
                                const float* b4, const float* half_vec, const float* diff_vec, const float* feat_src,
                                const float* rough_src, const int32_t* src_idx, int64_t R, float out_bias,
                                const float* d_out, float* d_xfeat, float* gW0, float* gb0, float* gW2, float* gb2,
                                float* gW4, float* gb4, int32_t max_workgroups, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_brdf_mlp_bwd: R < 0");
    if (R == 0) return NMF_OK;
    const float* ws[6] = {W0, b0, W2, b2, W4, b4};
    NMF_REQUIRE(check_w(ws) && half_vec && diff_vec && feat_src && rough_src && d_out && d_xfeat && gW0 && gb0 && gW2 &&
                    gb2 && gW4 && gb4,
                NMF_EINVAL, "nmf_brdf_mlp_bwd: null");
    MlpW w{W0, b0, W2, b2, W4, b4};
    hipError_t e = hipFuncSetAttribute((const void*)k_brdf_mlp_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    if (e != hipSuccess) return nmf_fail((int)e, "nmf_brdf_mlp_bwd: hipFuncSetAttribute");
    const int64_t tiles = cdiv(R, TR);
    // The kernel is compiled for two workgroups per CU (<= 256 registers, 2 x 76 KB of LDS).  The second one hides barrier
    // and LDS latency of the first but doubles the weight staging and the 8.5 k flush atomics per workgroup, so it only
    // pays for long launches (measured: 242 k rays 210 -> 187 us, 46 k rays 60 -> 68 us).
    int64_t cap = tiles >= 2048 ? 512 : 256;
    if (max_workgroups > 0 && max_workgroups < cap) cap = max_workgroups;
    const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
    hipLaunchKernelGGL(k_brdf_mlp_bwd, dim3(grid), dim3(256), BWD_LDS, (hipStream_t)stream, w, half_vec, diff_vec,
                       feat_src, rough_src, src_idx, R, out_bias, d_out, d_xfeat, gW0, gb0, gW2, gb2, gW4, gb4);
    NMF_CHECK_LAUNCH("nmf_brdf_mlp_bwd");
    return NMF_OK;
}
