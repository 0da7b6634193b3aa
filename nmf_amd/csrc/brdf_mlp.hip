// Fused BRDF MLP for gfx950: feature build (gather + ISH encodings) -> 66->64->64->4 MLP -> sigmoid,
// forward and backward, with the dense layers on the bf16 matrix cores at fp32-class accuracy.
// Replaces MLPBRDF.forward (reference: modules/brdf.py:177-261, modules/ish.py:94-105,
// modules/sh.py:251-308) and its autograd.
//
// Round 3 design (rounds 1-2: exact-fp32 v_mfma_f32_32x32x2_f32, a 64-ray tile shared by four waves, 11 barrier
// phases per tile: 26-29 % MFMA-busy at 0.9 waves per SIMD -- a latency machine).
//
// * Arithmetic: an fp32 operand v is carried as a sum of bf16 numbers (v = h + m [+ l], each the bf16 rounding of what
//   the ones before left over: 2^-18 |v| left after two terms, 2^-27 after three) and a matrix product is a handful of
//   v_mfma_f32_32x32x16_bf16 with fp32 accumulation (32 matrix-core cycles per K = 16; the fp32-input instruction
//   needs 512).  The FORWARD uses three terms and the six products down to 2^-18 (hh hm mh mm hl lh): its
//   pre-activations are fp32-accurate, which matters because a ReLU decides on their sign -- with two terms a few units
//   in 10^5 flip against an fp32 evaluation and the gradients of those rays change by finite amounts.  The forward
//   therefore also writes the two ReLU masks (16 bytes per ray), and the BACKWARD takes them together with the forward's
//   output: what it recomputes (H1, H2 as VALUES in dW2 / dW4) and its adjoint products use two terms and three
//   products (relative error 4e-6), nothing discontinuous depends on them.
// * Work split: ONE WAVE owns a tile of 32 rays through every phase -- no workgroup barrier inside the loop.  Its
//   activations live in a wave-private LDS region as row-major bf16 planes [ray][unit] (row stride an odd number of
//   16-byte slots: conflict-free ds_read_b128 operand reads); the weights (planes of W0 | b0, W2, W2^T,
//   W0[:, :24]^T) are staged once per workgroup in LDS and shared by its waves.
// * Layer products are "swapped" (D^T[unit][ray] = W . act^T): a lane then holds 4 CONSECUTIVE units of its ray per
//   accumulator quad = one ds_write_b64 into the row-major plane the next product reads.
// * Weight gradients (K = rays) need the activations with the unit per lane and the rays along the registers.  R5: that
//   transposition is the LDS read itself -- ds_read_b64_tr_b16 (gfx950) hands a group of 16 lanes the transpose of the
//   4 x 16 block of 16-bit values they address (lanes 4k .. 4k+3 supply row k, lane n receives column n: measured,
//   tools/ub/tr_read.hip), so two of them deliver 8 consecutive rays of one unit per lane = an MFMA operand of the dW
//   products straight from the row-major planes.  (R3-R4 transposed on the matrix core with 0/1 selector matrices: 34 of
//   the 186 matrix instructions of a tile and a conversion of every transposed value.)
// * dW2, dW0 accumulate in MFMA accumulators over ALL tiles of a wave (the 5 trailing input columns through a
//   transposed product whose 4 useful rows are kept), dW4 / db2 / db4 on the VALU; one LDS reduction over the waves
//   and one atomic flush per workgroup at the end.  b0 rides as an input column that is 1.0.
#include "common.hpp"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Measurement switches of the backward (a build with -DNMF_MLP_KNOCK=bits leaves a phase out; results are then wrong, the time
// difference is that phase's cost -- the kernel has no profiler view finer than a launch): 1 feature-row adjoint, 2 ISH
// encoding, 4 layer 2 as values + dW4, 8 dW0's trailing columns, 16 dW0, 32 dW2, 64 no tile loop, 128 return behind the prologue.
// R4, 242 k rays, 95 us: 14 / 2 / 6 / 1 / 5.5 / 8 us; prologue 10.3 us (4.5 of it the weight conversion), epilogue 3.3 us, the
// first tile of a wave 17 us against 8.2 us for the later ones (tools/mlp_bench.py with NMF_HIP_LIB=<variant>).
#ifndef NMF_MLP_KNOCK
#define NMF_MLP_KNOCK 0
#endif
constexpr int IN = NMF_MLP_IN;     // 66
constexpr int HID = NMF_MLP_HID;   // 64
constexpr int RT = 32;             // rays per wave tile
constexpr int SXB = 176;           // bytes per row of an input plane  (80 columns + 8 pad, bf16): 11 slots of 16 B
constexpr int SHB = 144;           // bytes per row of a 64-wide plane (64 + 8 pad): 9 slots
// input columns of the LDS image: 0-20 half vector (18 ISH + xyz) | 21-44 features | 45 = 1.0 (b0) | 46-47 zero |
// 48-68 diff vector (18 ISH + xyz) | 69-79 zero.  Lane half 0 of a wave builds columns 0-47, half 1 columns 48-79 (the
// encoded direction comes first in both, so the two halves run the same code on different inputs).
constexpr int COL_FEAT = 21, COL_BIAS = 45, COL_DIFF = 48;
constexpr int PX = RT * SXB, PH = RT * SHB;           // one plane of a wave's input tile / of a 64-wide tile
constexpr int PW0 = HID * SXB, PW2 = HID * SHB;       // one plane of W0 | b0 / of W2

// LDS maps (bytes).  Forward: three planes of everything.
constexpr int F_W0 = 0, F_W2 = F_W0 + 3 * PW0, F_W4 = F_W2 + 3 * PW2, F_B2 = F_W4 + 3 * HID * 4, F_B4 = F_B2 + HID * 4;
constexpr int FWD_SHARED = F_B4 + 16;
constexpr int FWD_PRIVATE = 3 * PX;                   // the hidden planes overwrite the input planes
constexpr int FWD_WAVES = 4;
constexpr int FWD_LDS = FWD_SHARED + FWD_WAVES * FWD_PRIVATE;
// Backward: two planes.
constexpr int B_W0 = 0, B_W2 = B_W0 + 2 * PW0, B_W2T = B_W2 + 2 * PW2, B_WFT = B_W2T + 2 * PW2;
constexpr int B_W4 = B_WFT + 2 * 32 * SHB, B_B2 = B_W4 + 3 * HID * 4;
constexpr int BWD_SHARED = B_B2 + HID * 4;
constexpr int BWD_PRIVATE = 2 * PX + 2 * PH + RT * 16;
constexpr int BWD_WAVES = 4;
constexpr int BWD_LDS = BWD_SHARED + BWD_WAVES * BWD_PRIVATE;
static_assert(FWD_SHARED % 16 == 0 && BWD_SHARED % 16 == 0 && BWD_PRIVATE % 16 == 0, "LDS alignment");
static_assert(BWD_LDS <= 160 * 1024 && FWD_LDS <= 160 * 1024, "LDS budget");

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

// v = p[0] + p[1] (+ p[2]) + O(2^-9 NP |v|)
template <int NP>
__device__ __forceinline__ void split(float v, __bf16 (&p)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        p[i] = (__bf16)v;
        if (i + 1 < NP) v -= (float)p[i];
    }
}
// plane i of an LDS image lies `plane` bytes behind plane i - 1
template <int NP>
__device__ __forceinline__ void st_split(char* s, int off, int plane, float v) {
    __bf16 p[NP];
    split<NP>(v, p);
#pragma unroll
    for (int i = 0; i < NP; ++i) *reinterpret_cast<__bf16*>(s + off + i * plane) = p[i];
}
template <int NP>
struct Op {
    bf16x8 p[NP];
};
template <int NP>
__device__ __forceinline__ Op<NP> ldop(const char* s, int off, int plane) {
    Op<NP> o;
#pragma unroll
    for (int i = 0; i < NP; ++i) o.p[i] = *reinterpret_cast<const bf16x8*>(s + off + i * plane);
    return o;
}
__device__ __forceinline__ floatx16 mfma(bf16x8 a, bf16x8 b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// The same instruction with the accumulator pinned to the accumulation registers (AGPRs): the backward keeps 128 of
// them alive across its whole loop, and everything the VALU touches has to fit into the 256 architectural VGPRs next to
// them (built with -amdgpu-mfma-vgpr-form the compiler's own MFMAs write VGPRs; without the flag it would route EVERY
// MFMA result of a 512-register kernel through AGPRs and copy it out).  The accumulators are read only after the loop.
__device__ __forceinline__ void mfma_acc(floatx16& c, bf16x8 a, bf16x8 b) {
    // (R5: with the operands coming out of LDS reads instead of VALU conversions the padding measured as unnecessary -- 80.1 us with,
    //  81.1 / 80.7 without or as a movable statement at 242 k rays -- and stays as the conservative form)
    asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// c += a b with the terms down to 2^-9 NP, smallest first: NP = 2: lh hl hh; NP = 3: lh hl mm mh hm hh  (a index, b index)
template <int NP>
struct Terms;
template <>
struct Terms<2> {
    static constexpr int N = 3;
    static constexpr int A[3] = {1, 0, 0};
    static constexpr int B[3] = {0, 1, 0};
};
template <>
struct Terms<3> {
    static constexpr int N = 6;
    static constexpr int A[6] = {2, 0, 1, 1, 0, 0};
    static constexpr int B[6] = {0, 2, 1, 0, 1, 0};
};
template <int NP>
__device__ __forceinline__ floatx16 mfma_n(const Op<NP>& a, const Op<NP>& b, floatx16 c) {
#pragma unroll
    for (int i = 0; i < Terms<NP>::N; ++i) c = mfma(a.p[Terms<NP>::A[i]], b.p[Terms<NP>::B[i]], c);
    return c;
}
// acc[ub] += A(ub, kk) . B(ub, kk) over NK k blocks for NB output blocks, the operands of block kk + 1 requested before the
// matrix instructions of block kk issue (one wave per SIMD: nothing else hides the LDS latency), the chains of the output
// blocks interleaved.  SA / SB: that operand is the same for every output block (loaded once per k block).
template <int NP, int NK, int NB, bool SA, bool SB, class FA, class FB>
__device__ __forceinline__ void product(floatx16 (&acc)[NB], FA load_a, FB load_b) {
    constexpr int NA_ = SA ? 1 : NB, NB_ = SB ? 1 : NB;
    Op<NP> a[2][NA_], b[2][NB_];
#pragma unroll
    for (int ub = 0; ub < NA_; ++ub) a[0][ub] = load_a(ub, 0);
#pragma unroll
    for (int ub = 0; ub < NB_; ++ub) b[0][ub] = load_b(ub, 0);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        if (kk + 1 < NK) {
#pragma unroll
            for (int ub = 0; ub < NA_; ++ub) a[(kk + 1) & 1][ub] = load_a(ub, kk + 1);
#pragma unroll
            for (int ub = 0; ub < NB_; ++ub) b[(kk + 1) & 1][ub] = load_b(ub, kk + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < Terms<NP>::N; ++i)
#pragma unroll
            for (int ub = 0; ub < NB; ++ub)
                acc[ub] = mfma(a[kk & 1][SA ? 0 : ub].p[Terms<NP>::A[i]], b[kk & 1][SB ? 0 : ub].p[Terms<NP>::B[i]], acc[ub]);
    }
}
// both lane halves' values of x: lanes l and l + 32 exchanged on the VALU (v_permlane32_swap_b32; a ds_bpermute shuffle is an
// LDS round trip, seven of them cost the forward ~700 cycles per tile)
typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uintx2 both_halves(unsigned int x) { return __builtin_amdgcn_permlane32_swap(x, x, false, false); }
__device__ __forceinline__ float half_sum(float x) {
    const uintx2 r = both_halves(__float_as_uint(x));
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ unsigned int half_or(unsigned int x) {
    const uintx2 r = both_halves(x);
    return r[0] | r[1];
}
// row of accumulator register r (C/D layout of the 32x32 MFMAs): (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// input column (LDS image) -> column of W0 / gW0, -1 = the bias column, -2 = padding
__device__ __forceinline__ int col_orig(int c) {
    if (c < COL_FEAT) return 24 + c;
    if (c < COL_BIAS) return c - COL_FEAT;
    if (c == COL_BIAS) return -1;
    if (c >= COL_DIFF && c < COL_DIFF + 21) return c - 3;
    return -2;
}

// 8 consecutive columns of one row of an LDS image as NP planes (one 16-byte store per plane)
template <int NP>
__device__ __forceinline__ void st_group(char* s, int off, int plane, const float (&v)[8]) {
    bf16x8 a[NP];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __bf16 p[NP];
        split<NP>(v[i], p);
#pragma unroll
        for (int k = 0; k < NP; ++k) a[k][i] = p[k];
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x8*>(s + off + k * plane) = a[k];
}
// Weight images (once per workgroup).  Every thread first REQUESTS all the values it converts (groups of 8 columns), then
// splits and stores them: one memory round trip instead of one per element.
template <int NP, bool BWD, int NT>
__device__ __forceinline__ void stage_weights(char* s, int off_w0, int off_w2, int off_w2t, int off_wft, const MlpW& w, int t) {
    constexpr int G0 = (HID * 10 + NT - 1) / NT, G2 = (HID * 8 + NT - 1) / NT, GF = (32 * 8 + NT - 1) / NT;
    float v0[G0][8], v2[G2][8], v2t[BWD ? G2 : 1][8], vf[BWD ? GF : 1][8];
#pragma unroll
    for (int k = 0; k < G0; ++k) {
        const int g = t + k * NT, u = g / 10, c0 = 8 * (g % 10);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int co = col_orig(c0 + i);
            v0[k][i] = g < HID * 10 ? (co >= 0 ? w.W0[u * IN + co] : (co == -1 ? w.b0[u] : 0.f)) : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < G2; ++k) {
        const int g = t + k * NT, r = g / 8, c0 = 8 * (g % 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v2[k][i] = g < HID * 8 ? w.W2[r * HID + c0 + i] : 0.f;
            if (BWD) v2t[k][i] = g < HID * 8 ? w.W2[(c0 + i) * HID + r] : 0.f;      // row r of W2^T
        }
    }
    if (BWD) {
#pragma unroll
        for (int k = 0; k < GF; ++k) {
            const int g = t + k * NT, f = g / 8, c0 = 8 * (g % 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) vf[k][i] = (g < 32 * 8 && f < 24) ? w.W0[(c0 + i) * IN + f] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < G0; ++k) {
        const int g = t + k * NT;
        if (g < HID * 10) st_group<NP>(s, off_w0 + (g / 10) * SXB + 16 * (g % 10), PW0, v0[k]);
    }
#pragma unroll
    for (int k = 0; k < G2; ++k) {
        const int g = t + k * NT;
        if (g < HID * 8) {
            st_group<NP>(s, off_w2 + (g / 8) * SHB + 16 * (g % 8), PW2, v2[k]);
            if (BWD) st_group<NP>(s, off_w2t + (g / 8) * SHB + 16 * (g % 8), PW2, v2t[k]);
        }
    }
    if (BWD) {
#pragma unroll
        for (int k = 0; k < GF; ++k) {
            const int g = t + k * NT;
            if (g < 32 * 8) st_group<NP>(s, off_wft + (g / 8) * SHB + 16 * (g % 8), 32 * SHB, vf[k]);
        }
    }
}

// The complete shared region of a kernel (weight images + the fp32 rows of the last layer) from the six parameter tensors ...
template <int NT>
__device__ __forceinline__ void stage_fwd(char* smem, const MlpW& w, int t) {
    stage_weights<3, false, NT>(smem, F_W0, F_W2, 0, 0, w, t);
    float* W4w = reinterpret_cast<float*>(smem + F_W4);
    for (int i = t; i < 3 * HID; i += NT) W4w[i] = w.W4[i];
    if (t < HID) reinterpret_cast<float*>(smem + F_B2)[t] = w.b2[t];
    if (t < 4) reinterpret_cast<float*>(smem + F_B4)[t] = w.b4[t];
}
template <int NT>
__device__ __forceinline__ void stage_bwd(char* smem, const MlpW& w, int t) {
    stage_weights<2, true, NT>(smem, B_W0, B_W2, B_W2T, B_WFT, w, t);
    float* W4w = reinterpret_cast<float*>(smem + B_W4);
    for (int i = t; i < 3 * HID; i += NT) W4w[i] = w.W4[i];
    if (t < HID) reinterpret_cast<float*>(smem + B_B2)[t] = w.b2[t];
}
// ... or as a byte copy of what k_brdf_mlp_pack left in memory (nmf_brdf_mlp_pack, once per optimizer step): every 16-byte load
// of a thread in flight at once and ~70 instructions instead of the ~2000 of the conversion (a workgroup runs its prologue once,
// from a cold instruction cache): 6 us less per launch, forward and backward.
template <int BYTES, int NT>
__device__ __forceinline__ void load_image(char* smem, const uint4* __restrict__ image, int t) {
    static_assert(BYTES % 16 == 0, "images are copied in 16-byte pieces");
    constexpr int N = BYTES / 16, G = (N + NT - 1) / NT;
    uint4 v[G];
#pragma unroll
    for (int k = 0; k < G; ++k) v[k] = image[min(t + k * NT, N - 1)];
#pragma unroll
    for (int k = 0; k < G; ++k)
        if (t + k * NT < N) reinterpret_cast<uint4*>(smem)[t + k * NT] = v[k];
}
constexpr int IMG_FWD = (FWD_SHARED + 255) & ~255;      // offset of the backward's image inside a packed image
constexpr int IMG_BYTES = IMG_FWD + ((BWD_SHARED + 255) & ~255);

// The global inputs of one lane: ray (lane & 31) of a tile, lane half 0: half vector, half 1: diff vector, both: the
// feature row (half 1 does not use it; loading it everywhere keeps the code free of branches, so that the compiler can
// schedule it between the matrix instructions of the tile before).  Loaded ahead in two stages: the row index first, what
// it addresses later.  Rays past the end read the last ray (their adjoint is zero, their outputs are not stored).
struct RayIn {
    int64_t r, rc, b;
    bool valid;
    float d0, d1, d2, rough;
    float4 f[6];
};
__device__ __forceinline__ void ray_in_stage1(RayIn& in, int64_t tile, int64_t R, int ray, int h,
                                              const float* __restrict__ half_v, const float* __restrict__ diff_v,
                                              const int32_t* __restrict__ src_idx) {
    in.r = tile * RT + ray;
    in.valid = in.r < R;
    in.rc = in.valid ? in.r : R - 1;
    in.b = src_idx ? (int64_t)src_idx[in.rc] : in.rc;
    const float* d = (h == 0 ? half_v : diff_v) + in.rc * 3;
    in.d0 = d[0]; in.d1 = d[1]; in.d2 = d[2];
}
__device__ __forceinline__ void ray_in_stage2(RayIn& in, const float* __restrict__ feat_src,
                                              const float* __restrict__ rough_src) {
    in.rough = rough_src[in.b];
    const float4* f = reinterpret_cast<const float4*>(feat_src + in.b * NMF_APP_DIM);
#pragma unroll
    for (int i = 0; i < 6; ++i) in.f[i] = f[i];
}

// lane (ray = lane & 31, half h): h = 0 writes input columns 0-47, h = 1 columns 48-79 of its ray
template <int NP>
__device__ __forceinline__ void build_x(char* x, const RayIn& in, int ray, int h) {
    float v[48];
    {
        const float kappa = 1.f / (in.rough + 1e-3f);
        if constexpr (NMF_MLP_KNOCK & 2) {
#pragma unroll
            for (int i = 0; i < 18; ++i) v[i] = kappa;
        } else
        ish18(in.d0, in.d1, in.d2, kappa, v);
        v[18] = in.d0; v[19] = in.d1; v[20] = in.d2;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v[COL_FEAT + 4 * i] = in.f[i].x; v[COL_FEAT + 4 * i + 1] = in.f[i].y;
        v[COL_FEAT + 4 * i + 2] = in.f[i].z; v[COL_FEAT + 4 * i + 3] = in.f[i].w;
    }
    v[COL_BIAS] = 1.f; v[46] = 0.f; v[47] = 0.f;
#pragma unroll
    for (int i = COL_FEAT; i < 32; ++i) v[i] = h ? 0.f : v[i];     // half 1: zero padding behind its 21 columns
    const int base = ray * SXB + (h ? 2 * COL_DIFF : 0);
    auto put = [&](int g) {
        bf16x8 a[NP];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __bf16 p[NP];
            split<NP>(v[8 * g + i], p);
#pragma unroll
            for (int k = 0; k < NP; ++k) a[k][i] = p[k];
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x8*>(x + k * PX + base + 16 * g) = a[k];
    };
#pragma unroll
    for (int g = 0; g < 4; ++g) put(g);
    if (h == 0) {
        put(4);
        put(5);
    }
}

// acc (swapped layout: register r = unit acc_row(r, h) of block ub, lane & 31 = ray) -> row-major planes
template <int NP>
__device__ __forceinline__ void write_rows(char* pl, int ray, int ub, int h, const floatx16& v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bf16x4 a[NP];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __bf16 p[NP];
            split<NP>(v[4 * q + i], p);
#pragma unroll
            for (int k = 0; k < NP; ++k) a[k][i] = p[k];
        }
        const int off = ray * SHB + 2 * (32 * ub + 8 * q + 4 * h);
#pragma unroll
        for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x4*>(pl + k * PH + off) = a[k];
    }
}

__device__ __forceinline__ bf16x8 pack8(const floatx16& t, int o) {
    bf16x8 p;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (__bf16)t[o + i];
    return p;
}

// ------------------------------------------------------------------------------------------------------------------
// Forward: three bf16 terms per operand, six products per K block.  act_mask (optional) [R][4]: per ray the ReLU masks
// of the two hidden layers as 64-bit sets, {layer 1 units 0-31, 32-63, layer 2 units 0-31, 32-63}.
// bit position of accumulator register q inside its 32-unit block, lane half 0 (half 1: 4 higher)
__device__ __forceinline__ constexpr int acc_bit(int q) { return (q & 3) + 8 * (q >> 2); }

__global__ void __launch_bounds__(64 * FWD_WAVES)
k_brdf_mlp_fwd(MlpW w, const float* __restrict__ half_v, const float* __restrict__ diff_v,
               const float* __restrict__ feat_src, const float* __restrict__ rough_src,
               const int32_t* __restrict__ src_idx, int64_t R, float out_bias, float* __restrict__ out,
               uint4* __restrict__ act_mask, const uint4* __restrict__ image) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, ray = lane & 31;
    const int64_t n_tiles = (R + RT - 1) / RT, stride = (int64_t)gridDim.x * FWD_WAVES;
    int64_t tile = (int64_t)blockIdx.x * FWD_WAVES + wave;
    RayIn cur;
    ray_in_stage1(cur, tile, R, ray, h, half_v, diff_v, src_idx);
    if (image) load_image<FWD_SHARED, 64 * FWD_WAVES>(smem, image, t);
    else stage_fwd<64 * FWD_WAVES>(smem, w, t);
    ray_in_stage2(cur, feat_src, rough_src);
    char* x = smem + FWD_SHARED + wave * FWD_PRIVATE;      // input planes, then the planes of H1
    const float* W4s = reinterpret_cast<const float*>(smem + F_W4);
    const float* b2s = reinterpret_cast<const float*>(smem + F_B2);
    const float* b4s = reinterpret_cast<const float*>(smem + F_B4);
    const int ox = ray * SXB + 16 * h, oh = ray * SHB + 16 * h;
    __syncthreads();
    for (; tile < n_tiles; tile += stride) {
        RayIn nxt;
        ray_in_stage1(nxt, tile + stride, R, ray, h, half_v, diff_v, src_idx);
        build_x<3>(x, cur, ray, h);
        __builtin_amdgcn_wave_barrier();
        floatx16 a1[2] = {{0}, {0}};
        product<3, 5, 2, false, true>(
            a1, [&](int ub, int kk) { return ldop<3>(smem, F_W0 + 32 * ub * SXB + ox + 32 * kk, PW0); },
            [&](int, int kk) { return ldop<3>(x, ox + 32 * kk, PX); });
        ray_in_stage2(nxt, feat_src, rough_src);
        uint32_t m1[2] = {0u, 0u}, m2[2] = {0u, 0u};
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                m1[ub] |= a1[ub][q] > 0.f ? (1u << acc_bit(q)) : 0u;
                a1[ub][q] = fmaxf(a1[ub][q], 0.f);
            }
            write_rows<3>(x, ray, ub, h, a1[ub]);
        }
        __builtin_amdgcn_wave_barrier();
        floatx16 a2[2];
#pragma unroll
        for (int ub = 0; ub < 2; ++ub)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const floatx4v b = *reinterpret_cast<const floatx4v*>(b2s + 32 * ub + 8 * q + 4 * h);
                a2[ub][4 * q] = b[0]; a2[ub][4 * q + 1] = b[1]; a2[ub][4 * q + 2] = b[2]; a2[ub][4 * q + 3] = b[3];
            }
        product<3, 4, 2, false, true>(
            a2, [&](int ub, int kk) { return ldop<3>(smem, F_W2 + 32 * ub * SHB + oh + 32 * kk, PW2); },
            [&](int, int kk) { return ldop<3>(x, oh + 32 * kk, PH); });
        // output layer (fp32 on the VALU): o_j = sum_u relu(H2)[u] W4[j][u], the units of both lane halves
        float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
            floatx4v w4[3][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 3; ++j) w4[j][q] = *reinterpret_cast<const floatx4v*>(W4s + j * HID + 32 * ub + 8 * q + 4 * h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) m2[ub] |= a2[ub][4 * q + i] > 0.f ? (1u << acc_bit(4 * q + i)) : 0u;
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[j] += fmaxf(a2[ub][4 * q + i], 0.f) * w4[j][q][i];
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            o[j] = half_sum(o[j]);
            const float s = 1.f / (1.f + expf(-(o[j] + b4s[j] + out_bias)));      // modules/brdf.py:131
            if (cur.valid && h == 0) out[cur.r * 3 + j] = s;
        }
        if (act_mask) {
            const uint4 mk = make_uint4(half_or(m1[0] << (4 * h)), half_or(m1[1] << (4 * h)), half_or(m2[0] << (4 * h)),
                                        half_or(m2[1] << (4 * h)));
            if (cur.valid && h == 0) act_mask[cur.r] = mk;
        }
        __builtin_amdgcn_wave_barrier();
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// A 32-column block of a row-major activation plane [ray][column] with the COLUMN on the lane and the rays along the registers:
// the operand layout of the weight-gradient products (K = rays).  [kp]: rays 16 kp .. 16 kp + 15, lane half h holding 8 h .. 8 h + 7
// of them in order; .p[i]: bf16 plane i.  Two transposing LDS reads per operand register pair (see the header).
struct UOp {
    Op<2> k[2];
};
typedef short short4v __attribute__((ext_vector_type(4)));
template <int STRIDE>
__device__ __forceinline__ bf16x8 ld_tr8(const char* at) {
    typedef __attribute__((address_space(3))) short4v* lds_p;
    union { short4v s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(at));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(at + 4 * STRIDE));
    return u.v;
}
// lane part of a transposing read's address: lanes 4 k .. 4 k + 3 of a 16-lane group address row k (4 columns each), the groups of a
// lane half take columns 0-15 / 16-31, the upper lane half the rays 8 .. 15 of the 16
template <int STRIDE>
__device__ __forceinline__ int tr_lane_offset(int lane) {
    return (8 * (lane >> 5) + ((lane & 15) >> 2)) * STRIDE + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
}
// columns [col0, col0 + 32) of the two planes at `pl` (plane stride `plane`), lane offset from tr_lane_offset<STRIDE>
template <int STRIDE>
__device__ __forceinline__ UOp ld_uop(const char* pl, int plane, int col0, int lane_off) {
    UOp u;
#pragma unroll
    for (int kp = 0; kp < 2; ++kp)
#pragma unroll
        for (int i = 0; i < 2; ++i) u.k[kp].p[i] = ld_tr8<STRIDE>(pl + i * plane + kp * 16 * STRIDE + 2 * col0 + lane_off);
    return u;
}
// sum of the 8 values of an operand register set (a bias gradient: column sums over the rays)
__device__ __forceinline__ float sum8(bf16x8 v) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)v[i];
    return s;
}

constexpr int N_PERSIST = 64 + 64 + 8 + 6 + 2 + 3;   // per-lane partial sums that are reduced over the waves at the end

// One ray set of a backward launch.  A launch takes up to TWO (R4: the BRDF evaluations of a level and of the level below it share
// the weights; as two launches each paid ~25 us of fixed cost -- prologue, a first tile from a cold instruction cache, epilogue,
// the reduction of the partials -- and the smaller one stood on the main stream's chain): tiles [0, tiles0) belong to set 0, the
// rest to set 1, a tile never straddles.
struct MlpBwdSeg {
    const float *half_v, *diff_v, *feat_src, *rough_src;
    const int32_t* src_idx;
    int64_t R;
    const float* fwd_out;
    const uint4* act_mask;
    const float* d_out;
    float* d_feat;
};
struct MlpBwdSegs {
    MlpBwdSeg s[2];      // (a launch with one set passes it twice: tiles past the end are prefetched, never worked on)
    int64_t tiles0, n_tiles;
};

// Backward.  fwd_out / act_mask are the forward's outputs for the same inputs.
__global__ void __launch_bounds__(64 * BWD_WAVES)
k_brdf_mlp_bwd(MlpW w, MlpBwdSegs G, float* __restrict__ partials, const uint4* __restrict__ image) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = BWD_WAVES, NT = 64 * BWD_WAVES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, ray = lane & 31;
    const int64_t n_tiles = G.n_tiles, stride = (int64_t)gridDim.x * NW;
    int64_t tile = (int64_t)blockIdx.x * NW + wave;
    // the ray set of a tile (wave-uniform selects of kernel arguments) and the tile's index inside it
    auto pick = [&](int64_t tl, int64_t& local) {
        const bool second = tl >= G.tiles0;
        local = second ? tl - G.tiles0 : tl;
        MlpBwdSeg S;
        S.half_v = second ? G.s[1].half_v : G.s[0].half_v;
        S.diff_v = second ? G.s[1].diff_v : G.s[0].diff_v;
        S.feat_src = second ? G.s[1].feat_src : G.s[0].feat_src;
        S.rough_src = second ? G.s[1].rough_src : G.s[0].rough_src;
        S.src_idx = second ? G.s[1].src_idx : G.s[0].src_idx;
        S.R = second ? G.s[1].R : G.s[0].R;
        S.fwd_out = second ? G.s[1].fwd_out : G.s[0].fwd_out;
        S.act_mask = second ? G.s[1].act_mask : G.s[0].act_mask;
        S.d_out = second ? G.s[1].d_out : G.s[0].d_out;
        S.d_feat = second ? G.s[1].d_feat : G.s[0].d_feat;
        return S;
    };
    // the forward's outputs and the incoming adjoint of a lane's ray, requested one tile ahead like RayIn (raw values: nothing
    // is computed from them before the tile that uses them, so the loads stay in flight across the tile before)
    struct Adj {
        uint4 mk;
        float s[3], go[3];
        bool valid;
        __device__ __forceinline__ float g(int j) const { return valid ? go[j] * s[j] * (1.f - s[j]) : 0.f; }
    };
    auto load_adj = [&](Adj& a, const RayIn& in, const MlpBwdSeg& S) {
        a.mk = S.act_mask[in.rc];
        a.valid = in.valid;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            a.s[j] = S.fwd_out[in.rc * 3 + j];
            a.go[j] = S.d_out[in.rc * 3 + j];
        }
    };
    RayIn cur;
    Adj adj;
    int64_t tl_local;
    MlpBwdSeg Sc = pick(tile, tl_local);
    ray_in_stage1(cur, tl_local, Sc.R, ray, h, Sc.half_v, Sc.diff_v, Sc.src_idx);
    load_adj(adj, cur, Sc);
    if (image) load_image<BWD_SHARED, NT>(smem, image + IMG_FWD / 16, t);
    else stage_bwd<NT>(smem, w, t);
    ray_in_stage2(cur, Sc.feat_src, Sc.rough_src);
    char* priv = smem + BWD_SHARED + wave * BWD_PRIVATE;
    char* x = priv;                                                 // input planes
    char* pl = priv + 2 * PX;                                       // 64-wide planes: H1, then dH2, then dH1
    float* gs = reinterpret_cast<float*>(priv + 2 * PX + 2 * PH);   // adjoint of the pre-sigmoid outputs [32][4]
    const float* W4s = reinterpret_cast<const float*>(smem + B_W4);
    const float* b2s = reinterpret_cast<const float*>(smem + B_B2);
    // operand offsets of this lane: row (lane & 31), 16-byte half h of a 32-byte k block; transposing reads: tr_lane_offset
    const int ox = ray * SXB + 16 * h, oh = ray * SHB + 16 * h;
    const int tx = tr_lane_offset<SXB>(lane), th_ = tr_lane_offset<SHB>(lane);
    floatx16 accW2[2][2], accW0[2][2];
    float accTail[2][4], accW4[3][2], accb2[2], accb4[3];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) accW2[a][b][q] = 0.f, accW0[a][b][q] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) accTail[a][q] = 0.f;
        accb2[a] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) accW4[j][a] = 0.f;
    }
    accb4[0] = accb4[1] = accb4[2] = 0.f;
    __syncthreads();
    build_x<2>(x, cur, ray, h);
    __builtin_amdgcn_wave_barrier();
    if constexpr (NMF_MLP_KNOCK & 128) return;
    if constexpr (!(NMF_MLP_KNOCK & 64))
    for (; tile < n_tiles; tile += stride) {
        // the input planes of this tile are in LDS; the row indices / adjoints of the next one are requested now
        RayIn nxt;
        Adj nadj;
        int64_t nx_local;
        const MlpBwdSeg Sn = pick(tile + stride, nx_local);
        ray_in_stage1(nxt, nx_local, Sn.R, ray, h, Sn.half_v, Sn.diff_v, Sn.src_idx);
        load_adj(nadj, nxt, Sn);
        const uint32_t m1[2] = {adj.mk.x >> (4 * h), adj.mk.y >> (4 * h)}, m2[2] = {adj.mk.z >> (4 * h), adj.mk.w >> (4 * h)};
        const float g[3] = {adj.g(0), adj.g(1), adj.g(2)};
        if (h == 0) {
            floatx4v gv = {g[0], g[1], g[2], 0.f};
            *reinterpret_cast<floatx4v*>(gs + 4 * ray) = gv;
            accb4[0] += g[0]; accb4[1] += g[1]; accb4[2] += g[2];
        }
        // ---- layer 1 (values; the forward's mask decides which units are alive): H1^T = [m1] (W0 X^T)
        {
            floatx16 a1[2] = {{0}, {0}};
            product<2, 5, 2, false, true>(
                a1, [&](int ub, int kk) { return ldop<2>(smem, B_W0 + 32 * ub * SXB + ox + 32 * kk, PW0); },
                [&](int, int kk) { return ldop<2>(x, ox + 32 * kk, PX); });
#pragma unroll
            for (int ub = 0; ub < 2; ++ub) {
#pragma unroll
                for (int q = 0; q < 16; ++q) a1[ub][q] = (m1[ub] >> acc_bit(q)) & 1u ? a1[ub][q] : 0.f;
                write_rows<2>(pl, ray, ub, h, a1[ub]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        UOp h1u[2];
        {
            Op<2> hb[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) hb[kk] = ldop<2>(pl, oh + 32 * kk, PH);
            // ---- layer 2 with the unit on the lane (rows = rays): relu(H2) as values for dW4
            floatx16 a2u[2] = {{0}, {0}};
            product<2, 4, 2, true, false>(
                a2u, [&](int, int kk) { return hb[kk]; },
                [&](int ub, int kk) { return ldop<2>(smem, B_W2 + 32 * ub * SHB + oh + 32 * kk, PW2); });
            h1u[0] = ld_uop<SHB>(pl, PH, 0, th_);
            h1u[1] = ld_uop<SHB>(pl, PH, 32, th_);
            // dW4[j][u] += sum_ray g[ray][j] relu(H2)[ray][u]   (the 16 adjoint rows requested in two batches, then used)
            const float bu0 = b2s[ray], bu1 = b2s[32 + ray];
            if constexpr (!(NMF_MLP_KNOCK & 4))
#pragma unroll
            for (int qb = 0; qb < 16; qb += 8) {
                floatx4v gv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) gv[q] = *reinterpret_cast<const floatx4v*>(gs + 4 * acc_row(qb + q, h));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v0 = fmaxf(a2u[0][qb + q] + bu0, 0.f), v1 = fmaxf(a2u[1][qb + q] + bu1, 0.f);
#pragma unroll
                    for (int j = 0; j < 3; ++j) accW4[j][0] += gv[q][j] * v0, accW4[j][1] += gv[q][j] * v1;
                }
            }
        }
        // ---- dH2 = [m2] (g W4)  -> row-major planes (H1's operands are in registers by now)
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
            floatx4v w4[3][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 3; ++j) w4[j][q] = *reinterpret_cast<const floatx4v*>(W4s + j * HID + 32 * ub + 8 * q + 4 * h);
            __builtin_amdgcn_sched_barrier(0);
            floatx16 d2;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = g[0] * w4[0][q][i] + g[1] * w4[1][q][i] + g[2] * w4[2][q][i];
                    d2[4 * q + i] = (m2[ub] >> acc_bit(4 * q + i)) & 1u ? d : 0.f;
                }
            write_rows<2>(pl, ray, ub, h, d2);
        }
        __builtin_amdgcn_wave_barrier();
        ray_in_stage2(nxt, Sn.feat_src, Sn.rough_src);     // the next tile's feature rows: used at the end of this iteration
        {
            Op<2> dh[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dh[kk] = ldop<2>(pl, oh + 32 * kk, PH);
            // dH1^T = [m1] (W2^T dH2^T)
            floatx16 d1[2] = {{0}, {0}};
            product<2, 4, 2, false, true>(
                d1, [&](int ub, int kk) { return ldop<2>(smem, B_W2T + 32 * ub * SHB + oh + 32 * kk, PW2); },
                [&](int, int kk) { return dh[kk]; });
            // dW2 += dH2^T H1, db2 += column sums of dH2 (register operands only)
            if constexpr (!(NMF_MLP_KNOCK & 32))
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const UOp d2u = ld_uop<SHB>(pl, PH, 32 * a, th_);
                accb2[a] += (sum8(d2u.k[0].p[0]) + sum8(d2u.k[0].p[1])) + (sum8(d2u.k[1].p[0]) + sum8(d2u.k[1].p[1]));
#pragma unroll
                for (int kp = 0; kp < 2; ++kp)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            mfma_acc(accW2[a][b], d2u.k[kp].p[Terms<2>::A[i]], h1u[b].k[kp].p[Terms<2>::B[i]]);
            }
#pragma unroll
            for (int ub = 0; ub < 2; ++ub) {
#pragma unroll
                for (int q = 0; q < 16; ++q) d1[ub][q] = (m1[ub] >> acc_bit(q)) & 1u ? d1[ub][q] : 0.f;
                write_rows<2>(pl, ray, ub, h, d1[ub]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        floatx16 dx[1] = {{0}};
        {
            Op<2> dh[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dh[kk] = ldop<2>(pl, oh + 32 * kk, PH);
            // adjoint of the gathered feature columns: dX[:, 0:24]^T = W0[:, 0:24]^T dH1^T
            product<2, 4, 1, false, true>(
                dx, [&](int, int kk) { return ldop<2>(smem, B_WFT + oh + 32 * kk, 32 * SHB); },
                [&](int, int kk) { return dh[kk]; });
            // dW0 += dH1^T X (input columns 0-63 as two 32-column blocks; 64-79 through the transposed product)
            UOp d1u[2];
            d1u[0] = ld_uop<SHB>(pl, PH, 0, th_);
            d1u[1] = ld_uop<SHB>(pl, PH, 32, th_);
            if constexpr (!(NMF_MLP_KNOCK & 16))
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const UOp xu = ld_uop<SXB>(x, PX, 32 * cb, tx);
#pragma unroll
                for (int kp = 0; kp < 2; ++kp)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int a = 0; a < 2; ++a)
                            mfma_acc(accW0[a][cb], d1u[a].k[kp].p[Terms<2>::A[i]], xu.k[kp].p[Terms<2>::B[i]]);
            }
            if constexpr (!(NMF_MLP_KNOCK & 8)) {
                // the trailing input columns 64 .. (the diff vector's last five) as ROWS of a transposed product: columns 64-95 of the
                // image on the lanes (80-95 are padding / the next row: their result rows are never read)
                const UOp xtail = ld_uop<SXB>(x, PX, 64, tx);
                const Op<2> (&xk)[2] = xtail.k;
                floatx16 tt[2] = {{0}, {0}};
#pragma unroll
                for (int kp = 0; kp < 2; ++kp)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int a = 0; a < 2; ++a)
                            tt[a] = mfma(xk[kp].p[Terms<2>::A[i]], d1u[a].k[kp].p[Terms<2>::B[i]], tt[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) accTail[a][q] += tt[a][q];
            }
        }
        // ---- the input planes of the next tile (this tile's have been read)
        __builtin_amdgcn_wave_barrier();
        build_x<2>(x, nxt, ray, h);
        // ---- adjoint of the gathered feature rows (dx: lane = ray, registers = columns 8 q + 4 h + i): the rays of a bounce row
        //      are consecutive, so the tile's 32 x 24 block goes through LDS once (the planes of dH1 are in registers by now) and
        //      ONE atomic per (row, column, half tile) leaves -- instead of [R][24] floats to memory and a segmented-sum launch
        //      behind this kernel.  Lanes 0-23 / 24-47 take column (lane % 24) of rays 0-15 / 16-31: sixteen independent LDS reads,
        //      then a wave-uniform loop over the RUNS of equal row indices (a ballot of the run starts; 1-4 per tile), each run a
        //      masked sum of the registers.  (R3 walked the 32 rays one by one on 24 lanes, an LDS round trip and a branch per ray:
        //      14 of the 95 us of a 242 k-ray launch.)
        if constexpr (!(NMF_MLP_KNOCK & 1)) {
            float* xs = reinterpret_cast<float*>(pl);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                floatx4v v = {dx[0][4 * q], dx[0][4 * q + 1], dx[0][4 * q + 2], dx[0][4 * q + 3]};
                *reinterpret_cast<floatx4v*>(xs + ray * 24 + 8 * q + 4 * h) = v;
            }
            __builtin_amdgcn_wave_barrier();
            const int part = lane >= 24 ? 1 : 0, col = lane - 24 * part;       // (lanes 48-63: no column, they only vote)
            const bool owner = lane < 48;
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = xs[(16 * part + j) * 24 + (owner ? col : 0)];
            const int brow = (int)cur.b;                                         // lanes l and l + 32 hold the same ray
            const int before = __shfl_up(brow, 1, 64);
            uint32_t starts = (uint32_t)__ballot(lane < 32 && (lane == 0 || brow != before));
            while (starts) {
                const int s = __builtin_ctz(starts);
                starts &= starts - 1;
                const int e = starts ? __builtin_ctz(starts) : 32;
                const int row = __builtin_amdgcn_readlane(brow, s);
                // rays [s, e) as bits of this lane's sixteen
                const uint32_t sel = (uint32_t)((((1ull << e) - 1ull) & ~((1ull << s) - 1ull)) >> (16 * part)) & 0xffffu;
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc += (sel >> j) & 1u ? v[j] : 0.f;
                if (owner && sel) atomicAdd(Sc.d_feat + (int64_t)row * 24 + col, acc);
            }
        }
        __builtin_amdgcn_wave_barrier();
        cur = nxt;
        adj = nadj;
        Sc = Sn;
    }
    // ---- the per-lane sums of the four waves -> one partial per workgroup in the workspace (k_brdf_mlp_reduce adds the
    //      partials of all workgroups in a fixed order: one atomic per gradient element and call instead of one per workgroup)
    asm volatile("s_nop 15\n\ts_nop 15");      // the last accumulator MFMAs have retired before their registers are read
    __syncthreads();
    {
        float* red = reinterpret_cast<float*>(smem) + wave * (N_PERSIST * 64);      // the weight images are dead by now
        int idx = 0;
        auto put = [&](float v) {
            red[idx * 64 + lane] = v;
            ++idx;
        };
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 16; ++q) put(accW2[a][b][q]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 16; ++q) put(accW0[a][b][q]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) put(accTail[a][q]);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int a = 0; a < 2; ++a) put(accW4[j][a]);
        put(accb2[0]);
        put(accb2[1]);
#pragma unroll
        for (int j = 0; j < 3; ++j) put(h == 0 ? accb4[j] : 0.f);
    }
    __syncthreads();
    {
        const float* red = reinterpret_cast<const float*>(smem);
        float* part = partials + (int64_t)blockIdx.x * (N_PERSIST * 64);
        for (int e = t; e < N_PERSIST * 64; e += NT) {
            float v = red[e];
#pragma unroll
            for (int wv = 1; wv < NW; ++wv) v += red[wv * (N_PERSIST * 64) + e];
            part[e] = v;
        }
    }
}

// gradient element e of a workgroup partial (layout: k_brdf_mlp_bwd's `put` order x 64 lanes) += v
__device__ __forceinline__ void mlp_grad_emit(int e, float v, float* __restrict__ gW0, float* __restrict__ gb0,
                                              float* __restrict__ gW2, float* __restrict__ gb2, float* __restrict__ gW4,
                                              float* __restrict__ gb4) {
    const int idx = e >> 6, ln = e & 63, hh = ln >> 5, c = ln & 31;
    if (idx < 64) {                                   // dW2[a][b]: row = u2, column = u1
        const int a = idx >> 5, b = (idx >> 4) & 1, q = idx & 15;
        atomicAdd(gW2 + (32 * a + acc_row(q, hh)) * HID + 32 * b + c, v);
    } else if (idx < 128) {                           // dW0[a][cb]: row = u1, column = input column 32 cb + c
        const int k = idx - 64, a = k >> 5, cb = (k >> 4) & 1, q = k & 15;
        const int u = 32 * a + acc_row(q, hh), co = col_orig(32 * cb + c);
        if (co >= 0) atomicAdd(gW0 + u * IN + co, v);
        else if (co == -1) atomicAdd(gb0 + u, v);
    } else if (idx < 136) {                           // input columns 64 + acc_row(q, hh), unit 32 a + c
        const int k = idx - 128, a = k >> 2, q = k & 3;
        const int co = col_orig(64 + acc_row(q, hh));
        if (co >= 0) atomicAdd(gW0 + (32 * a + c) * IN + co, v);
    } else if (idx < 142) {
        const int k = idx - 136, j = k >> 1, a = k & 1;
        atomicAdd(gW4 + j * HID + 32 * a + c, v);
    } else if (idx < 144) {
        atomicAdd(gb2 + 32 * (idx - 142) + c, v);
    } else {
        atomicAdd(gb4 + (idx - 144), v);
    }
}

// sum of the workgroup partials, one atomic per gradient element and CALL (R4).  A workgroup takes 32 consecutive elements (one
// 128-byte run of every partial) and splits the partials over its eight 32-lane groups -- a lane reads n_wg / 8 values in batches of
// eight independent loads --, the groups meet in LDS.  Rounds 3's form (a thread per element and slice of 32 partials, one atomic
// per slice) took 18-31 us for 4-9 MB: four dependent load batches per thread and 3-8 atomics per address; finer slices were
// worse (8 partials per thread: 112 -> 169 us for the whole backward at 242 k rays -- same-address float atomics retire one by one).
__global__ void __launch_bounds__(256)
k_brdf_mlp_reduce(const float* __restrict__ partials, int n_wg, float* __restrict__ gW0, float* __restrict__ gb0,
                  float* __restrict__ gW2, float* __restrict__ gb2, float* __restrict__ gW4, float* __restrict__ gb4) {
    __shared__ float s_part[8][32];
    const int el = threadIdx.x & 31, wl = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;                     // N_PERSIST * 64 is a multiple of 32
    float v = 0.f;
    for (int wb = wl; wb < n_wg; wb += 64) {
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = wb + 8 * k < n_wg ? partials[(int64_t)(wb + 8 * k) * (N_PERSIST * 64) + e] : 0.f;
        v += ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    s_part[wl][el] = v;
    __syncthreads();
    if (wl == 0) {
        float t = ((s_part[0][el] + s_part[1][el]) + (s_part[2][el] + s_part[3][el])) +
                  ((s_part[4][el] + s_part[5][el]) + (s_part[6][el] + s_part[7][el]));
        mlp_grad_emit(e, t, gW0, gb0, gW2, gb2, gW4, gb4);
    }
}

// both images of a packed weight set: workgroup 0 the forward's (three bf16 terms), workgroup 1 the backward's (two terms, W2^T,
// W0[:, :24]^T), each staged exactly as the kernels stage them and copied out byte by byte
__global__ void __launch_bounds__(256) k_brdf_mlp_pack(MlpW w, uint4* __restrict__ image) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int bytes = blockIdx.x == 0 ? FWD_SHARED : BWD_SHARED;
    for (int i = t; i < bytes / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);      // (the row pads)
    __syncthreads();
    if (blockIdx.x == 0) stage_fwd<256>(smem, w, t);
    else stage_bwd<256>(smem, w, t);
    __syncthreads();
    uint4* dst = image + (blockIdx.x == 0 ? 0 : IMG_FWD / 16);
    for (int i = t; i < bytes / 16; i += 256) dst[i] = reinterpret_cast<const uint4*>(smem)[i];
}

}  // namespace

static int check_w(const float* const* p) {
    for (int i = 0; i < 6; ++i)
        if (!p[i]) return 0;
    return 1;
}

static int set_lds_once(const void* fn, int bytes, const char* what) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e == hipSuccess ? NMF_OK : nmf_fail((int)e, what);
}

static int mlp_fwd_impl(const MlpW& w, const void* image, const float* half_vec, const float* diff_vec, const float* feat_src,
                        const float* rough_src, const int32_t* src_idx, int64_t R, float out_bias, float* out, uint32_t* act_mask,
                        int32_t max_workgroups, void* stream) {
    NMF_REQUIRE(half_vec && diff_vec && feat_src && rough_src && out, NMF_EINVAL, "nmf_brdf_mlp_fwd: null");
    static const int lds = set_lds_once((const void*)k_brdf_mlp_fwd, FWD_LDS, "nmf_brdf_mlp_fwd: hipFuncSetAttribute");
    if (lds != NMF_OK) return lds;
    const int64_t wgs = cdiv(cdiv(R, RT), FWD_WAVES);
    int64_t cap = 256;                                  // one workgroup per CU (127 KB of LDS)
    if (max_workgroups > 0 && max_workgroups < cap) cap = max_workgroups;
    const unsigned grid = (unsigned)(wgs < cap ? wgs : cap);
    NMF_LAUNCH(k_brdf_mlp_fwd, dim3(grid), dim3(64 * FWD_WAVES), FWD_LDS, (hipStream_t)stream, w, half_vec,
                       diff_vec, feat_src, rough_src, src_idx, R, out_bias, out, reinterpret_cast<uint4*>(act_mask),
                       static_cast<const uint4*>(image));
    NMF_CHECK_LAUNCH("nmf_brdf_mlp_fwd");
    return NMF_OK;
}

extern "C" int nmf_brdf_mlp_fwd(const float* W0, const float* b0, const float* W2, const float* b2, const float* W4,
                                const float* b4, const float* half_vec, const float* diff_vec, const float* feat_src,
                                const float* rough_src, const int32_t* src_idx, int64_t R, float out_bias, float* out,
                                uint32_t* act_mask, int32_t max_workgroups, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_brdf_mlp_fwd: R < 0");
    if (R == 0) return NMF_OK;
    const float* ws[6] = {W0, b0, W2, b2, W4, b4};
    NMF_REQUIRE(check_w(ws), NMF_EINVAL, "nmf_brdf_mlp_fwd: null");
    return mlp_fwd_impl(MlpW{W0, b0, W2, b2, W4, b4}, nullptr, half_vec, diff_vec, feat_src, rough_src, src_idx, R, out_bias, out,
                        act_mask, max_workgroups, stream);
}

extern "C" int64_t nmf_brdf_mlp_image_bytes(void) { return IMG_BYTES; }

extern "C" int nmf_brdf_mlp_pack(const float* W0, const float* b0, const float* W2, const float* b2, const float* W4,
                                 const float* b4, void* image, int64_t image_bytes, void* stream) {
    const float* ws[6] = {W0, b0, W2, b2, W4, b4};
    NMF_REQUIRE(check_w(ws) && image, NMF_EINVAL, "nmf_brdf_mlp_pack: null");
    NMF_REQUIRE(image_bytes >= IMG_BYTES && ((uintptr_t)image & 15) == 0, NMF_EINVAL,
                "nmf_brdf_mlp_pack: image too small (nmf_brdf_mlp_image_bytes) or not 16-byte aligned");
    constexpr int lds_bytes = FWD_SHARED > BWD_SHARED ? FWD_SHARED : BWD_SHARED;
    static const int lds = set_lds_once((const void*)k_brdf_mlp_pack, lds_bytes, "nmf_brdf_mlp_pack: hipFuncSetAttribute");
    if (lds != NMF_OK) return lds;
    NMF_LAUNCH(k_brdf_mlp_pack, dim3(2), dim3(256), lds_bytes, (hipStream_t)stream, MlpW{W0, b0, W2, b2, W4, b4},
                       static_cast<uint4*>(image));
    NMF_CHECK_LAUNCH("nmf_brdf_mlp_pack");
    return NMF_OK;
}

extern "C" int nmf_brdf_mlp_fwd_packed(const void* image, const float* half_vec, const float* diff_vec, const float* feat_src,
                                       const float* rough_src, const int32_t* src_idx, int64_t R, float out_bias, float* out,
                                       uint32_t* act_mask, int32_t max_workgroups, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_brdf_mlp_fwd_packed: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(image && ((uintptr_t)image & 15) == 0, NMF_EINVAL, "nmf_brdf_mlp_fwd_packed: image null or not 16-byte aligned");
    return mlp_fwd_impl(MlpW{}, image, half_vec, diff_vec, feat_src, rough_src, src_idx, R, out_bias, out, act_mask, max_workgroups,
                        stream);
}

static unsigned bwd_grid_tiles(int64_t tiles, int32_t max_workgroups) {
    // One workgroup of 4 waves per CU (150 KB of LDS, one wave per SIMD).  Every workgroup stages the weight images
    // and writes a 37 KB partial, so short launches use fewer of them: at least 2 tiles per wave (45 k rays: 54 us against 63
    // with 4 tiles per wave).
    const int64_t wgs = cdiv(tiles, BWD_WAVES * 2);
    int64_t cap = 256;
    if (max_workgroups > 0 && max_workgroups < cap) cap = max_workgroups;
    return (unsigned)(wgs < cap ? wgs : cap);
}
static unsigned bwd_grid(int64_t R, int32_t max_workgroups) { return bwd_grid_tiles(cdiv(R, RT), max_workgroups); }

extern "C" int64_t nmf_brdf_mlp_bwd_workspace_bytes(int64_t R, int32_t max_workgroups) {
    return R <= 0 ? 0 : (int64_t)bwd_grid(R, max_workgroups) * N_PERSIST * 64 * (int64_t)sizeof(float);
}

extern "C" int64_t nmf_brdf_mlp_bwd_segments_workspace_bytes(const int64_t* Rs, int32_t n_segs, int32_t max_workgroups) {
    int64_t tiles = 0;
    for (int i = 0; Rs && i < n_segs; ++i) tiles += Rs[i] > 0 ? cdiv(Rs[i], RT) : 0;
    return tiles <= 0 ? 0 : (int64_t)bwd_grid_tiles(tiles, max_workgroups) * N_PERSIST * 64 * (int64_t)sizeof(float);
}

// segs: one or two ray sets with R > 0
static int mlp_bwd_launch(const MlpW& w, const void* image, const nmf_mlp_bwd_segment* segs, int n, float* gW0, float* gb0,
                          float* gW2, float* gb2, float* gW4, float* gb4, int32_t max_workgroups, void* workspace,
                          int64_t workspace_bytes, void* stream) {
    NMF_REQUIRE(gW0 && gb0 && gW2 && gb2 && gW4 && gb4, NMF_EINVAL, "nmf_brdf_mlp_bwd: null");
    MlpBwdSegs G;
    int64_t tiles[2] = {0, 0};
    for (int i = 0; i < n; ++i) {
        const nmf_mlp_bwd_segment& q = segs[i];
        NMF_REQUIRE(q.half_vec && q.diff_vec && q.feat_src && q.rough_src && q.fwd_out && q.act_mask && q.d_out && q.d_feat,
                    NMF_EINVAL, "nmf_brdf_mlp_bwd: null");
        G.s[i] = MlpBwdSeg{q.half_vec, q.diff_vec, q.feat_src, q.rough_src, q.src_idx, q.R, q.fwd_out,
                           reinterpret_cast<const uint4*>(q.act_mask), q.d_out, q.d_feat};
        tiles[i] = cdiv(q.R, RT);
    }
    if (n == 1) G.s[1] = G.s[0];
    G.tiles0 = tiles[0];
    G.n_tiles = tiles[0] + tiles[1];
    const unsigned grid = bwd_grid_tiles(G.n_tiles, max_workgroups);
    NMF_REQUIRE(workspace && workspace_bytes >= (int64_t)grid * N_PERSIST * 64 * (int64_t)sizeof(float), NMF_EINVAL,
                "nmf_brdf_mlp_bwd: workspace too small (nmf_brdf_mlp_bwd_workspace_bytes)");
    static_assert(BWD_WAVES * N_PERSIST * 64 * 4 <= BWD_LDS, "the per-wave sums fit into the kernel's LDS");
    static const int lds = set_lds_once((const void*)k_brdf_mlp_bwd, BWD_LDS, "nmf_brdf_mlp_bwd: hipFuncSetAttribute");
    if (lds != NMF_OK) return lds;
    float* partials = static_cast<float*>(workspace);
    NMF_LAUNCH(k_brdf_mlp_bwd, dim3(grid), dim3(64 * BWD_WAVES), BWD_LDS, (hipStream_t)stream, w, G, partials,
                       static_cast<const uint4*>(image));
    NMF_CHECK_LAUNCH("nmf_brdf_mlp_bwd");
    static_assert((N_PERSIST * 64) % 32 == 0, "k_brdf_mlp_reduce takes 32 elements per workgroup");
    NMF_LAUNCH(k_brdf_mlp_reduce, dim3(N_PERSIST * 64 / 32), dim3(256), 0, (hipStream_t)stream, partials, (int)grid, gW0,
                       gb0, gW2, gb2, gW4, gb4);
    NMF_CHECK_LAUNCH("nmf_brdf_mlp_bwd (reduce)");
    return NMF_OK;
}

static int mlp_bwd_impl(const MlpW& w, const void* image, const float* half_vec, const float* diff_vec, const float* feat_src,
                        const float* rough_src, const int32_t* src_idx, int64_t R, const float* fwd_out, const uint32_t* act_mask,
                        const float* d_out, float* d_feat, float* gW0, float* gb0, float* gW2, float* gb2, float* gW4, float* gb4,
                        int32_t max_workgroups, void* workspace, int64_t workspace_bytes, void* stream) {
    const nmf_mlp_bwd_segment seg{half_vec, diff_vec, feat_src, rough_src, src_idx, R, fwd_out, act_mask, d_out, d_feat};
    return mlp_bwd_launch(w, image, &seg, 1, gW0, gb0, gW2, gb2, gW4, gb4, max_workgroups, workspace, workspace_bytes, stream);
}

extern "C" int nmf_brdf_mlp_bwd_segments(const void* image, const float* W0, const float* b0, const float* W2, const float* b2,
                                         const float* W4, const float* b4, const nmf_mlp_bwd_segment* segs, int32_t n_segs,
                                         float* gW0, float* gb0, float* gW2, float* gb2, float* gW4, float* gb4,
                                         int32_t max_workgroups, void* workspace, int64_t workspace_bytes, void* stream) {
    NMF_REQUIRE(segs && n_segs >= 1 && n_segs <= 2, NMF_EINVAL, "nmf_brdf_mlp_bwd_segments: one or two ray sets");
    const float* ws[6] = {W0, b0, W2, b2, W4, b4};
    NMF_REQUIRE((image && ((uintptr_t)image & 15) == 0) || check_w(ws), NMF_EINVAL,
                "nmf_brdf_mlp_bwd_segments: a packed image (16-byte aligned) or the six weight tensors");
    nmf_mlp_bwd_segment live[2];
    int n = 0;
    for (int i = 0; i < n_segs; ++i) {
        NMF_REQUIRE(segs[i].R >= 0, NMF_EINVAL, "nmf_brdf_mlp_bwd_segments: R < 0");
        if (segs[i].R > 0) live[n++] = segs[i];
    }
    if (n == 0) return NMF_OK;
    return mlp_bwd_launch(image ? MlpW{} : MlpW{W0, b0, W2, b2, W4, b4}, image, live, n, gW0, gb0, gW2, gb2, gW4, gb4, max_workgroups,
                          workspace, workspace_bytes, stream);
}

extern "C" int nmf_brdf_mlp_bwd(const float* W0, const float* b0, const float* W2, const float* b2, const float* W4,
                                const float* b4, const float* half_vec, const float* diff_vec, const float* feat_src,
                                const float* rough_src, const int32_t* src_idx, int64_t R, const float* fwd_out,
                                const uint32_t* act_mask, const float* d_out, float* d_feat, float* gW0, float* gb0,
                                float* gW2, float* gb2, float* gW4, float* gb4, int32_t max_workgroups, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_brdf_mlp_bwd: R < 0");
    if (R == 0) return NMF_OK;
    const float* ws[6] = {W0, b0, W2, b2, W4, b4};
    NMF_REQUIRE(check_w(ws), NMF_EINVAL, "nmf_brdf_mlp_bwd: null");
    return mlp_bwd_impl(MlpW{W0, b0, W2, b2, W4, b4}, nullptr, half_vec, diff_vec, feat_src, rough_src, src_idx, R, fwd_out, act_mask,
                        d_out, d_feat, gW0, gb0, gW2, gb2, gW4, gb4, max_workgroups, workspace, workspace_bytes, stream);
}

extern "C" int nmf_brdf_mlp_bwd_packed(const void* image, const float* half_vec, const float* diff_vec, const float* feat_src,
                                       const float* rough_src, const int32_t* src_idx, int64_t R, const float* fwd_out,
                                       const uint32_t* act_mask, const float* d_out, float* d_feat, float* gW0, float* gb0,
                                       float* gW2, float* gb2, float* gW4, float* gb4, int32_t max_workgroups, void* workspace,
                                       int64_t workspace_bytes, void* stream) {
    NMF_REQUIRE(R >= 0, NMF_EINVAL, "nmf_brdf_mlp_bwd_packed: R < 0");
    if (R == 0) return NMF_OK;
    NMF_REQUIRE(image && ((uintptr_t)image & 15) == 0, NMF_EINVAL, "nmf_brdf_mlp_bwd_packed: image null or not 16-byte aligned");
    return mlp_bwd_impl(MlpW{}, image, half_vec, diff_vec, feat_src, rough_src, src_idx, R, fwd_out, act_mask, d_out, d_feat, gW0,
                        gb0, gW2, gb2, gW4, gb4, max_workgroups, workspace, workspace_bytes, stream);
}
