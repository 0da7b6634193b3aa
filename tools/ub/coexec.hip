#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
// MODE 0: chain of dependent 32x32x2 f32 MFMAs.  MODE 1: + NV independent v_fma between them.  MODE 2: only the v_fma.
// MODE 3: MFMA chain + NV ds_read_b32 between.   MODE 4: two waves per SIMD: even waves MFMA, odd waves VALU.
template <int MODE, int NV>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1.f;
    __syncthreads();
    floatx16 acc = {0};
    float a = lane * 0.001f, b = 1.0001f;
    float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    float s = 0.f;
    const bool do_mfma = MODE == 0 || MODE == 1 || MODE == 3 || (MODE == 4 && (wave & 4) == 0);
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 4 && (wave & 4) != 0);
    const bool do_lds = MODE == 3 || (MODE == 5 && (wave & 4) != 0);
    const bool do_mfma5 = MODE == 5 && (wave & 4) == 0;
    for (int it = 0; it < iters; ++it) {
        if (do_mfma || do_mfma5) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        if (do_valu) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], 1.0001f, 0.5f);
        }
        if (do_lds) {
#pragma unroll
            for (int i = 0; i < NV; ++i) s += lds[(lane + 64 * i + it) & 4095];
        }
    }
    float r = s;
    for (int i = 0; i < 16; ++i) r += acc[i];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE, int NV>
void run(const char* name, int threads) {
    float* out; hipMalloc(&out, 1024 * 512 * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    printf("%-46s %8.1f us  -> %.1f ns / iteration\n", name, ms * 1e3, ms * 1e6 / iters);
    hipFree(out);
}
int main() {
    run<0, 0>("mfma chain only (1 wave/SIMD)", 256);
    run<2, 8>("8 fma only", 256);
    run<1, 8>("mfma + 8 independent fma, same wave", 256);
    run<2, 14>("14 fma only", 256);
    run<1, 14>("mfma + 14 independent fma, same wave", 256);
    run<3, 4>("mfma + 4 ds_read, same wave", 256);
    run<3, 8>("mfma + 8 ds_read, same wave", 256);
    run<4, 8>("2 waves/SIMD: one mfma, other 8 fma", 512);
    run<4, 14>("2 waves/SIMD: one mfma, other 14 fma", 512);
    run<0, 0>("mfma chain, 2 waves/SIMD both mfma", 512);
    run<2, 8>("8 fma only, 2 waves/SIMD", 512);
    run<5, 4>("2 waves/SIMD: one mfma, other 4 ds_read", 512);
    run<5, 8>("2 waves/SIMD: one mfma, other 8 ds_read", 512);
    run<3, 8>("8 ds_read + mfma, 2 waves/SIMD both", 512);
    return 0;
}
