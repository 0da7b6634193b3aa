#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// Companion of lds_atom.hip: integer LDS atomics (ds_add_u32 / ds_add_u64) and a plain read-modify-write at random addresses
// of a per-workgroup window -- is a fixed-point accumulator faster than ds_add_f32 (200 G lane-ops/s chip-wide)?
__device__ __forceinline__ uint32_t hash(uint32_t x){ x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
extern __shared__ unsigned long long win64[];
template <int MODE>   // 0 u32 atomic, 1 u64 atomic, 2 plain u32 rmw (racy: ceiling only), 3 f32 atomic
__global__ void __launch_bounds__(1024) k(unsigned long long* out, int n, int iters) {
    uint32_t* w32 = reinterpret_cast<uint32_t*>(win64);
    float* wf = reinterpret_cast<float*>(win64);
    const int n32 = MODE == 1 ? 2 * n : n;
    for (int i = threadIdx.x; i < n32; i += blockDim.x) w32[i] = 0;
    __syncthreads();
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        const uint32_t h = hash(tid * 9781u + i * 6271u + 17u) % n;
        if (MODE == 0) atomicAdd(&w32[h], 3u);
        else if (MODE == 1) atomicAdd(&win64[h], 3ull);
        else if (MODE == 2) w32[h] += 3u;
        else atomicAdd(&wf[h], 1.0f);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = win64[0] + w32[7];
}
int main() {
    unsigned long long* buf; hipMalloc(&buf, 4096 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int n = 18432;                          // accumulators per window: 147 KB of u64, 74 KB of u32
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 8);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 8);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 8);
    hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 8);
    for (int threads : {256, 1024}) {
        const int iters = 512;
        float ms[4];
        for (int m = 0; m < 4; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (m == 0) k<0><<<256, threads, n * 8>>>(buf, n, iters);
                if (m == 1) k<1><<<256, threads, n * 8>>>(buf, n, iters);
                if (m == 2) k<2><<<256, threads, n * 8>>>(buf, n, iters);
                if (m == 3) k<3><<<256, threads, n * 8>>>(buf, n, iters);
                hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&ms[m], a, b);
            }
        }
        const double ops = 256.0 * threads * iters;
        printf("threads %4d: ds_add_u32 %.0f G/s | ds_add_u64 %.0f G/s | plain rmw u32 %.0f G/s | ds_add_f32 %.0f G/s (lane-ops)\n", threads,
               ops / ms[0] / 1e6, ops / ms[1] / 1e6, ops / ms[2] / 1e6, ops / ms[3] / 1e6);
    }
    return 0;
}
