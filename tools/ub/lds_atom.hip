#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// LDS float atomics at random addresses of a large per-workgroup window (tools/ub/README.md): what a binned env-map adjoint
// (a SAT tile + apron accumulated in LDS, flushed once) could reach, against 156 G lane-atomics/s of the memory-side units.
__device__ __forceinline__ uint32_t hash(uint32_t x){ x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
extern __shared__ float win[];
template <int MODE>   // 0: one random address per lane-op; 1: 12 adds of one bilinear corner per lane (2x2 texels x 3 channels, row stride RS)
__global__ void __launch_bounds__(1024) k(float* out, int n, int iters, int RS, int flush) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) win[i] = 0.f;
    __syncthreads();
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        const uint32_t h = hash(tid * 9781u + i * 6271u + 17u);
        if (MODE == 0) {
            atomicAdd(&win[h % n], 1.0f);
        } else {
            const uint32_t rows = n / RS;
            const uint32_t x = h % (RS / 3 - 1), y = (h >> 12) % (rows - 1);
            float* p = win + y * RS + x * 3;
#pragma unroll
            for (int c = 0; c < 6; ++c) { atomicAdd(p + c, 0.25f); atomicAdd(p + RS + c, 0.25f); }
        }
    }
    __syncthreads();
    if (flush) {
        float* o = out + (size_t)(blockIdx.x % 64) * n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) { const float v = win[i]; if (v != 0.f) atomicAdd(o + i, v); }
    }
}
int main() {
    const int n = 96 * 128 * 3;                     // 147 456 B window
    float* buf; hipMalloc(&buf, (size_t)64 * n * 4); hipMemset(buf, 0, (size_t)64 * n * 4);
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 4);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int threads : {256, 1024}) for (int iters : {0, 64, 512}) for (int flush : {0, 1}) {
        float ms0, ms1;
        k<0><<<256, threads, n * 4>>>(buf, n, iters, 128 * 3, flush);
        hipEventRecord(a); k<0><<<256, threads, n * 4>>>(buf, n, iters, 128 * 3, flush); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms0, a, b);
        hipEventRecord(a); k<1><<<256, threads, n * 4>>>(buf, n, iters / 8, 128 * 3, flush); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms1, a, b);
        const double ops0 = 256.0 * threads * iters, ops1 = 256.0 * threads * (iters / 8) * 12;
        printf("threads %4d iters %3d flush %d: random %.1f us (%.0f G lane-ops/s) | corner x12 %.1f us (%.0f G lane-ops/s)\n", threads, iters,
               flush, ms0 * 1e3, ops0 / ms0 / 1e6, ms1 * 1e3, ops1 / ms1 / 1e6);
    }
    return 0;
}
