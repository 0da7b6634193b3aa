#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash(uint32_t x){ x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }
// 8-lane groups -> 8 consecutive floats at a random 32B-aligned place (the env-map adjoint pattern)
// SCOPE 0: agent-scope atomicAdd on one table.  SCOPE 1: workgroup-scope atomics on the XCD's own copy of the table.
template <int SCOPE>
__global__ void k(float* buf, uint32_t n_floats, int iters, int* xcd_seen) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int gs = 8;
    uint32_t grp = tid / gs, sub = tid % gs;
    const int x = xcc_id();
    if (threadIdx.x == 0) xcd_seen[blockIdx.x] = x;
    float* tab = SCOPE ? buf + (size_t)x * n_floats : buf;
    for (int i = 0; i < iters; ++i) {
        uint32_t h = hash(grp * 9781u + i * 6271u + 17u);
        uint32_t base = (h % (n_floats / gs)) * gs;
        if (SCOPE) __hip_atomic_fetch_add(tab + base + sub, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else atomicAdd(tab + base + sub, 1.0f);
    }
}
__global__ void k_sum(const float* buf, uint32_t n, int copies, double* out) {
    double s = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int c = 0; c < copies; ++c) s += buf[(size_t)c * n + i];
    atomicAdd(out, s);
}
int main() {
    const uint32_t n = 4 * 512 * 1024;  // 8 MB
    float* buf; hipMalloc(&buf, (size_t)n * 4 * 8);
    int* seen; hipMalloc(&seen, 4096 * 4);
    double* out; hipMalloc(&out, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int threads = 238740 / 256 * 256, iters = 48;
    for (int scope = 0; scope < 2; ++scope) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(buf, 0, (size_t)n * 4 * 8); hipMemset(out, 0, 8);
            hipDeviceSynchronize();
            hipEventRecord(a);
            if (scope) k<1><<<threads / 256, 256>>>(buf, n, iters, seen); else k<0><<<threads / 256, 256>>>(buf, n, iters, seen);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            k_sum<<<1024, 256>>>(buf, n, scope ? 8 : 1, out);
            double h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
            printf("scope %s: %.3f ms, %.1f G lane-atomics/s, sum %.0f (expect %.0f)\n", scope ? "workgroup, per-XCD copies" : "agent", ms,
                   (double)threads * iters / ms / 1e6, h, (double)threads * iters);
        }
    }
    int hs[64]; hipMemcpy(hs, seen, 64 * 4, hipMemcpyDeviceToHost);
    printf("xcc of blocks 0..23:"); for (int i = 0; i < 24; ++i) printf(" %d", hs[i]); printf("\n");
    return 0;
}
