// One float atomic per WAVE onto ONE address (the d_mipbias pattern of k_env_lookup_bwd) against one per wave onto 64 / 4096
// different addresses and against a block-level pre-reduction:   hipcc --offload-arch=gfx950 -O3 same_addr_atomic.hip -o same_addr && ./same_addr
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* p, int n_addr, int work) {
    float v = threadIdx.x * 1e-9f;
    for (int i = 0; i < work; ++i) v = v * 1.0001f + 1e-7f;          // a little arithmetic in front, like a real kernel
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) atomicAdd(p + (wave % n_addr) * 16, v);
}
int main() {
    float* p; hipMalloc(&p, 4096 * 64); hipMemset(p, 0, 4096 * 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {187, 945, 3780}) for (int n_addr : {1, 64, 4096}) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, p, n_addr, 64);
        hipEventRecord(a);
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, p, n_addr, 64);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("blocks %5d (%6d waves) addresses %5d: %.1f us per launch\n", blocks, blocks * 4, n_addr, ms / 20 * 1e3);
    }
    return 0;
}
