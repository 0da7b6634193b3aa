// What ds_read_b64_tr_b16 returns (gfx950): LDS holds the 16-bit value i at element i; lane l reads 8 bytes at a per-lane address
// given by one of a few patterns; the four 16-bit values each lane receives are printed.
//   hipcc --offload-arch=gfx950 -O2 tr_read.hip -o tr_read && ./tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k(int pattern, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    uint32_t addr;
    if (pattern == 0) addr = l * 8;                                   // contiguous: lane l -> elements 4l .. 4l+3
    else if (pattern == 1) addr = (l & 3) * 256 + (l >> 2) * 8;       // lanes 4q+j: row j (128 elements apart), 4 columns at 4q
    else addr = (l & 15) * 256 + (l >> 4) * 8;                        // lanes 16g+j: row j, columns 4g
    addr += (uint32_t)(uintptr_t)lds;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}
