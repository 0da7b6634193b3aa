// Does hipExtAnyOrderLaunch let two consecutive, independent kernels of ONE stream overlap on gfx950?
// (hip_ext.h says the flag "is not supported on AMD GFX9xx boards" for the module API; measured here for hipExtLaunchKernel.)
//   hipcc --offload-arch=gfx950 -O2 tools/ub/anyorder.hip -o /tmp/anyorder && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}
int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const long long cyc = 5000;   // wall_clock64 ticks at 100 MHz: 50 us
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: A, B both ordered; 1: B any-order; 2: A and B any-order
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, st);
            for (int i = 0; i < 20; ++i) {
                hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, nullptr, nullptr, mode == 2 ? hipExtAnyOrderLaunch : 0, cyc, (int*)nullptr);
                hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, nullptr, nullptr, mode >= 1 ? hipExtAnyOrderLaunch : 0, cyc, (int*)nullptr);
            }
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("mode %d: %.1f us per pair of 50 us kernels\n", mode, ms * 1e3 / 20);
        }
    }
    // tiny dependent kernels: per-launch latency with and without the flag
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, st);
            for (int i = 0; i < 200; ++i)
                hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, 0LL, (int*)nullptr);
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("tiny kernels, any-order %d: %.2f us per launch\n", mode, ms * 1e3 / 200);
        }
    }
    return 0;
}
