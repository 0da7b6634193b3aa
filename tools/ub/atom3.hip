#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// Env-map adjoint scatter, two table layouts (tools/ub/README.md):
//  A  [H][W][4]: a corner = two instructions (rows y0, y0+1), 8 lanes each on one 32-byte run (2 texels x 4 channels)
//  B  [H/2][W][2][4]: rows 2k, 2k+1 of a column are adjacent: a corner with even y0 = ONE instruction of 16 lanes on a
//     64-byte run, odd y0 = two instructions of 8 lanes (two 16-byte pieces 32 bytes apart each)
__device__ __forceinline__ uint32_t hash(uint32_t x){ x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
constexpr int H = 512, W = 1024;
template <int LAYOUT>
__global__ void k(float* buf, int iters) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (LAYOUT == 0) {
        const uint32_t grp = tid >> 3, t = tid & 7, dx = t >> 2, ch = t & 3;
        for (int i = 0; i < iters; ++i) {
            const uint32_t h = hash(grp * 9781u + i * 6271u + 17u);
            const uint32_t x0 = h % (W - 1), y0 = (h >> 12) % (H - 1);
            float* p = buf + ((size_t)y0 * W + x0 + dx) * 4 + ch;
            atomicAdd(p, 1.0f);
            atomicAdd(p + (size_t)W * 4, 1.0f);
        }
    } else {
        const uint32_t grp = tid >> 4, t = tid & 15, dy = t >> 3, dx = (t >> 2) & 1, ch = t & 3;
        for (int i = 0; i < iters; ++i) {
            const uint32_t h = hash(grp * 9781u + i * 6271u + 17u);
            const uint32_t x0 = h % (W - 1), y0 = (h >> 12) % (H - 1);
            const uint32_t y = y0 + dy;
            float* p = buf + ((((size_t)(y >> 1) * W + x0 + dx) * 2 + (y & 1)) * 4) + ch;
            atomicAdd(p, 1.0f);
        }
    }
}
int main() {
    float* buf; hipMalloc(&buf, (size_t)H * W * 4 * 4); hipMemset(buf, 0, (size_t)H * W * 4 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int corners = 242432 / 8 * 8, iters = 5;        // lookups x ~5 corners
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        hipEventRecord(a); k<0><<<corners * 8 / 256, 256>>>(buf, iters); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); printf("layout A [H][W][4]      : %.1f us\n", ms * 1e3);
        hipEventRecord(a); k<1><<<corners * 16 / 256, 256>>>(buf, iters); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); printf("layout B [H/2][W][2][4] : %.1f us\n", ms * 1e3);
    }
    return 0;
}
