// Loss terms of the training step for gfx950 (reference: train.py:598-601 photometric term, fields/tensoRF.py:332-340
// density L1 regulariser).  Both are tiny reductions that cost the reference ~60 elementwise launches per step
// (clip / sub / pow / sum and abs / mean over six tensors, plus their autograd mirrors); here each is one launch per
// direction.
#include "common.hpp"

namespace {

constexpr int L1_MAX = 8;

struct L1Tab {
    const float* x[L1_MAX];
    float* g[L1_MAX];
    int64_t n[L1_MAX];
};

__device__ __forceinline__ float block_sum(float v, float* ws) {
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) ws[wid] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += ws[w];
    return t;   // valid on thread 0
}

// out += sum_i mean(|x_i|): blockIdx.y = tensor, grid-stride over its elements
__global__ void __launch_bounds__(256) k_l1_fwd(L1Tab tab, float* __restrict__ out) {
    __shared__ float ws[4];
    const int i = blockIdx.y;
    const int64_t n = tab.n[i];
    const float* __restrict__ x = tab.x[i];
    float a = 0.f;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        a += fabsf(x[k]);
    const float t = block_sum(a, ws);
    if (threadIdx.x == 0 && t != 0.f) atomicAdd(out, t / (float)n);
}

// g_i = d_out * sgn(x_i) / n_i  (accumulate: added to g_i, which then already holds the other gradient of x_i)
__global__ void __launch_bounds__(256) k_l1_bwd(L1Tab tab, const float* __restrict__ d_out, int accumulate) {
    const int i = blockIdx.y;
    const int64_t n = tab.n[i];
    const float* __restrict__ x = tab.x[i];
    float* __restrict__ g = tab.g[i];
    const float s = d_out[0] / (float)n;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[k];
        const float gv = v > 0.f ? s : (v < 0.f ? -s : 0.f);
        g[k] = accumulate ? g[k] + gv : gv;
    }
}

// out += scale * w_i * sum(x_i): blockIdx.y = tensor
struct MixTab {
    const float* x[L1_MAX];
    float* g[L1_MAX];
    int64_t n[L1_MAX];
    float w[L1_MAX];
};

__global__ void __launch_bounds__(256) k_mix_fwd(MixTab tab, float scale, float* __restrict__ out) {
    __shared__ float ws[4];
    const int i = blockIdx.y;
    const int64_t n = tab.n[i];
    const float* __restrict__ x = tab.x[i];
    float a = 0.f;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) a += x[k];
    const float t = block_sum(a, ws);
    if (threadIdx.x == 0 && t != 0.f) atomicAdd(out, t * tab.w[i] * scale);
}

__global__ void __launch_bounds__(256) k_mix_bwd(MixTab tab, float scale, const float* __restrict__ d_out) {
    const int i = blockIdx.y;
    const int64_t n = tab.n[i];
    float* __restrict__ g = tab.g[i];
    const float s = d_out[0] * scale * tab.w[i];
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) g[k] = s;
}

__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// out += sum (clip(pred, 0, 1) - clip(gt, 0, 1))^2  over n floats
__global__ void __launch_bounds__(256) k_sqerr_fwd(const float* __restrict__ pred, const float* __restrict__ gt,
                                                   int64_t n, float* __restrict__ out) {
    __shared__ float ws[4];
    float a = 0.f;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const float d = clip01(pred[k]) - clip01(gt[k]);
        a += d * d;
    }
    const float t = block_sum(a, ws);
    if (threadIdx.x == 0 && t != 0.f) atomicAdd(out, t);
}

__global__ void __launch_bounds__(256) k_sqerr_bwd(const float* __restrict__ pred, const float* __restrict__ gt,
                                                   int64_t n, const float* __restrict__ d_out,
                                                   float* __restrict__ d_pred) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float p = pred[k];
    const bool pass = p >= 0.f && p <= 1.f;            // clamp backward passes the closed interval (ATen)
    d_pred[k] = pass ? 2.f * (p - clip01(gt[k])) * d_out[0] : 0.f;
}

// The photometric term and the constant adjoints of one training chunk in ONE launch (the step's critical path pays ~5 us per
// tiny launch).  Every block writes its 256 entries of d_pred exactly as k_mix_bwd + k_sqerr_bwd would (same operations in
// the same order), fills the per-ray constants g_a / g_b and leaves the sum of its squared errors in the workspace; the block
// that finishes LAST (ticket counter, reset for the next launch) adds the partial sums in block order and WRITES the loss:
// no zero fill, no float atomics, the same bits every time.
__global__ void __launch_bounds__(256) k_loss_head(const float* __restrict__ pred, const float* __restrict__ gt, int64_t n,
                                                   int64_t n_rays, const float* __restrict__ d_out, float scale, float w_pred,
                                                   float w_a, float w_b, float* __restrict__ loss,
                                                   float* __restrict__ d_pred, float* __restrict__ g_a,
                                                   float* __restrict__ g_b, uint32_t* __restrict__ ticket,
                                                   float* __restrict__ partial) {
    __shared__ float ws[4];
    __shared__ bool last;
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float dv = d_out[0];
    float sq = 0.f;
    if (k < n) {
        const float s = dv * scale * w_pred;
        const float p = pred[k], c = clip01(gt[k]);
        const float d = clip01(p) - c;
        sq = d * d;
        const bool pass = p >= 0.f && p <= 1.f;
        d_pred[k] = pass ? 2.f * (p - c) * s : 0.f;
    }
    if (k < n_rays) {
        if (g_a) g_a[k] = dv * scale * w_a;
        if (g_b) g_b[k] = dv * scale * w_b;
    }
    const float t = block_sum(sq, ws);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = t;
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    float a = 0.f;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 256) a += __builtin_nontemporal_load(partial + i);
    const float total = block_sum(a, ws);
    if (threadIdx.x == 0) {
        loss[0] = total;
        *ticket = 0u;
    }
}

static int fill_tab(L1Tab& t, const float* const x[], float* const g[], const int64_t n[], int count, bool need_g) {
    for (int i = 0; i < count; ++i) {
        if (n[i] < 0 || (n[i] > 0 && (!x[i] || (need_g && !g[i])))) return -1;
        t.x[i] = x[i];
        t.g[i] = g ? g[i] : nullptr;
        t.n[i] = n[i];
    }
    return 0;
}

static unsigned blocks_for(const int64_t n[], int count) {
    int64_t big = 1;
    for (int i = 0; i < count; ++i) big = n[i] > big ? n[i] : big;
    const int64_t b = cdiv(big, 256 * 8);
    return (unsigned)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

}  // namespace

extern "C" int nmf_l1_mean_fwd(const float* const x[], const int64_t numel[], int32_t count, float* out, void* stream) {
    NMF_REQUIRE(count >= 0 && count <= L1_MAX, NMF_ERANGE, "nmf_l1_mean_fwd: at most 8 tensors");
    NMF_REQUIRE(out && (count == 0 || (x && numel)), NMF_EINVAL, "nmf_l1_mean_fwd: null");
    if (count == 0) return NMF_OK;
    L1Tab t;
    memset(&t, 0, sizeof(t));
    NMF_REQUIRE(fill_tab(t, x, nullptr, numel, count, false) == 0, NMF_EINVAL, "nmf_l1_mean_fwd: bad tensor");
    NMF_LAUNCH(k_l1_fwd, dim3(blocks_for(numel, count), (unsigned)count), dim3(256), 0, (hipStream_t)stream, t, out);
    NMF_CHECK_LAUNCH("nmf_l1_mean_fwd");
    return NMF_OK;
}

extern "C" int nmf_l1_mean_bwd(const float* const x[], const int64_t numel[], int32_t count, const float* d_out,
                               float* const g[], int32_t accumulate, void* stream) {
    NMF_REQUIRE(count >= 0 && count <= L1_MAX, NMF_ERANGE, "nmf_l1_mean_bwd: at most 8 tensors");
    NMF_REQUIRE(d_out && (count == 0 || (x && numel && g)), NMF_EINVAL, "nmf_l1_mean_bwd: null");
    if (count == 0) return NMF_OK;
    L1Tab t;
    memset(&t, 0, sizeof(t));
    NMF_REQUIRE(fill_tab(t, x, g, numel, count, true) == 0, NMF_EINVAL, "nmf_l1_mean_bwd: bad tensor");
    NMF_LAUNCH(k_l1_bwd, dim3(blocks_for(numel, count), (unsigned)count), dim3(256), 0, (hipStream_t)stream, t,
                       d_out, (int)accumulate);
    NMF_CHECK_LAUNCH("nmf_l1_mean_bwd");
    return NMF_OK;
}

extern "C" int nmf_sqerr_fwd(const float* pred, const float* gt, int64_t n, float* out, void* stream) {
    NMF_REQUIRE(n >= 0 && out, NMF_EINVAL, "nmf_sqerr_fwd: bad argument");
    if (n == 0) return NMF_OK;
    NMF_REQUIRE(pred && gt, NMF_EINVAL, "nmf_sqerr_fwd: null");
    int64_t b = cdiv(n, 256 * 4);
    b = b > 256 ? 256 : b;
    NMF_LAUNCH(k_sqerr_fwd, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, pred, gt, n, out);
    NMF_CHECK_LAUNCH("nmf_sqerr_fwd");
    return NMF_OK;
}

extern "C" int nmf_sqerr_bwd(const float* pred, const float* gt, int64_t n, const float* d_out, float* d_pred,
                             void* stream) {
    NMF_REQUIRE(n >= 0, NMF_EINVAL, "nmf_sqerr_bwd: n < 0");
    if (n == 0) return NMF_OK;
    NMF_REQUIRE(pred && gt && d_out && d_pred, NMF_EINVAL, "nmf_sqerr_bwd: null");
    NMF_LAUNCH(k_sqerr_bwd, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, gt, n, d_out,
                       d_pred);
    NMF_CHECK_LAUNCH("nmf_sqerr_bwd");
    return NMF_OK;
}

extern "C" int nmf_loss_mix_fwd(const float* const x[], const int64_t numel[], const float w[], int32_t count, float scale,
                                float* out, void* stream) {
    NMF_REQUIRE(count >= 0 && count <= L1_MAX, NMF_ERANGE, "nmf_loss_mix_fwd: at most 8 tensors");
    NMF_REQUIRE(out && (count == 0 || (x && numel && w)), NMF_EINVAL, "nmf_loss_mix_fwd: null");
    if (count == 0) return NMF_OK;
    MixTab t;
    memset(&t, 0, sizeof(t));
    for (int i = 0; i < count; ++i) {
        NMF_REQUIRE(numel[i] >= 0 && (numel[i] == 0 || x[i]), NMF_EINVAL, "nmf_loss_mix_fwd: bad tensor");
        t.x[i] = x[i]; t.n[i] = numel[i]; t.w[i] = w[i];
    }
    NMF_LAUNCH(k_mix_fwd, dim3(blocks_for(numel, count), (unsigned)count), dim3(256), 0, (hipStream_t)stream, t,
                       scale, out);
    NMF_CHECK_LAUNCH("nmf_loss_mix_fwd");
    return NMF_OK;
}

extern "C" int nmf_loss_mix_bwd(const int64_t numel[], const float w[], int32_t count, float scale, const float* d_out,
                                float* const g[], void* stream) {
    NMF_REQUIRE(count >= 0 && count <= L1_MAX, NMF_ERANGE, "nmf_loss_mix_bwd: at most 8 tensors");
    NMF_REQUIRE(d_out && (count == 0 || (numel && w && g)), NMF_EINVAL, "nmf_loss_mix_bwd: null");
    if (count == 0) return NMF_OK;
    MixTab t;
    memset(&t, 0, sizeof(t));
    for (int i = 0; i < count; ++i) {
        NMF_REQUIRE(numel[i] >= 0 && (numel[i] == 0 || g[i]), NMF_EINVAL, "nmf_loss_mix_bwd: bad tensor");
        t.g[i] = g[i]; t.n[i] = numel[i]; t.w[i] = w[i];
    }
    NMF_LAUNCH(k_mix_bwd, dim3(blocks_for(numel, count), (unsigned)count), dim3(256), 0, (hipStream_t)stream, t,
                       scale, d_out);
    NMF_CHECK_LAUNCH("nmf_loss_mix_bwd");
    return NMF_OK;
}

// 16 bytes of ticket + one partial sum per workgroup (at least one: an empty chunk still launches a workgroup that writes loss = 0)
extern "C" int64_t nmf_loss_head_workspace_bytes(int64_t n_rays) {
    const int64_t blocks = cdiv(3 * (n_rays > 0 ? n_rays : 0), 256);
    return 16 + 4 * (blocks > 0 ? blocks : 1);
}

extern "C" int nmf_loss_head(const float* pred, const float* gt, int64_t n_rays, const float* d_out, float scale, float w_pred,
                             float w_a, float w_b, float* loss, float* d_pred, float* g_a, float* g_b, void* workspace,
                             int64_t workspace_bytes, void* stream) {
    NMF_REQUIRE(n_rays >= 0, NMF_EINVAL, "nmf_loss_head: n_rays < 0");
    NMF_REQUIRE(loss && d_out && workspace && (n_rays == 0 || (pred && gt && d_pred)), NMF_EINVAL, "nmf_loss_head: null");
    NMF_REQUIRE(workspace_bytes >= nmf_loss_head_workspace_bytes(n_rays) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
                NMF_EINVAL, "nmf_loss_head: workspace too small or not 16-byte aligned");
    const int64_t n = 3 * n_rays;
    const int64_t blocks = n ? cdiv(n, 256) : 1;
    NMF_LAUNCH(k_loss_head, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, gt, n, n_rays, d_out, scale,
                       w_pred, w_a, w_b, loss, d_pred, g_a, g_b, static_cast<uint32_t*>(workspace),
                       reinterpret_cast<float*>(static_cast<char*>(workspace) + 16));
    NMF_CHECK_LAUNCH("nmf_loss_head");
    return NMF_OK;
}
