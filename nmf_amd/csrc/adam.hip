// Adam step for every parameter tensor of the scene in ONE launch (reference: train.py:443-469 builds a
// torch.optim.Adam over ~31 tensors in 11 param groups; its foreach implementation issues ~13 kernels per group
// = ~160 launches per optimizer step, more host time than the 14 MB of state deserves).
//
// The host passes a table of slots (pointers + per-group hyper-parameters, already bias-corrected as torch does
// on the host in double precision); slots travel BY VALUE in the kernel argument buffer, so there is no H2D copy
// and no lifetime problem.  blockIdx.y = slot, blockIdx.x grid-strides over the tensor.  The update is torch's
// _multi_tensor_adam (optim/adam.py, amsgrad=False, maximize=False, capturable=False):
//   g   = grad + weight_decay * p
//   m   = m + (g - m) * (1 - beta1)                          (lerp_)
//   v   = v * beta2 + (1 - beta2) * g * g                    (mul_, addcmul_)
//   p   = p - step_size * m / (sqrt(v) / bc2_sqrt + eps)     (sqrt, div_, add_, addcdiv_)
// in the parameter's own dtype (fp32 tables, fp64 for the three env-map scalars).  Elements with a non-finite gradient
// are skipped (see adam_slot).
#include "common.hpp"

namespace {

constexpr int SLOTS_PER_LAUNCH = 40;

struct Batch {
    nmf_adam_slot s[SLOTS_PER_LAUNCH];
};

template <typename T>
__device__ __forceinline__ void adam_one(T& p, T g, T& m, T& v, T wd, T omb1, T b2, T omb2, T step, T bc2s, T eps) {
    if (wd != T(0)) g = g + wd * p;
    // at::lerp: weight < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
    const T diff = g - m;
    m = (omb1 < T(0.5)) ? m + omb1 * diff : g - diff * (T(1) - omb1);
    v = v * b2 + omb2 * g * g;
    const T denom = sqrt(v) / bc2s + eps;
    p = p + (-step) * (m / denom);
}

template <typename T>
__device__ void adam_slot(const nmf_adam_slot& s) {
    T* __restrict__ p = static_cast<T*>(s.param);
    const T* __restrict__ g = static_cast<const T*>(s.grad);
    T* __restrict__ m = static_cast<T*>(s.exp_avg);
    T* __restrict__ v = static_cast<T*>(s.exp_avg_sq);
    const T wd = (T)s.weight_decay, omb1 = (T)(1.0 - (double)s.beta1), b2 = (T)s.beta2,
            omb2 = (T)(1.0 - (double)s.beta2), step = (T)s.step_size, bc2s = (T)s.bc2_sqrt, eps = (T)s.eps;
    const int64_t n = s.numel;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if constexpr (sizeof(T) == 4) {
        // 16 bytes per lane and array (R4): the update streams 28 B per parameter through HBM -- 3.3 TB/s with 4-byte accesses
        if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
              reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
            const int64_t n4 = n >> 2;
            float4* p4 = reinterpret_cast<float4*>(p);
            const float4* g4 = reinterpret_cast<const float4*>(g);
            float4* m4 = reinterpret_cast<float4*>(m);
            float4* v4 = reinterpret_cast<float4*>(v);
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
                const float4 gq = g4[i];
                float4 pq = p4[i], mq = m4[i], vq = v4[i];
                float gg[4] = {gq.x, gq.y, gq.z, gq.w}, pp[4] = {pq.x, pq.y, pq.z, pq.w}, mm[4] = {mq.x, mq.y, mq.z, mq.w},
                      vv[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (isfinite(gg[q])) adam_one<T>(pp[q], gg[q], mm[q], vv[q], wd, omb1, b2, omb2, step, bc2s, eps);
                p4[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
                m4[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
                v4[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
            }
            done = n4 << 2;
        }
    }
    for (int64_t i = done + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T gg = g[i];
        // last line of defence (the step-level guard below is the first): an element whose gradient is not finite keeps its
        // parameter and its moments
        if (!isfinite(gg)) continue;
        T pp = p[i], mm = m[i], vv = v[i];
        adam_one<T>(pp, gg, mm, vv, wd, omb1, b2, omb2, step, bc2s, eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

// guard: a device float, e.g. the summed loss of the step's chunks.  train.py:704-705 drops a chunk whose loss is NaN after a
// host read-back of the loss; here the loss stays on the device and a non-finite value makes the WHOLE update a no-op
// (parameters, moments): the gradients of such a step are contaminated whatever chunk produced the NaN, and with one chunk
// per step -- the usual case -- this is the reference's behaviour without its synchronisation.
__global__ void __launch_bounds__(256) k_adam(Batch b, const float* __restrict__ guard) {
    if (guard && !isfinite(*guard)) return;
    const nmf_adam_slot& s = b.s[blockIdx.y];
    if ((int64_t)blockIdx.x * blockDim.x >= s.numel) return;      // (no element of any loop below starts at or beyond numel)
    if (s.is_f64) adam_slot<double>(s);
    else adam_slot<float>(s);
}

// multi-tensor copy with fp32 <-> fp64 conversion: gradient pack / unpack around the all-reduce in ONE launch each
constexpr int COPY_SLOTS = 96;
struct CopyBatch {
    nmf_copy_slot s[COPY_SLOTS];
};

template <typename S, typename D>
__device__ __forceinline__ void copy_slot(const nmf_copy_slot& s) {
    const S* __restrict__ src = static_cast<const S*>(s.src);
    D* __restrict__ dst = static_cast<D*>(s.dst);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < s.numel; i += stride) dst[i] = (D)src[i];
}

// fp32 -> bf16, round to nearest even (NaN stays NaN)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__device__ __forceinline__ void copy_slot_bf16(const nmf_copy_slot& s) {
    const float* __restrict__ src = static_cast<const float*>(s.src);
    uint16_t* __restrict__ dst = static_cast<uint16_t*>(s.dst);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < s.numel; i += stride) dst[i] = f32_to_bf16(src[i]);
}

__global__ void __launch_bounds__(256) k_multi_copy(CopyBatch b) {
    const nmf_copy_slot& s = b.s[blockIdx.y];
    if ((int64_t)blockIdx.x * blockDim.x >= s.numel) return;
    if (s.dst_is_f64 == 2) {             // bf16 destination (fp32 source): the bf16 table copies of configs[1]
        copy_slot_bf16(s);
        return;
    }
    if (s.src_is_f64) {
        if (s.dst_is_f64) copy_slot<double, double>(s);
        else copy_slot<double, float>(s);
    } else {
        if (s.dst_is_f64) copy_slot<float, double>(s);
        else copy_slot<float, float>(s);
    }
}

}  // namespace

extern "C" int nmf_multi_copy(const nmf_copy_slot* slots, int32_t n_slots, void* stream) {
    NMF_REQUIRE(n_slots >= 0 && (slots || n_slots == 0), NMF_EINVAL, "nmf_multi_copy: bad slot table");
    for (int32_t i = 0; i < n_slots; ++i)
        NMF_REQUIRE(slots[i].numel >= 0 && (slots[i].numel == 0 || (slots[i].src && slots[i].dst)), NMF_EINVAL,
                    "nmf_multi_copy: null tensor pointer");
    for (int32_t base = 0; base < n_slots; base += COPY_SLOTS) {
        const int32_t n = (n_slots - base < COPY_SLOTS) ? n_slots - base : COPY_SLOTS;
        CopyBatch b;
        memset(&b, 0, sizeof(b));
        int64_t biggest = 0;
        for (int32_t i = 0; i < n; ++i) {
            b.s[i] = slots[base + i];
            if (b.s[i].numel > biggest) biggest = b.s[i].numel;
        }
        if (biggest == 0) continue;
        int64_t bx = cdiv(biggest, 256 * 4);
        bx = bx > 1024 ? 1024 : (bx < 1 ? 1 : bx);
        NMF_LAUNCH(k_multi_copy, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, b);
        NMF_CHECK_LAUNCH("k_multi_copy");
    }
    return NMF_OK;
}

extern "C" int nmf_adam_step(const nmf_adam_slot* slots, int32_t n_slots, void* stream) {
    return nmf_adam_step_guarded(slots, n_slots, nullptr, stream);
}

extern "C" int nmf_adam_step_guarded(const nmf_adam_slot* slots, int32_t n_slots, const float* guard, void* stream) {
    NMF_REQUIRE(slots != nullptr || n_slots == 0, NMF_EINVAL, "nmf_adam_step: null slot table");
    NMF_REQUIRE(n_slots >= 0, NMF_EINVAL, "nmf_adam_step: negative slot count");
    for (int32_t i = 0; i < n_slots; ++i) {
        const nmf_adam_slot& s = slots[i];
        NMF_REQUIRE(s.numel >= 0, NMF_EINVAL, "nmf_adam_step: negative numel");
        NMF_REQUIRE(s.numel == 0 || (s.param && s.grad && s.exp_avg && s.exp_avg_sq), NMF_EINVAL,
                    "nmf_adam_step: null tensor pointer");
    }
    for (int32_t base = 0; base < n_slots; base += SLOTS_PER_LAUNCH) {
        const int32_t n = (n_slots - base < SLOTS_PER_LAUNCH) ? n_slots - base : SLOTS_PER_LAUNCH;
        Batch b;
        memset(&b, 0, sizeof(b));
        int64_t biggest = 0;
        for (int32_t i = 0; i < n; ++i) {
            b.s[i] = slots[base + i];
            if (b.s[i].numel > biggest) biggest = b.s[i].numel;
        }
        if (biggest == 0) continue;
        int64_t bx = cdiv(biggest, 256 * 4);
        if (bx > 1024) bx = 1024;
        if (bx < 1) bx = 1;
        NMF_LAUNCH(k_adam, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, b, guard);
        NMF_CHECK_LAUNCH("k_adam");
    }
    return NMF_OK;
}
