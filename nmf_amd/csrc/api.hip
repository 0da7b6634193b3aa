// Library-level entry points of libnmf_hip.so.
#include "common.hpp"
#include <chrono>
#include <cstring>

thread_local char nmf_err_buf[256] = "no error";

extern "C" int nmf_version(void) { return NMF_ABI_VERSION; }

extern "C" const char* nmf_last_error_string(void) { return nmf_err_buf; }

nmf_launch_probe_fn nmf_launch_probe = nullptr;
extern "C" int nmf_set_launch_probe(nmf_launch_probe_fn probe) {
    nmf_launch_probe = probe;
    return NMF_OK;
}

// ---- runtime plumbing for host-side drivers that do not include the HIP headers (csrc/host_ext.cpp is plain g++) -------
extern "C" int nmf_event_create(void** event) {
    NMF_REQUIRE(event, NMF_EINVAL, "nmf_event_create: null");
    hipEvent_t e;
    hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (r != hipSuccess) return nmf_fail((int)r, "nmf_event_create: hipEventCreateWithFlags");
    *event = (void*)e;
    return NMF_OK;
}
extern "C" int nmf_event_destroy(void* event) {
    hipError_t r = hipEventDestroy((hipEvent_t)event);
    return r == hipSuccess ? NMF_OK : nmf_fail((int)r, "nmf_event_destroy");
}
extern "C" int nmf_event_record(void* event, void* stream) {
    hipError_t r = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
    return r == hipSuccess ? NMF_OK : nmf_fail((int)r, "nmf_event_record");
}
extern "C" int nmf_event_synchronize(void* event) {
    hipError_t r = hipEventSynchronize((hipEvent_t)event);
    return r == hipSuccess ? NMF_OK : nmf_fail((int)r, "nmf_event_synchronize");
}
extern "C" int nmf_stream_wait_event(void* stream, void* event) {
    hipError_t r = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
    return r == hipSuccess ? NMF_OK : nmf_fail((int)r, "nmf_stream_wait_event");
}
extern "C" int nmf_memcpy_d2h_async(void* dst_host, const void* src_dev, int64_t nbytes, void* stream) {
    NMF_REQUIRE(dst_host && src_dev && nbytes >= 0, NMF_EINVAL, "nmf_memcpy_d2h_async: args");
    hipError_t r = hipMemcpyAsync(dst_host, src_dev, (size_t)nbytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    return r == hipSuccess ? NMF_OK : nmf_fail((int)r, "nmf_memcpy_d2h_async");
}
extern "C" int nmf_event_create_timed(void** event) {
    NMF_REQUIRE(event, NMF_EINVAL, "nmf_event_create_timed: null");
    hipEvent_t e;
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) return nmf_fail((int)r, "nmf_event_create_timed: hipEventCreate");
    *event = (void*)e;
    return NMF_OK;
}
extern "C" int nmf_event_elapsed_ms(void* start, void* stop, float* ms) {
    NMF_REQUIRE(ms, NMF_EINVAL, "nmf_event_elapsed_ms: null");
    hipError_t r = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    return r == hipSuccess ? NMF_OK : nmf_fail((int)r, "nmf_event_elapsed_ms");
}

// ---- size read-back through host memory the device writes directly ----------------------------------------------------------
// The event path (nmf_memcpy_d2h_async + nmf_event_record + nmf_event_synchronize) costs a copy command, a marker and the
// runtime's signal wait per read-back.  Here the sizes are PUBLISHED: a one-thread kernel stores them into mapped, coherent
// host memory followed by a sequence number, and the host spins on that number (nmf_wait_seq).
namespace {
__global__ void k_publish_i64x2(const int64_t* __restrict__ src, volatile int64_t* __restrict__ dst, int64_t seq) {
    dst[0] = src[0];
    dst[1] = src[1];
    __threadfence_system();
    dst[2] = seq;
}
}  // namespace
extern "C" int nmf_host_alloc_mapped(void** host_ptr, void** dev_ptr, int64_t nbytes) {
    NMF_REQUIRE(host_ptr && dev_ptr && nbytes > 0, NMF_EINVAL, "nmf_host_alloc_mapped: args");
    void* h = nullptr;
    hipError_t r = hipHostMalloc(&h, (size_t)nbytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (r != hipSuccess) return nmf_fail((int)r, "nmf_host_alloc_mapped: hipHostMalloc");
    void* d = nullptr;
    r = hipHostGetDevicePointer(&d, h, 0);
    if (r != hipSuccess) { (void)hipHostFree(h); return nmf_fail((int)r, "nmf_host_alloc_mapped: hipHostGetDevicePointer"); }
    memset(h, 0, (size_t)nbytes);
    *host_ptr = h;
    *dev_ptr = d;
    return NMF_OK;
}
extern "C" int nmf_host_free_mapped(void* host_ptr) {
    hipError_t r = hipHostFree(host_ptr);
    return r == hipSuccess ? NMF_OK : nmf_fail((int)r, "nmf_host_free_mapped");
}
extern "C" int nmf_publish_i64x2(const int64_t* src_dev, void* dst_mapped_dev, int64_t seq, void* stream) {
    NMF_REQUIRE(src_dev && dst_mapped_dev, NMF_EINVAL, "nmf_publish_i64x2: args");
    NMF_LAUNCH(k_publish_i64x2, dim3(1), dim3(1), 0, (hipStream_t)stream, src_dev, (volatile int64_t*)dst_mapped_dev, seq);
    NMF_CHECK_LAUNCH("nmf_publish_i64x2");
    return NMF_OK;
}
extern "C" int nmf_wait_seq(const void* host_ptr, int64_t seq, double timeout_s) {
    NMF_REQUIRE(host_ptr, NMF_EINVAL, "nmf_wait_seq: null");
    const int64_t* p = static_cast<const int64_t*>(host_ptr);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0;; ++spin) {
        if (__atomic_load_n(p + 2, __ATOMIC_ACQUIRE) == seq) return NMF_OK;
        if ((spin & 0xfff) == 0xfff &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
            // slow is not lost (a counter-collecting profiler serialises kernels, a shared GPU, code objects still loading): keep
            // looking under a second, longer deadline -- never a call that may not return (a device synchronisation on a wedged
            // queue would hang here for good, and on the wrong device if the caller's current device is not the buffer's)
            const double hard = timeout_s < 30.0 ? 300.0 : 10.0 * timeout_s;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > hard)
                return nmf_fail(NMF_EINVAL, "nmf_wait_seq: the sequence number never arrived (soft and hard deadline passed)");
        }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
}
