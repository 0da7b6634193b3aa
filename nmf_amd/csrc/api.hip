// Library-level entry points of libnmf_hip.so.
#include "common.hpp"

thread_local char nmf_err_buf[256] = "no error";

extern "C" int nmf_version(void) { return 100; }   // 0.1.0

extern "C" const char* nmf_last_error_string(void) { return nmf_err_buf; }

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
