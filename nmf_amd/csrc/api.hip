// Library-level entry points of libnmf_hip.so.
#include "common.hpp"

thread_local char nmf_err_buf[256] = "no error";

extern "C" int nmf_version(void) { return 100; }   // 0.1.0

extern "C" const char* nmf_last_error_string(void) { return nmf_err_buf; }
