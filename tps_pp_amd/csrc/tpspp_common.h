// Shared host-side helpers of libtpspp_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "tpspp.h"

#define TPSPP_EXPORT extern "C" __attribute__((visibility("default")))

namespace tpspp {

// thread-local message returned by tpspp_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

// hipGetLastError() after a launch -> 0 or TPSPP_EIO with the runtime's message
int check_launch(const char* what);

inline hipStream_t as_stream(tpspp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace tpspp

#define TPSPP_REQUIRE(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) return ::tpspp::fail(TPSPP_EINVAL, __VA_ARGS__); \
    } while (0)
