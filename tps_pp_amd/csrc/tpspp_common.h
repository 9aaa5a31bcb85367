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

// Per-device one-off set-up (hipFuncSetAttribute for > 64 KB of dynamic LDS is a per-device property of a kernel):
//     static bool done[tpspp::kMaxDevices] = {};  if (tpspp::first_use_on_device(done)) { ... }
constexpr int kMaxDevices = 64;
inline bool first_use_on_device(bool (&done)[kMaxDevices])
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return true;   // unknown device: always set up
    if (done[dev]) return false;
    done[dev] = true;
    return true;
}

}  // namespace tpspp

#define TPSPP_REQUIRE(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) return ::tpspp::fail(TPSPP_EINVAL, __VA_ARGS__); \
    } while (0)
