// Error reporting + ABI version of libtpspp_hip.so.
#include "tpspp_common.h"

#include <cstring>

namespace tpspp {

char* err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return TPSPP_OK;
    return fail(TPSPP_EIO, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace tpspp

TPSPP_EXPORT int tpspp_abi_version(void) { return TPSPP_ABI_VERSION; }

TPSPP_EXPORT const char* tpspp_last_error(void) { return tpspp::err_buf(); }
