#include "common_host.h"

#include <hip/hip_runtime.h>

namespace sfcvit {
namespace {
thread_local char g_err[512] = "";
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

void clear_error() { g_err[0] = 0; }

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SFCVIT_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SFCVIT_OK;
}

}  // namespace sfcvit

extern "C" int sfcvit_abi_version(void) { return SFCVIT_ABI_VERSION; }

extern "C" const char *sfcvit_last_error(void) { return sfcvit::g_err; }

extern "C" int sfcvit_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -int(e);
    return n;
}
