#include "common_host.h"

#include <hip/hip_runtime.h>
#include <mutex>

namespace sfcvit {
namespace {
thread_local char g_err[512] = "";
thread_local int g_gemm[5] = {0, 0, 0, 0, 0};
thread_local char g_attn[96] = "none";
thread_local char g_rowwise[96] = "none";
}

void note_rowwise_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_rowwise, sizeof(g_rowwise), fmt, ap);
    va_end(ap);
}

void note_attn_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_attn, sizeof(g_attn), fmt, ap);
    va_end(ap);
}

bool gemm_fused_colsum() { return g_gemm[0] == 1 && (g_gemm[2] & 16); }
bool gemm_fused_actmask() { return g_gemm[0] == 1 && (g_gemm[2] & 32); }

void note_gemm_kernel(int family, int a, int b, int c, int d) {
    g_gemm[0] = family; g_gemm[1] = a; g_gemm[2] = b; g_gemm[3] = c; g_gemm[4] = d;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

void clear_error() { g_err[0] = 0; }

// hipFuncSetAttribute applies to the kernel on the CURRENT device only: one flag per (kernel, device).
int raise_lds_limit(const void *kernel, int bytes, const char *what) {
    struct Entry { const void *fn; uint64_t devs; };
    static Entry table[128];
    static int used = 0;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(SFCVIT_ENODEV, "%s: no current HIP device", what);
    std::lock_guard<std::mutex> lock(mu);
    Entry *e = nullptr;
    for (int i = 0; i < used && !e; i++)
        if (table[i].fn == kernel) e = &table[i];
    if (e && (e->devs >> dev & 1)) return SFCVIT_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return check_launch(what);
    if (!e && used < 128) { e = &table[used++]; e->fn = kernel; e->devs = 0; }
    if (e) e->devs |= uint64_t(1) << dev;          // a full table only costs the repeated (idempotent) call
    return SFCVIT_OK;
}

// Compute units of the current device (cached per device); 0 if it cannot be told.
int device_cu_count() {
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus[dev] = prop.multiProcessorCount;
    }
    return cus[dev];
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SFCVIT_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SFCVIT_OK;
}

}  // namespace sfcvit

extern "C" int sfcvit_abi_version(void) { return SFCVIT_ABI_VERSION; }

extern "C" const char *sfcvit_last_error(void) { return sfcvit::g_err; }

extern "C" int sfcvit_last_gemm_kernel(char *buf, int n) {
    using sfcvit::g_gemm;
    if (!buf || n <= 0) return SFCVIT_EINVAL;
    const char *tf[2] = {"false", "true"};
    switch (g_gemm[0]) {
    case 1: snprintf(buf, size_t(n), "gemm8p_kernel<%d, %d, %s>", g_gemm[1], g_gemm[2], tf[g_gemm[3] & 1]); break;
    case 2: snprintf(buf, size_t(n), "gemm8p_km_kernel<%s>", tf[g_gemm[1] & 1]); break;
    case 3: snprintf(buf, size_t(n), "gemm256_kernel<%s, %s, %d, %s>", tf[g_gemm[1] & 1], tf[g_gemm[2] & 1], g_gemm[3], tf[g_gemm[4] & 1]); break;
    case 4: snprintf(buf, size_t(n), "gemm_kernel<%s, %s, %s>", tf[g_gemm[1] & 1], tf[g_gemm[2] & 1], tf[g_gemm[3] & 1]); break;
    default: snprintf(buf, size_t(n), "none"); break;
    }
    return SFCVIT_OK;
}

extern "C" int sfcvit_last_attn_kernel(char *buf, int n) {
    if (!buf || n <= 0) return SFCVIT_EINVAL;
    snprintf(buf, size_t(n), "%s", sfcvit::g_attn);
    return SFCVIT_OK;
}

extern "C" int sfcvit_last_rowwise_kernel(char *buf, int n) {
    if (!buf || n <= 0) return SFCVIT_EINVAL;
    snprintf(buf, size_t(n), "%s", sfcvit::g_rowwise);
    return SFCVIT_OK;
}

extern "C" int sfcvit_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -int(e);
    return n;
}
