// Measurement helper for tools/bench_busy_cus.py (NOT part of libsfcvit_hip.so): keeps `n_wgs` CUs busy for about
// `cycles` shader cycles on `stream` with a kernel whose LDS footprint (64 KiB) keeps the 8-phase GEMM's 128 KiB
// workgroups off those CUs -- a stand-in for an RCCL kernel running beside backward.  `sink`: any 4 device bytes.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/occupy/occupy.hip -o tools/occupy/liboccupy.so
#include <hip/hip_runtime.h>

namespace {
__global__ __launch_bounds__(256) void occupy_kernel(long long cycles, int *sink) {
    extern __shared__ char hog[];
    const long long t0 = __builtin_amdgcn_s_memtime();
    int acc = 0;
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) acc += hog[(threadIdx.x * 17 + acc) & 1023];
    if (acc == 0x7fffffff) *sink = acc;
}
}  // namespace

extern "C" int lab_occupy(int n_wgs, long long cycles, void *sink, void *stream) {
    if (n_wgs <= 0 || n_wgs > 256 || cycles <= 0 || cycles > (1ll << 32) || !sink) return 1;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess)
            return 2;
        attr = true;
    }
    hipLaunchKernelGGL(occupy_kernel, dim3(n_wgs), dim3(256), 65536, static_cast<hipStream_t>(stream), cycles, static_cast<int *>(sink));
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
