// In-kernel clock of an MFMA-dense loop on MI355X (MI355X_MICROARCH.md "DVFS give-back" item 6): the chip lowers its clock
// under matrix load, so "2.5 PFLOP/s" (2.4 GHz x 256 CUs x 4 SIMDs x 1024 FLOP/clock) is not what a bf16 loop can reach.
// Every workgroup stamps s_memtime (shader clocks) and s_memrealtime (100 MHz) around a loop of independent
// v_mfma_f32_16x16x32_bf16 on random operands; clock = d(memtime) / d(memrealtime) x 100 MHz, median over workgroups,
// after >= 2 s of back-to-back launches.  Three bodies: MFMA only; MFMA + the ds_read_b128 fragment reads of a GEMM phase;
// idle-ish (s_sleep) for the unloaded clock.
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int BODY>
__global__ __launch_bounds__(256) void probe(const bf16x8 *__restrict__ src, float *__restrict__ sink, unsigned long long *__restrict__ stamps, int iters) {
    __shared__ bf16x8 lds[2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 256) lds[i] = src[(blockIdx.x * 2048 + i) & 0xFFFF];
    __syncthreads();
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; i++) { a[i] = lds[(tid + 64 * i) & 2047]; b[i] = lds[(tid * 3 + 64 * i + 7) & 2047]; }
    f32x4 acc[16];
    for (int i = 0; i < 16; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (BODY == 2) {
            __builtin_amdgcn_s_sleep(64);
        } else {
            if (BODY == 1) {                     // 8 fragment reads per 16 MFMAs, as a GEMM phase
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    a[i] = lds[(tid + 64 * i + it) & 2047];
                    b[i] = lds[(tid * 3 + 64 * i + it) & 2047];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[4 * i + j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int BODY>
static void run(const char *name, const bf16x8 *src, float *sink, unsigned long long *stamps, int blocks, int iters, double seconds) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f, total = 0.f;
    int launches = 0;
    while (total < seconds * 1e3) {              // keep the chip under this load before reading the stamps
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(probe<BODY>, dim3(blocks), dim3(256), 0, 0, src, sink, stamps, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        total += ms;
        launches += 20;
    }
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (int b = 0; b < blocks; b++) mhz.push_back(double(h[2 * b]) / double(h[2 * b + 1]) * 100.0);
    std::sort(mhz.begin(), mhz.end());
    const double per_launch_ms = ms / 20.0;
    const double flops = BODY == 2 ? 0.0 : double(blocks) * 4 /*waves*/ * iters * 16.0 * (2.0 * 16 * 16 * 32);
    printf("%-34s clock median %7.1f MHz (min %7.1f, max %7.1f)   %8.3f ms / launch   %7.1f TFLOP/s   [%d launches]\n", name, mhz[blocks / 2],
           mhz.front(), mhz.back(), per_launch_ms, flops / (per_launch_ms * 1e-3) / 1e12, launches);
}

int main() {
    const int blocks = 256 * 4;                  // 4 workgroups of 4 waves per CU: one... four waves per SIMD
    bf16x8 *src;
    float *sink;
    unsigned long long *stamps;
    hipMalloc(&src, 65536 * sizeof(bf16x8));
    hipMalloc(&sink, blocks * 256 * sizeof(float));
    hipMalloc(&stamps, 2 * blocks * sizeof(unsigned long long));
    std::vector<unsigned short> h(65536 * 8);
    srand(1);
    for (auto &v : h) {                          // random bf16 in roughly [-2, 2)
        const float f = (rand() / float(RAND_MAX)) * 4.f - 2.f;
        unsigned u;
        memcpy(&u, &f, 4);
        v = (unsigned short)(u >> 16);
    }
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<2>("idle (s_sleep)", src, sink, stamps, blocks, 2000, 0.5);
    run<0>("MFMA only, random operands", src, sink, stamps, blocks, 20000, 2.0);
    run<1>("MFMA + fragment reads (LDS)", src, sink, stamps, blocks, 20000, 2.0);
    run<2>("idle again", src, sink, stamps, blocks, 2000, 0.5);
    return 0;
}
