// VALU issue-rate probe for gfx950: how many cycles one SIMD needs per wave64 VALU instruction as a function of the
// number of resident waves per SIMD, for the instruction kinds the attention kernels are made of.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/valu_rate.hip -o tools/probe/valu_rate && tools/probe/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void probe(float *out, int iters, unsigned long long *clk) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {          // v_fma_f32, 8 independent chains, 64 instructions per trip
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (KIND == 1) {   // v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 2) {   // v_mul_lo_u32
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                              "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(0x9E3779B1u));)
        } else if (KIND == 3) {   // v_xor_b32 (plain integer ALU)
            REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                              "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(0x9E3779B1u));)
        } else if (KIND == 4) {   // v_cvt_pk_bf16_f32
            REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                              "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 5) {   // v_mad_u32_u24 (24-bit multiply-add: full rate?)
            REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %8\n v_mad_u32_u24 %1, %1, %8, %8\n v_mad_u32_u24 %2, %2, %8, %8\n v_mad_u32_u24 %3, %3, %8, %8\n"
                              "v_mad_u32_u24 %4, %4, %8, %8\n v_mad_u32_u24 %5, %5, %8, %8\n v_mad_u32_u24 %6, %6, %8, %8\n v_mad_u32_u24 %7, %7, %8, %8"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(0x9E3779u));)
        } else if (KIND == 6) {   // v_pk_mul_f32 (2 floats per lane per instruction)
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                              : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(const double *)&m));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + float(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name) {
    float *out;
    unsigned long long *clk, h[256];
    hipMalloc(&out, 512 * 1024 * 4);
    hipMalloc(&clk, 512 * 8);
    const int iters = 2000;
    printf("%-20s", name);
    for (int cfg = 0; cfg < 5; cfg++) {
        const int threads = cfg == 0 ? 64 : cfg == 1 ? 256 : cfg == 2 ? 512 : 1024, grid = cfg == 4 ? 512 : 256;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(threads), 0, 0, out, iters, clk);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(threads), 0, 0, out, iters, clk);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < 256; i++) avg += double(h[i]);
        avg /= 256;
        const double wps = (threads < 256 ? 1.0 : threads / 256.0) * (grid / 256.0);   // waves per busy SIMD
        // wall: wave-instructions per SIMD per ns (a 2.4 GHz SIMD that retires one wave64 op per 2 cycles does 1.2)
        printf("  %gw/SIMD: %.2f ticks/instr/wave, %.3f instr/ns/SIMD", wps, avg / (iters * 64.0), wps * iters * 64.0 / (ms * 1e6));
    }
    printf("\n");
    hipFree(out);
    hipFree(clk);
}

int main() {
    run<0>("v_fma_f32");
    run<1>("v_exp_f32");
    run<2>("v_mul_lo_u32");
    run<3>("v_xor_b32");
    run<4>("v_cvt_pk_bf16_f32");
    run<5>("v_mad_u32_u24");
    run<6>("v_pk_mul_f32");
    return 0;
}
