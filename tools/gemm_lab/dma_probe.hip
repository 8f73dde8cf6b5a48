// Lab probe (not product code): how fast can a CU pull GEMM operand tiles into LDS with
// global_load_lds_dwordx4, as a function of how many tile-batches are kept in flight?
//   mode 0: issue one 64 KB batch, vmcnt(0), barrier (what the 2-buffer kernel does)
//   mode 1: keep 2 batches in flight (vmcnt(8)) -- needs 2 buffers
//   mode 2: keep 1.5 .. (vmcnt(4))
// The tile addressing is exactly gemm256's kc x kc pattern for the QKV shape.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int ROWS>
__device__ __forceinline__ void dma_tile(char *img, const uint16_t *src, int ld, int row0, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / 512; i++) {
        const int p = i * 512 + tid;
        const int row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
        const uint16_t *g = src + size_t(row0 + row) * ld + k0 + (c << 3);
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(img + p * 16), 16, 0, 0);
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const uint16_t *A, const uint16_t *B, int M, int N, int K, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int tiles_n = N / 256;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
    const int nk = K / 64;
    for (int kt = 0; kt < nk; kt++) {
        char *buf = smem + (kt & 1) * 65536;
        dma_tile<256>(buf, A, K, m0, kt * 64, tid);
        dma_tile<256>(buf + 32768, B, K, n0, kt * 64, tid);
        if (MODE == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if (MODE == 1) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if (MODE == 2) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink && tid == 0 && blockIdx.x == 99999) sink[0] = ((float *)smem)[tid];
}

int main(int argc, char **argv) {
    const int M = 50176, N = argc > 1 ? atoi(argv[1]) : 2304, K = argc > 2 ? atoi(argv[2]) : 768;
    uint16_t *A, *B;
    hipMalloc(&A, size_t(M) * K * 2);
    hipMalloc(&B, size_t(N) * K * 2);
    hipMemset(A, 1, size_t(M) * K * 2);
    hipMemset(B, 1, size_t(N) * K * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dim3 grid((M / 256) * (N / 256)), block(512);
    auto run = [&](int mode) {
        auto k = mode == 0 ? probe<0> : mode == 1 ? probe<1> : probe<2>;
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k, grid, block, 131072, 0, A, B, M, N, K, (float *)nullptr);
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k, grid, block, 131072, 0, A, B, M, N, K, (float *)nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        double bytes = double(grid.x) * (K / 64) * 65536.0;
        printf("N=%d K=%d mode %d: %.1f us, %.2f TB/s LDS-DMA intake (%.1f GB/s per CU), err=%s\n", N, K, mode, ms * 1e3,
               bytes / ms / 1e9, bytes / ms / 1e6 / 256, hipGetErrorString(hipGetLastError()));
    };
    for (int mode = 0; mode < 3; mode++) run(mode);
    return 0;
}
