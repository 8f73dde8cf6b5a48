// Lab probe: time to write a [M, N] bf16 matrix as 256x256 tiles (one workgroup of 512 threads per
// tile, 16-byte stores, 8 lanes per 128-byte row segment), as a function of the LDS the workgroup
// reserves (0 => many workgroups per CU, 128 KB => one) and of an optional prologue DMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int PRO>
__global__ __launch_bounds__(512) void store_tiles(uint16_t *C, const uint16_t *A, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = N / 256;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
    if (PRO && PRO < 4) {   // one k-tile of DMA like the GEMM prologue
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = i * 512 + tid, row = p >> 3, c = p & 7;
            __builtin_amdgcn_global_load_lds((gptr_t)(A + size_t(m0 + row) * K + c * 8), (lptr_t)(smem + p * 16), 16, 0, 0);
        }
        __syncthreads();
    }
    const int wm = wave >> 2, wn = wave & 3;
    if (PRO >= 4) {
        // register epilogues of gemm8p: lane (q = lane >> 4, nl = lane & 15) holds row 16 i + nl of fragment row i.
        //   4: its 16 consecutive columns 16 q .. 16 q + 15 (two 16-byte stores 16 B apart)           [the product]
        //   5: columns 8 q .. 8 q + 7 and 32 + 8 q .. (two stores 64 B apart: a store instruction covers 64-byte runs)
        const int q4 = lane >> 4, nl = lane & 15;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            u32x4 v = {uint32_t(i), uint32_t(q4), 3u, 4u};
            uint16_t *c = C + size_t(m0 + wm * 128 + 16 * i + nl) * N + n0 + wn * 64;
            if (PRO == 4) {
                *reinterpret_cast<u32x4 *>(c + 16 * q4) = v;
                *reinterpret_cast<u32x4 *>(c + 16 * q4 + 8) = v;
            } else {
                *reinterpret_cast<u32x4 *>(c + 8 * q4) = v;
                *reinterpret_cast<u32x4 *>(c + 8 * q4 + 32) = v;
            }
        }
    } else if (PRO < 2) {
    // wave tile 128 x 64: 16 instructions, each 8 rows x 128 B
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int row = k * 8 + (lane >> 3), cg = lane & 7;
        u32x4 v = {uint32_t(row), uint32_t(cg), 3u, 4u};
        *reinterpret_cast<u32x4 *>(C + size_t(m0 + wm * 128 + row) * N + n0 + wn * 64 + cg * 8) = v;
    }
    } else {
        // the product's epilogue_tile<8,4>: 4 passes of 32 rows through a padded fp32 LDS patch
        typedef __attribute__((ext_vector_type(4))) float f32x4;
        float *patch = reinterpret_cast<float *>(smem) + wave * (32 * 68);
        f32x4 acc[8][4];
        for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) acc[i][j] = f32x4{float(i), float(j), float(lane), 1.f};
#pragma unroll
        for (int p = 0; p < 4; p++) {
#pragma unroll
            for (int ii = 0; ii < 2; ii++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<f32x4 *>(patch + (16 * ii + (lane & 15)) * 68 + 16 * j + 4 * (lane >> 4)) = acc[2 * p + ii][j];
            if (PRO == 2) __syncthreads(); else __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int row = (k * 64 + lane) / 8, cg = lane & 7;
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(patch + row * 68 + 8 * cg);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(patch + row * 68 + 8 * cg + 4);
                u32x4 v = {__float_as_uint(lo[0] + lo[1]), __float_as_uint(lo[2] + lo[3]), __float_as_uint(hi[0] + hi[1]), __float_as_uint(hi[2] + hi[3])};
                *reinterpret_cast<u32x4 *>(C + size_t(m0 + wm * 128 + 32 * p + row) * N + n0 + wn * 64 + cg * 8) = v;
            }
            if (PRO == 2) __syncthreads(); else __builtin_amdgcn_wave_barrier();
        }
    }
}

int main(int argc, char **argv) {
    const int M = 50176, N = argc > 1 ? atoi(argv[1]) : 2304, K = 768;
    uint16_t *C, *A;
    (void)hipMalloc(&C, size_t(M) * N * 2);
    (void)hipMalloc(&A, size_t(M) * K * 2);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    dim3 grid((M / 256) * (N / 256)), block(512);
    for (int pro = 0; pro < 6; pro++)
        for (int lds : {0, 81920, 131072}) {
            if (pro && lds < 32768) continue;
            auto k = pro == 0 ? store_tiles<0> : pro == 1 ? store_tiles<1> : pro == 2 ? store_tiles<2> : pro == 3 ? store_tiles<3> : pro == 4 ? store_tiles<4> : store_tiles<5>;
            (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k, grid, block, lds, 0, C, A, M, N, K);
            (void)hipEventRecord(e0);
            for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k, grid, block, lds, 0, C, A, M, N, K);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            printf("N=%d prologue=%d lds=%6d: %.1f us  %.2f TB/s\n", N, pro, lds, ms * 1e3, double(M) * N * 2 / ms / 1e9);
        }
    return 0;
}
