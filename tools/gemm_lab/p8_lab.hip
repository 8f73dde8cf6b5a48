// Lab (not product code): persistent 256 x 256 x 64 "8-phase" bf16 GEMM, C[M,N] = A[M,K] B[N,K]^T (+ bias).
//   8 waves = 2 wave groups (M halves) that run staggered by one barrier: while one group issues its 16 MFMAs
//   the other reads fragments and issues LDS-DMA (cdna_hip_programming.md §5 "The 256^2 8-phase template").
//   Here the k-tile stream does not stop at a tile boundary: one workgroup per CU walks its list of output
//   tiles, and the ring keeps prefetching the next tile's k-tiles under the current tile's last phases and
//   its epilogue.
// Build: hipcc --offload-arch=gfx950 -O3 -o p8_lab p8_lab.hip ; run: ./p8_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int T = 512, HALF = 16384, KTB = 65536, LDS_BYTES = 2 * KTB;

__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
    uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
    ua += 0x7fffu + ((ua >> 16) & 1u);
    ub += 0x7fffu + ((ub >> 16) & 1u);
    return (ua >> 16) | (ub & 0xffff0000u);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

struct Cursor {           // position of the k-tile being staged / computed in this workgroup's tile list
    int tile, k0;
    const uint16_t *a, *b; // A + m0*lda + k0, B + n0*ldb + k0 (per-thread row / chunk offsets are added on top)
};

__global__ __launch_bounds__(T) void p8_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ B,
                                               const uint16_t *__restrict__ bias, uint16_t *__restrict__ C, int M, int N,
                                               int K, int lda, int ldb, int ldc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 2, wc = wid & 3;
    const int q = lane >> 4, nl = lane & 15;
    const int NT = N / 256, ntiles = (M / 256) * NT, KT = K / 64;
    // tile list of this workgroup: round r -> tile r*G + u, u chosen so that the 32 workgroups of an XCD
    // (blockIdx % 8) take 32 consecutive tiles (they share A row panels in that XCD's L2)
    const int G = gridDim.x;
    const int per_xcd = G >> 3;
    const int u = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (u >= ntiles) return;
    const int my_tiles = (ntiles - u + G - 1) / G;

    // --- staging addresses (per thread): piece `wid` (rows 8*wid + lane/8) and piece wid+8 (64 rows further) ---
    const int r1 = 8 * wid + (lane >> 3);
    const int sc = (lane & 7) ^ ((r1 >> 1) & 7);                 // source chunk that lands in LDS chunk lane&7
    const int b_row = ((r1 & 15) >> 2) * 16 + ((r1 >> 4) & 3) * 4 + (r1 & 3);   // LDS row j*16+nl holds column (nl>>2)*16 + 4j + (nl&3)
    const size_t offA = size_t(r1) * lda + sc * 8, offB = size_t(b_row) * ldb + sc * 8;
    const size_t a64 = size_t(64) * lda, b64 = size_t(64) * ldb, a128 = size_t(128) * lda, b128 = size_t(128) * ldb;
    char *const lds_piece = smem + tid * 16;

    auto cursor_at = [&](int t, int k0) {
        Cursor c;
        c.tile = t;
        c.k0 = k0;
        const int tile = (t < my_tiles ? t : 0) * G + u;       // past the end: re-stage the first tile (never consumed)
        const int m0 = (tile / NT) * 256, n0 = (tile % NT) * 256;
        c.a = A + size_t(m0) * lda + k0;
        c.b = B + size_t(n0) * ldb + k0;
        return c;
    };
    auto advance = [&](Cursor &c) {
        int k0 = c.k0 + 64, t = c.tile;
        if (k0 == K) { k0 = 0; t++; c = cursor_at(t, 0); }
        else { c.k0 = k0; c.a += 64; c.b += 64; }
    };
    // half-tile `h` (0/1) of A or B of the cursor's k-tile -> LDS buffer `buf`
    auto stage_a = [&](const Cursor &c, int buf, int h) {
        const uint16_t *g = c.a + offA + (h ? a128 : 0);
        char *d = lds_piece + buf * KTB + h * HALF;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(g + a64), (lptr_t)(d + 8192), 16, 0, 0);
    };
    auto stage_b = [&](const Cursor &c, int buf, int h) {
        const uint16_t *g = c.b + offB + (h ? b128 : 0);
        char *d = lds_piece + buf * KTB + 2 * HALF + h * HALF;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(g + b64), (lptr_t)(d + 8192), 16, 0, 0);
    };

    // --- fragment read addresses ---
    const int f = (nl >> 1) & 7;
    const int o0 = nl * 128 + ((q ^ f) << 4);                     // k-step 0; k-step 1 is o0 ^ 64
    const int a_off[2] = {wr * HALF + o0, wr * HALF + (o0 ^ 64)};
    const int b_off[2] = {2 * HALF + (wc >> 1) * HALF + (wc & 1) * 8192 + o0, 2 * HALF + (wc >> 1) * HALF + (wc & 1) * 8192 + (o0 ^ 64)};
    bf16x8 fa0[4][2], fa1[4][2], fb0[2][2], fb1[2][2];
    auto read_a = [&](bf16x8 (&fa)[4][2], int buf, int sub) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            fa[i][0] = *reinterpret_cast<const bf16x8 *>(smem + a_off[0] + buf * KTB + (sub * 4 + i) * 2048);
            fa[i][1] = *reinterpret_cast<const bf16x8 *>(smem + a_off[1] + buf * KTB + (sub * 4 + i) * 2048);
        }
    };
    auto read_b = [&](bf16x8 (&fb)[2][2], int buf, int sub) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            fb[j][0] = *reinterpret_cast<const bf16x8 *>(smem + b_off[0] + buf * KTB + (sub * 2 + j) * 2048);
            fb[j][1] = *reinterpret_cast<const bf16x8 *>(smem + b_off[1] + buf * KTB + (sub * 2 + j) * 2048);
        }
    };

    f32x4 acc[8][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto mma = [&](const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[2][2], int asub, int bsub) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[asub * 4 + i][bsub * 2 + j] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[i][kk], acc[asub * 4 + i][bsub * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto epilogue = [&](int t) {
        const int tile = t * G + u;
        const int m0 = (tile / NT) * 256 + wr * 128 + nl, n0 = (tile % NT) * 256 + wc * 64 + q * 16;
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // MFMA -> VALU read hazard across the branch
        float bv[16];
        if (bias) {
            const u32x4 b0 = *reinterpret_cast<const u32x4 *>(bias + n0), b1 = *reinterpret_cast<const u32x4 *>(bias + n0 + 8);
#pragma unroll
            for (int d = 0; d < 4; d++) {
                bv[2 * d] = __uint_as_float(b0[d] << 16); bv[2 * d + 1] = __uint_as_float(b0[d] & 0xffff0000u);
                bv[8 + 2 * d] = __uint_as_float(b1[d] << 16); bv[8 + 2 * d + 1] = __uint_as_float(b1[d] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int d = 0; d < 16; d++) bv[d] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint16_t *row = C + size_t(m0 + i * 16) * ldc + n0;
            u32x4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t d0 = pack2bf(acc[i][j][0] + bv[4 * j], acc[i][j][1] + bv[4 * j + 1]);
                const uint32_t d1 = pack2bf(acc[i][j][2] + bv[4 * j + 2], acc[i][j][3] + bv[4 * j + 3]);
                if (j < 2) { lo[2 * j] = d0; lo[2 * j + 1] = d1; } else { hi[2 * (j - 2)] = d0; hi[2 * (j - 2) + 1] = d1; }
            }
            *reinterpret_cast<u32x4 *>(row) = lo;
            *reinterpret_cast<u32x4 *>(row + 8) = hi;
        }
    };

    // --- prologue: k-tile 0 complete, B0 / A0 / B1 of k-tile 1 in flight ---
    Cursor cs = cursor_at(0, 0);
    stage_b(cs, 0, 0); stage_a(cs, 0, 0); stage_b(cs, 0, 1); stage_a(cs, 0, 1);
    advance(cs);
    stage_b(cs, 1, 0); stage_a(cs, 1, 0); stage_b(cs, 1, 1);
    wait_vm<6>();
    bar();
    if (wr == 1) bar();                       // wave group 1 runs one barrier behind group 0
    zero_acc();

    // One k-tile = 4 phases.  `cs` is k-tile g+1 in phase 0 (its A1 is the last half-tile missing) and k-tile g+2 after.
    auto ktile = [&](int buf) {
        // phase 0: quadrant (A0, B0)
        read_b(fb0, buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(fa0, buf, 0);
        stage_a(cs, buf ^ 1, 1);
        advance(cs);
        wait_lgkm<8>();                       // the B0 reads are done: B0 of this buffer may be overwritten after the barrier
        bar();
        wait_lgkm<0>();
        mma(fa0, fb0, 0, 0);
        bar();
        // phase 1: quadrant (A0, B1)
        read_b(fb1, buf, 1);
        stage_b(cs, buf, 0);
        bar();
        wait_lgkm<0>();
        mma(fa0, fb1, 0, 1);
        bar();
        // phase 2: quadrant (A1, B1)
        read_a(fa1, buf, 1);
        stage_a(cs, buf, 0);
        bar();
        wait_lgkm<0>();
        mma(fa1, fb1, 1, 1);
        bar();
        // phase 3: quadrant (A1, B0); the other buffer (k-tile g+1) must have landed before the next phase reads it
        stage_b(cs, buf, 1);
        wait_vm<6>();
        bar();
        mma(fa1, fb0, 1, 0);
        bar();
    };
    int kt = 0, t = 0;
    const int total = my_tiles * KT;
    for (int g = 0; g < total; g += 2) {
        ktile(0);
        ktile(1);
        kt += 2;
        if (kt == KT) {
            epilogue(t);
            zero_acc();
            kt = 0;
            t++;
        }
    }
    if (wr == 0) bar();
    wait_vm<0>();
}

static uint16_t f2bf(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return uint16_t(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = uint32_t(b) << 16; float x; memcpy(&x, &u, 4); return x; }

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 50176;
    const int shapes[][2] = {{768, 768}, {2304, 768}, {3072, 768}, {768, 3072}, {768, 2304}};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)p8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    for (auto &sh : shapes) {
        const int N = sh[0], K = sh[1];
        std::vector<uint16_t> ha(size_t(M) * K), hb(size_t(N) * K), hbias(N);
        uint32_t s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float((s >> 8) & 0xffff) / 32768.f - 1.f); };
        for (auto &v : ha) v = f2bf(rnd());
        for (auto &v : hb) v = f2bf(rnd() * 0.05f);
        for (auto &v : hbias) v = f2bf(rnd());
        uint16_t *A, *B, *bias, *C;
        (void)hipMalloc(&A, ha.size() * 2); (void)hipMalloc(&B, hb.size() * 2); (void)hipMalloc(&bias, N * 2);
        (void)hipMalloc(&C, size_t(M) * N * 2);
        (void)hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(bias, hbias.data(), N * 2, hipMemcpyHostToDevice);
        (void)hipMemset(C, 0xff, size_t(M) * N * 2);
        dim3 grid(256), block(T);
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(p8_kernel, grid, block, LDS_BYTES, 0, A, B, bias, C, M, N, K, K, K, N);
        (void)hipEventRecord(e0);
        const int reps = 10;
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(p8_kernel, grid, block, LDS_BYTES, 0, A, B, bias, C, M, N, K, K, K, N);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        std::vector<uint16_t> hc(size_t(M) * N);
        (void)hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
        // check: 4096 sampled entries + every entry of 2 rows against double accumulation
        double max_err = 0; long bad = 0, checked = 0;
        auto check = [&](int m, int n) {
            double r = bf2f(hbias[n]);
            for (int k = 0; k < K; k++) r += double(bf2f(ha[size_t(m) * K + k])) * bf2f(hb[size_t(n) * K + k]);
            const double got = bf2f(hc[size_t(m) * N + n]), err = fabs(got - r);
            if (err > max_err) max_err = err;
            if (err > 0.02 + 0.01 * fabs(r)) bad++;
            checked++;
        };
        for (int i = 0; i < 4096; i++) { s = s * 1664525u + 1013904223u; const int m = (s >> 4) % M; s = s * 1664525u + 1013904223u; check(m, (s >> 4) % N); }
        for (int n = 0; n < N; n++) { check(0, n); check(M - 1, n); check(M / 2 + 131, n); }
        printf("M=%d N=%4d K=%4d: %8.1f us %7.1f TFLOP/s | checked %ld bad %ld max_err %.4f (%s)\n", M, N, K, ms * 1e3,
               2.0 * M * N * K / ms / 1e9, checked, bad, max_err, hipGetErrorString(hipGetLastError()));
        (void)hipFree(A); (void)hipFree(B); (void)hipFree(bias); (void)hipFree(C);
    }
    return 0;
}
