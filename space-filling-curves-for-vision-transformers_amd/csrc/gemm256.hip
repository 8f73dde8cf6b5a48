// Large-tile bf16 MFMA GEMM for gfx950: 256 x BN x 32 stages (BN = 256 or 128), 8 waves.
// Used by sfcvit_gemm whenever M % 256 == 0, N % BN == 0 and the k range of a workgroup is a
// multiple of 32 -- every large GEMM of ViT-B/L at the benchmark batch; the generic 128 x 128
// kernel (gemm.hip) takes everything else.
//
// Why 256-wide tiles: at 128 x 128 a workgroup moves 32 KB per 2.1 MFLOP, i.e. 64 B/clk/CU at
// full MFMA rate, more than an XCD's L2 delivers per CU; 256 x 256 halves the bytes per flop.
//
// Structure (measured motivation in DESIGN.md §5):
//  * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
//    ds_write pass) into a 4-slot ring of 32-deep stages (4 x 32 KiB).  The DMA of stage s+4 is
//    issued as soon as stage s has been read, so 2-3 stages (64-96 KB per CU) are in flight
//    across barriers: one batch in flight drained every k-tile was memory-latency bound
//    (tools/gemm_lab/dma_probe.hip: 61 -> 85 GB/s per CU with two in flight).
//  * counted waits only: `s_waitcnt vmcnt(2 stages)` + a raw s_barrier once per stage, placed in
//    the MIDDLE of the stage's 32 MFMAs; the fragments of stage s+1 are then read from LDS into a
//    second register set while the second half of stage s's MFMAs runs, so no MFMA waits on LDS.
//  * LDS-DMA writes are lane-linear (wave-uniform base + lane * 16 B), so the bank-conflict
//    swizzles of the LDS images are applied to the per-lane SOURCE address and again on the
//    fragment reads (cdna_hip_programming.md §5.4 rule 21).  Images: "kc32" [rows][32 k], 64-B
//    rows, 16-B chunk ^ 2*bit2(row) (ds_read_b128, conflict-free); "st" [32 k][128 cols]
//    sub-images as in device_common.h (ds_read_b64_tr_b16).
//
//   waves: 2 (M) x 4 (N); wave tile 128 x BN/4 = 8 x (BN/64) fragments of 16x16x32
#include "common_host.h"
#include "gemm_core.h"

namespace sfcvit {
namespace {

using namespace gemm_core;

constexpr int A_STAGE = 256 * 32 * 2;     // 16 KiB
// Two configurations:
//   BN = 256: 8 waves (2 x 4), ring of 4 x 32 KiB = 128 KiB, one workgroup per CU
//   BN = 128: 4 waves (2 x 2), ring of 3 x 24 KiB =  72 KiB, TWO workgroups per CU, so that one
//             workgroup's epilogue (30 % of a K = 768 tile) overlaps the other's k-loop
// Every wave owns a 128 x 64 tile (8 x 4 fragments) in both.
template <int BN_> struct Cfg {
    static constexpr int WN = BN_ / 64;              // waves along N
    static constexpr int T = 128 * WN;               // threads
    static constexpr int NSLOT = BN_ == 256 ? 4 : 3;
    static constexpr int STAGE = A_STAGE + BN_ * 32 * 2;
    static constexpr int G = (256 + BN_) * 4 / T;    // LDS-DMA instructions per thread per stage
    static constexpr int LDS = NSLOT * STAGE;
};

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ int kc32_off(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 2) & 1) << 1)) << 4); }

// Issue the LDS-DMA of one operand stage: ROWS x 32 (k-contiguous, "kc32" image) or 32 x ROWS
// (k-major; ROWS/128 "st" sub-images of 32 x 128).  ROWS * 4 16-byte slots, slot p = i*512 + tid
// lands at LDS byte p * 16.
template <bool KMAJOR, int ROWS, int T>
__device__ __forceinline__ void dma_stage(char *img, const uint16_t *__restrict__ src, int ld, int row0, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 4 / T; i++) {
        const int p = i * T + tid;
        const uint16_t *g;
        if (!KMAJOR) {
            const int row = p >> 2, c = (p & 3) ^ (((row >> 2) & 1) << 1);
            g = src + size_t(row0 + row) * ld + k0 + (c << 3);
        } else {
            const int half = p >> 9, krow = (p >> 4) & 31, c16 = p & 15;
            const int c32 = (c16 >> 1) ^ ((krow & 3) | (((krow >> 3) & 1) << 2));
            g = src + size_t(k0 + krow) * ld + row0 + half * 128 + (((c32 << 1) | (c16 & 1)) << 3);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(img + p * 16), 16, 0, 0);
    }
}

// 16 rows x 32 k fragment of a stage image: lane holds row (lane&15), k = 8*(lane>>4)+j.
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 frag32(const char *img, int row0, int lane) {
    if (!KMAJOR) return *reinterpret_cast<const bf16x8 *>(img + kc32_off(row0 + (lane & 15), lane >> 4));
    else return st_frag(img + (row0 >> 7) * (32 * 128 * 2), row0 & 127, 0, lane);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Wait until at most `stages` (0..3) stages of G LDS-DMA instructions each are outstanding.
template <int G>
__device__ __forceinline__ void wait_stages(int stages) {
    if (stages >= 3) wait_vm<3 * G>();
    else if (stages == 2) wait_vm<2 * G>();
    else if (stages == 1) wait_vm<G>();
    else wait_vm<0>();
}

template <bool A_KM, bool B_KM, int BN_, bool HEAVY>
__global__ __launch_bounds__(Cfg<BN_>::T, 2) void gemm256_kernel(const sfcvit_gemm_args g, int k_per_split) {
    using C = Cfg<BN_>;
    constexpr int NF = 4;                         // n-fragments per wave (wave tile 128 x 64)
    constexpr int STAGE = C::STAGE, G = C::G, NSLOT = C::NSLOT, T = C::T;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // ONE array: ring of NSLOT x [A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int tiles_n = g.N / BN_;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * BN_;
    const uint16_t *A = static_cast<const uint16_t *>(g.a);
    const uint16_t *B = static_cast<const uint16_t *>(g.b);
    const int kbeg = blockIdx.z * k_per_split;
    const int kend = min(g.K, kbeg + k_per_split);
    const int nk = (kend - kbeg) / 32;            // stages

    f32x4 acc[8][NF];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < NF; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int s) {
        char *slot = smem + (s % NSLOT) * STAGE;
        dma_stage<A_KM, 256, T>(slot, A, g.lda, m0, kbeg + s * 32, tid);
        dma_stage<B_KM, BN_, T>(slot + A_STAGE, B, g.ldb, n0, kbeg + s * 32, tid);
    };
    // Fragment registers: B double-buffered across stages, A in two halves (rows 0-63 / 64-127 of
    // the wave tile) that are re-loaded just in time -- 64 VGPRs instead of 96 for two full sets.
    bf16x8 fb0[NF], fb1[NF], fa_lo[4], fa_hi[4];
    auto read_b = [&](int s, bf16x8 (&fb)[NF]) {
        const char *slot = smem + (s % NSLOT) * STAGE + A_STAGE;
#pragma unroll
        for (int j = 0; j < NF; j++) fb[j] = frag32<B_KM>(slot, wn * 64 + j * 16, lane);
    };
    auto read_a = [&](int s, bf16x8 (&fa)[4], int half) {
        const char *slot = smem + (s % NSLOT) * STAGE;
#pragma unroll
        for (int i = 0; i < 4; i++) fa[i] = frag32<A_KM>(slot, wm * 128 + (4 * half + i) * 16, lane);
    };
    auto mma_half = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[NF], int half) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < NF; j++)
                acc[4 * half + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[4 * half + i][j], 0, 0, 0);
    };
    // One stage (fa_lo and cb hold stage s):
    //   read A rows 64-127 of stage s | 16 MFMAs (rows 0-63) | all waves: stage s+1 landed and
    //   stage s fully read -> barrier | refill the slot of stage s with stage s+NSLOT | read B and
    //   A rows 0-63 of stage s+1 | 16 MFMAs (rows 64-127).  Every LDS read is issued one MFMA
    //   half (>= 256 cycles) before its first use.
    auto stage = [&](int s, const bf16x8 (&cb)[NF], bf16x8 (&nb)[NF]) {
        read_a(s, fa_hi, 1);
        mma_half(fa_lo, cb, 0);
        if (s + 1 < nk) {
            wait_stages<G>(min(s + NSLOT - 1, nk - 1) - (s + 1));   // stages issued beyond s+1 may stay in flight
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of stage s are complete
            __builtin_amdgcn_s_barrier();
            if (s + NSLOT < nk) issue(s + NSLOT);
            read_b(s + 1, nb);
            read_a(s + 1, fa_lo, 0);
        }
        mma_half(fa_hi, cb, 1);
    };

#pragma unroll
    for (int s = 0; s < NSLOT; s++)
        if (s < nk) issue(s);
    wait_stages<G>(min(NSLOT - 1, nk - 1));
    __builtin_amdgcn_s_barrier();
    read_b(0, fb0);
    read_a(0, fa_lo, 0);
    for (int s = 0; s < nk; s += 2) {
        stage(s, fb0, fb1);
        if (s + 1 < nk) stage(s + 1, fb1, fb0);
    }
    mfma_fence();
    __syncthreads();   // every wave is done with the operand ring: the epilogue reuses it

    // acc[i][j][r] = C[m][n], m = m0 + wm*128 + i*16 + (lane&15), n = n0 + wn*64 + j*16 + 4*(lane>>4) + r
    if (gridDim.z > 1) {
        float *slab = static_cast<float *>(g.workspace) + size_t(blockIdx.z) * g.M * g.N;
#pragma unroll
        for (int j = 0; j < NF; j++) {
            const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int m = m0 + wm * 128 + i * 16 + (lane & 15);
                *reinterpret_cast<f32x4 *>(slab + size_t(m) * g.N + n) = acc[i][j];
            }
        }
        return;
    }
    epilogue_tile<8, NF, HEAVY>(g, acc, reinterpret_cast<float *>(smem) + wave * (32 * (16 * NF + 4)), m0 + wm * 128,
                                n0 + wn * 64, lane);
}

template <bool A_KM, bool B_KM, int BN_, bool HEAVY>
int launch1(const sfcvit_gemm_args &a, int splits, int k_per_split, hipStream_t s) {
    constexpr size_t lds = Cfg<BN_>::LDS;
    if (int rc = raise_lds_limit(reinterpret_cast<const void *>(&gemm256_kernel<A_KM, B_KM, BN_, HEAVY>), int(lds), "gemm256 attribute"))
        return rc;
    dim3 grid((a.M / 256) * (a.N / BN_), 1, splits), block(Cfg<BN_>::T);
    note_gemm_kernel(3, A_KM, B_KM, BN_, HEAVY);
    hipLaunchKernelGGL((gemm256_kernel<A_KM, B_KM, BN_, HEAVY>), grid, block, lds, s, a, k_per_split);
    return check_launch("gemm256");
}

template <bool A_KM, bool B_KM, int BN_>
int launch(const sfcvit_gemm_args &a, int splits, int k_per_split, hipStream_t s) {
    if (a.act == SFCVIT_ACT_GELU || a.dact == SFCVIT_ACT_GELU) return launch1<A_KM, B_KM, BN_, true>(a, splits, k_per_split, s);
    return launch1<A_KM, B_KM, BN_, false>(a, splits, k_per_split, s);
}

}  // namespace

// Called by sfcvit_gemm after argument validation.  Returns -1 when the shape is not
// eligible (caller falls back to the generic kernel), else a status code.
int gemm256_dispatch(const sfcvit_gemm_args &a, int splits, int k_per_split, hipStream_t s) {
    if (a.M % 256 || a.N % 128 || a.K % 32 || k_per_split % 32) return -1;
    // BN = 256 unless that leaves the last round of workgroups mostly idle on 256 CUs.
    bool bn256 = a.N % 256 == 0;
    if (bn256) {
        const long t = long(a.M / 256) * (a.N / 256) * splits;
        const long rounds = (t + 255) / 256;
        if (t < 200 || double(t) / double(rounds * 256) < 0.85) bn256 = false;
    }
    // Measured on the ViT-B shapes (tools/bench_gemm.py, profiles/r1): the 256 x 128 two-workgroup
    // configuration wins when both operands are k-contiguous (forward GEMMs: 730-880 TFLOP/s vs
    // 650-760 generic, 650-820 for 256 x 256); with a k-major operand (dX, dW: transposed LDS reads,
    // twice the LDS instructions) the generic kernel's 128 x 128 tiles are as fast or faster.
    if (a.force_generic == 0) {
        if (a.a_kmajor || a.b_kmajor) return -1;
        bn256 = false;
    }
    if (a.force_generic == 6) bn256 = false;             // tests / benchmarking: force the 256 x 128 configuration
    if (a.force_generic == 7 && a.N % 256 == 0) bn256 = true;
#define SFCVIT_GO(AK, BK)                                                                          \
    return bn256 ? launch<AK, BK, 256>(a, splits, k_per_split, s) : launch<AK, BK, 128>(a, splits, k_per_split, s)
    if (!a.a_kmajor && !a.b_kmajor) SFCVIT_GO(false, false);
    if (!a.a_kmajor && a.b_kmajor) SFCVIT_GO(false, true);
    if (a.a_kmajor && !a.b_kmajor) SFCVIT_GO(true, false);
    SFCVIT_GO(true, true);
#undef SFCVIT_GO
}

}  // namespace sfcvit
