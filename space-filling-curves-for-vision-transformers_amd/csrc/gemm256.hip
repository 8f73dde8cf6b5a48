// Large-tile bf16 MFMA GEMM for gfx950: 256 x BN x 64 tiles (BN = 256 or 128), 8 waves,
// operands staged HBM -> LDS directly with global_load_lds_dwordx4 (LDS-DMA: no staging
// registers, no ds_write pass), two LDS buffers, the DMA of k-tile t+1 in flight under the
// MFMAs of k-tile t.  Used by sfcvit_gemm whenever M % 256 == 0, N % BN == 0 and the k range
// of a workgroup is a multiple of 64 -- every large GEMM of ViT-B/L at the benchmark batch --
// and the generic 128 x 128 kernel (gemm.hip) takes everything else.
//
// Why 256-wide tiles: at 128 x 128 x 64 a workgroup moves 32 KB per 2.1 MFLOP, i.e. 64 B/clk/CU
// at full MFMA rate, more than an XCD's L2 delivers per CU; 256 x 256 halves the bytes per flop.
//
// LDS-DMA writes are lane-linear (wave-uniform base + lane * 16 B), so the bank-conflict
// swizzles of the "kc" / "st" images (device_common.h) are applied to the per-lane SOURCE
// address while the LDS destination stays linear; fragment reads use the same swizzle
// (cdna_hip_programming.md §5.4 rule 21).
//
//   waves: 2 (M) x 4 (N); wave tile 128 x BN/4 = 8 x (BN/64) fragments of 16x16x32
//   LDS:   2 buffers x (A 32 KiB + B BN*128 B)  = 128 KiB (BN=256) / 96 KiB (BN=128)
#include "common_host.h"
#include "gemm_core.h"

namespace sfcvit {
namespace {

using namespace gemm_core;

constexpr int T256 = 512;                 // threads
constexpr int A_BYTES = 256 * 64 * 2;     // 32 KiB

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// Issue the LDS-DMA of one operand tile: ROWS x 64 (k-contiguous, "kc" image) or
// 64 x ROWS (k-major; ROWS/128 "st" sub-images of 64 x 128).  ROWS * 8 16-byte slots,
// slot p = i * 512 + tid lands at LDS byte p * 16.
template <bool KMAJOR, int ROWS>
__device__ __forceinline__ void dma_tile(char *img, const uint16_t *__restrict__ src, int ld, int row0, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / T256; i++) {
        const int p = i * T256 + tid;
        const uint16_t *g;
        if (!KMAJOR) {
            const int row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
            g = src + size_t(row0 + row) * ld + k0 + (c << 3);
        } else {
            const int half = p >> 10, krow = (p >> 4) & 63, c16 = p & 15;
            const int c32 = (c16 >> 1) ^ ((krow & 3) | (((krow >> 3) & 1) << 2));
            g = src + size_t(k0 + krow) * ld + row0 + half * 128 + (((c32 << 1) | (c16 & 1)) << 3);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(img + p * 16), 16, 0, 0);
    }
}

template <bool KMAJOR>
__device__ __forceinline__ bf16x8 frag256(const char *img, int row0, int kk, int lane) {
    if (!KMAJOR) return kc_frag(img, row0, kk, lane);
    else return st_frag(img + (row0 >> 7) * (64 * 128 * 2), row0 & 127, kk, lane);
}

template <bool A_KM, bool B_KM, int BN_>
__global__ __launch_bounds__(T256, 2) void gemm256_kernel(const sfcvit_gemm_args g, int k_per_split) {
    constexpr int NF = BN_ / 64;                  // n-fragments per wave
    constexpr int B_BYTES = BN_ * 64 * 2;
    constexpr int BUF = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // ONE array: [2][A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_n = g.N / BN_;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * BN_;
    const uint16_t *A = static_cast<const uint16_t *>(g.a);
    const uint16_t *B = static_cast<const uint16_t *>(g.b);
    const int kbeg = blockIdx.z * k_per_split;
    const int kend = min(g.K, kbeg + k_per_split);
    const int nk = (kend - kbeg) / 64;

    f32x4 acc[8][NF];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < NF; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    dma_tile<A_KM, 256>(smem, A, g.lda, m0, kbeg, tid);
    dma_tile<B_KM, BN_>(smem + A_BYTES, B, g.ldb, n0, kbeg, tid);
    __syncthreads();   // LDS-DMA pending => hipcc drains vmcnt(0) here

    for (int kt = 0; kt < nk; kt++) {
        const char *ia = smem + (kt & 1) * BUF;
        const char *ib = ia + A_BYTES;
        if (kt + 1 < nk && g.force_generic != 3) {
            char *oa = smem + ((kt + 1) & 1) * BUF;
            dma_tile<A_KM, 256>(oa, A, g.lda, m0, kbeg + (kt + 1) * 64, tid);
            dma_tile<B_KM, BN_>(oa + A_BYTES, B, g.ldb, n0, kbeg + (kt + 1) * 64, tid);
        }
        if (g.force_generic != 2)
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            bf16x8 fb[NF];
#pragma unroll
            for (int j = 0; j < NF; j++) fb[j] = frag256<B_KM>(ib, wn * (BN_ / 4) + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const bf16x8 fa = frag256<A_KM>(ia, wm * 128 + i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < NF; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa, acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // acc[i][j][r] = C[m][n], m = m0 + wm*128 + i*16 + (lane&15), n = n0 + wn*BN/4 + j*16 + 4*(lane>>4) + r
    if (gridDim.z > 1) {
        float *slab = static_cast<float *>(g.workspace) + size_t(blockIdx.z) * g.M * g.N;
#pragma unroll
        for (int j = 0; j < NF; j++) {
            const int n = n0 + wn * (BN_ / 4) + j * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int m = m0 + wm * 128 + i * 16 + (lane & 15);
                *reinterpret_cast<f32x4 *>(slab + size_t(m) * g.N + n) = acc[i][j];
            }
        }
        return;
    }
    const uint16_t *bias = static_cast<const uint16_t *>(g.bias);
#pragma unroll
    for (int j = 0; j < NF; j++) {
        const int n = n0 + wn * (BN_ / 4) + j * 16 + 4 * (lane >> 4);
        float bv[4];
        load_bias4(bias, n, bv);
#pragma unroll
        for (int i = 0; i < 8; i++) epilogue4(g, m0 + wm * 128 + i * 16 + (lane & 15), n, acc[i][j], bv);
    }
}

template <bool A_KM, bool B_KM, int BN_>
int launch(const sfcvit_gemm_args &a, int splits, int k_per_split, hipStream_t s) {
    constexpr size_t lds = 2 * (A_BYTES + BN_ * 64 * 2);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256_kernel<A_KM, B_KM, BN_>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess)
            return check_launch("gemm256 attribute");
        attr_set = true;
    }
    dim3 grid((a.M / 256) * (a.N / BN_), 1, splits), block(T256);
    hipLaunchKernelGGL((gemm256_kernel<A_KM, B_KM, BN_>), grid, block, lds, s, a, k_per_split);
    return check_launch("gemm256");
}

}  // namespace

// Called by sfcvit_gemm after argument validation.  Returns -1 when the shape is not
// eligible (caller falls back to the generic kernel), else a status code.
int gemm256_dispatch(const sfcvit_gemm_args &a, int splits, int k_per_split, hipStream_t s) {
    if (a.M % 256 || a.N % 128 || a.K % 64 || k_per_split % 64) return -1;
    // Measured on ViT-B shapes (tools/bench_gemm.py): with one k-tile of prefetch this kernel wins
    // on the weight-gradient layout (both operands k-major, long K) and loses to the generic
    // kernel's two independent workgroups per CU elsewhere; force_generic == 4 forces it (tests).
    if (!(a.a_kmajor && a.b_kmajor) && a.force_generic != 4 && a.force_generic != 2 && a.force_generic != 3) return -1;
    // BN = 256 unless that leaves the last round of workgroups mostly idle on 256 CUs.
    bool bn256 = a.N % 256 == 0;
    if (bn256) {
        const long t = long(a.M / 256) * (a.N / 256) * splits;
        const long rounds = (t + 255) / 256;
        if (t < 200 || double(t) / double(rounds * 256) < 0.85) bn256 = false;
    }
#define SFCVIT_GO(AK, BK)                                                                          \
    return bn256 ? launch<AK, BK, 256>(a, splits, k_per_split, s) : launch<AK, BK, 128>(a, splits, k_per_split, s)
    if (!a.a_kmajor && !a.b_kmajor) SFCVIT_GO(false, false);
    if (!a.a_kmajor && a.b_kmajor) SFCVIT_GO(false, true);
    if (a.a_kmajor && !a.b_kmajor) SFCVIT_GO(true, false);
    SFCVIT_GO(true, true);
#undef SFCVIT_GO
}

}  // namespace sfcvit
