// Shared tile machinery of the MFMA GEMM kernels (gemm.hip, patch_embed.hip).
//
// Workgroup = 256 threads = 4 waves (2 x 2), tile 128 x 128 x 64; each wave owns a
// 64 x 64 sub-tile as 4 x 4 accumulator fragments of v_mfma_f32_16x16x32_bf16.
// The MFMA is issued as D = Bfrag * Afrag, so the accumulator holds C^T:
//   acc[i][j][r] = C[m][n],  m = wm*64 + i*16 + (lane & 15),  n = wn*64 + j*16 + 4*(lane >> 4) + r
// i.e. a lane owns 4 consecutive n of one m (8-byte bf16 / 16-byte fp32 epilogue vectors).
#pragma once
#include "../../include/sfcvit.h"
#include "device_common.h"

namespace sfcvit {
namespace gemm_core {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int THREADS = 256;
constexpr int TILE_BYTES = 128 * 64 * 2;  // one operand tile

struct Stage {
    u32x4 v[4];
};

// Global -> registers for one operand tile.  KMAJOR = false: memory is [rows][ld], k
// contiguous; tile = 128 rows x 64 k.  KMAJOR = true: memory is [K][ld], rows
// contiguous; tile = 64 k x 128 rows.  Out-of-range vectors are zero.
template <bool KMAJOR>
__device__ __forceinline__ void load_tile(Stage &s, const uint16_t *__restrict__ p, int ld, int row0, int nrows,
                                          int k0, int K, int tid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = tid + THREADS * i;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (!KMAJOR) {
            const int r = row0 + (v >> 3), k = k0 + ((v & 7) << 3);
            if (r < nrows && k < K) val = *reinterpret_cast<const u32x4 *>(p + size_t(r) * ld + k);
        } else {
            const int k = k0 + (v >> 4), r = row0 + ((v & 15) << 3);
            if (k < K && r < nrows) val = *reinterpret_cast<const u32x4 *>(p + size_t(k) * ld + r);
        }
        s.v[i] = val;
    }
}

template <bool KMAJOR>
__device__ __forceinline__ void store_tile(const Stage &s, char *img, int tid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = tid + THREADS * i;
        int off;
        if (!KMAJOR) off = kc_off(v >> 3, v & 7);
        else off = st_off(v >> 4, (v & 15) << 3);
        *reinterpret_cast<u32x4 *>(img + off) = s.v[i];
    }
}

template <bool KMAJOR>
__device__ __forceinline__ bf16x8 frag(const char *img, int row0, int kk, int lane) {
    if (!KMAJOR) return kc_frag(img, row0, kk, lane);
    else return st_frag(img, row0, kk, lane);
}


// One 64-deep k-tile of MFMAs from LDS images ia (A tile) and ib (B tile).
template <bool A_KM, bool B_KM>
__device__ __forceinline__ void mma_tile(f32x4 (&acc)[4][4], const char *ia, const char *ib, int wm, int wn, int lane) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) fa[i] = frag<A_KM>(ia, wm * 64 + i * 16, kk, lane);
#pragma unroll
        for (int j = 0; j < 4; j++) fb[j] = frag<B_KM>(ib, wn * 64 + j * 16, kk, lane);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
}

// XCD-aware tile order (cdna_hip_programming.md T1).  Workgroups are dealt round-robin over the
// 8 XCDs, each with a private L2: block b and b + 8 share one.  The linear tile order (n fastest,
// so neighbours share the A panel and every tile of a chunk shares the B panel) is cut into 8
// contiguous chunks, one per XCD; bijective for any tile count.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void zero_acc(f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Split-K partial: plain 16-byte stores of the fp32 tile into slab z of the workspace
// ([splits][M][N] fp32, N % 4 == 0).
__device__ __forceinline__ void store_partial(const f32x4 (&acc)[4][4], float *slab, int M, int N, int m0, int n0,
                                              int wm, int wn, int lane) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
        if (n >= N) continue;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = m0 + wm * 64 + i * 16 + (lane & 15);
            if (m < M) *reinterpret_cast<f32x4 *>(slab + size_t(m) * N + n) = acc[i][j];
        }
    }
}

// Bias of 4 consecutive columns as fp32.
__device__ __forceinline__ void load_bias4(const uint16_t *bias, int n, float (&bv)[4]) {
    bv[0] = bv[1] = bv[2] = bv[3] = 0.f;
    if (bias) {
        const u32x2 b2 = *reinterpret_cast<const u32x2 *>(bias + n);
        bv[0] = bf2f(uint16_t(b2[0])); bv[1] = bf2f(uint16_t(b2[0] >> 16));
        bv[2] = bf2f(uint16_t(b2[1])); bv[3] = bf2f(uint16_t(b2[1] >> 16));
    }
}

// Fused epilogue of one 4-vector C[m][n..n+3] (order documented at sfcvit_gemm in include/sfcvit.h).
__device__ __forceinline__ void epilogue4(const sfcvit_gemm_args &g, int m, int n, const f32x4 &acc, const float (&bv)[4]) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = acc[r] + bv[r];
    if (g.aux_out) {
        u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *reinterpret_cast<u32x2 *>(static_cast<uint16_t *>(g.aux_out) + size_t(m) * g.ldaux + n) = o;
    }
    if (g.act == SFCVIT_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
    } else if (g.act == SFCVIT_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = gelu_erf(v[r]);
    }
    if (g.dropout_p > 0.f) {
        const uint32_t th = drop_thresh(g.dropout_p);
        const float sc = 1.f / (1.f - g.dropout_p);
        const uint64_t pair = uint64_t(m) * uint64_t((g.N + 1) >> 1) + uint64_t(n >> 1);
        bool k[4];
        drop_keep2(g.dropout_seed, pair, th, k[0], k[1]);
        drop_keep2(g.dropout_seed, pair + 1, th, k[2], k[3]);
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = k[r] ? v[r] * sc : 0.f;
    }
    if (g.residual) {
        const u32x2 r2 = *reinterpret_cast<const u32x2 *>(static_cast<const uint16_t *>(g.residual) + size_t(m) * g.ldr + n);
        v[0] += bf2f(uint16_t(r2[0])); v[1] += bf2f(uint16_t(r2[0] >> 16));
        v[2] += bf2f(uint16_t(r2[1])); v[3] += bf2f(uint16_t(r2[1] >> 16));
    }
    if (g.dact != SFCVIT_ACT_NONE) {
        const u32x2 a2 = *reinterpret_cast<const u32x2 *>(static_cast<const uint16_t *>(g.aux_in) + size_t(m) * g.ldaux + n);
        const float a[4] = {bf2f(uint16_t(a2[0])), bf2f(uint16_t(a2[0] >> 16)), bf2f(uint16_t(a2[1])), bf2f(uint16_t(a2[1] >> 16))};
        const float ds = g.dact_scale != 0.f ? g.dact_scale : 1.f;
#pragma unroll
        for (int r = 0; r < 4; r++)
            v[r] = (g.dact == SFCVIT_ACT_RELU) ? (a[r] > 0.f ? v[r] * ds : 0.f) : v[r] * gelu_erf_grad(a[r]) * ds;
    }
    if (g.c_is_f32) {
        *reinterpret_cast<f32x4 *>(static_cast<float *>(g.c) + size_t(m) * g.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
        u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *reinterpret_cast<u32x2 *>(static_cast<uint16_t *>(g.c) + size_t(m) * g.ldc + n) = o;
    }
}

}  // namespace gemm_core
}  // namespace sfcvit
