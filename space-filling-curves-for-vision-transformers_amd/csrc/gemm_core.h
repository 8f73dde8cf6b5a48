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

__device__ __forceinline__ void unpack8f(const u32x4 &v, float *f) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        f[2 * i] = bf2f(uint16_t(v[i]));
        f[2 * i + 1] = bf2f(uint16_t(v[i] >> 16));
    }
}

// Fused epilogue of NV (4 or 8) consecutive columns C[m][n .. n+NV-1], in the order documented at
// sfcvit_gemm (include/sfcvit.h): bias, aux_out, act, dropout, residual, dact, store.
// HEAVY = the launch uses the erf GELU (forward or gradient); everything else compiles without it.
template <int NV, bool HEAVY>
__device__ __forceinline__ void epilogue_vec(const sfcvit_gemm_args &g, int m, int n, float (&v)[8], const float (&bv)[8]) {
#pragma unroll
    for (int r = 0; r < NV; r++) v[r] += bv[r];
    if (g.aux_out) {
        uint16_t *p = static_cast<uint16_t *>(g.aux_out) + size_t(m) * g.ldaux + n;
        if (NV == 8) *reinterpret_cast<u32x4 *>(p) = u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
        else *reinterpret_cast<u32x2 *>(p) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    }
    if (g.act == SFCVIT_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < NV; r++) v[r] = fmaxf(v[r], 0.f);
    } else if (HEAVY && g.act == SFCVIT_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < NV; r++) v[r] = gelu_erf(v[r]);
    }
    if (g.dropout_p > 0.f) {
        const uint32_t th = drop_thresh(g.dropout_p);
        const float sc = 1.f / (1.f - g.dropout_p);
        const uint32_t rk = drop_row_key(eff_seed(g.dropout_seed, g.seed_off), uint64_t(m) + uint64_t(uint32_t(g.row_offset)));
#pragma unroll
        for (int q = 0; q < NV / 2; q++) {
            bool k0, k1;
            drop_keep2(rk, uint32_t(n >> 1) + q, th, k0, k1);
            v[2 * q] = k0 ? v[2 * q] * sc : 0.f;
            v[2 * q + 1] = k1 ? v[2 * q + 1] * sc : 0.f;
        }
    }
    if (g.residual) {
        const uint16_t *p = static_cast<const uint16_t *>(g.residual) + size_t(m) * g.ldr + n;
        float rv[8];
        if (NV == 8) {
            unpack8f(*reinterpret_cast<const u32x4 *>(p), rv);
        } else {
            const u32x2 r2 = *reinterpret_cast<const u32x2 *>(p);
            rv[0] = bf2f(uint16_t(r2[0])); rv[1] = bf2f(uint16_t(r2[0] >> 16));
            rv[2] = bf2f(uint16_t(r2[1])); rv[3] = bf2f(uint16_t(r2[1] >> 16));
        }
#pragma unroll
        for (int r = 0; r < NV; r++) v[r] += rv[r];
    }
    if (g.dact != SFCVIT_ACT_NONE) {
        const uint16_t *p = static_cast<const uint16_t *>(g.aux_in) + size_t(m) * g.ldaux + n;
        float a[8];
        if (NV == 8) {
            unpack8f(*reinterpret_cast<const u32x4 *>(p), a);
        } else {
            const u32x2 a2 = *reinterpret_cast<const u32x2 *>(p);
            a[0] = bf2f(uint16_t(a2[0])); a[1] = bf2f(uint16_t(a2[0] >> 16));
            a[2] = bf2f(uint16_t(a2[1])); a[3] = bf2f(uint16_t(a2[1] >> 16));
        }
        const float ds = g.dact_scale != 0.f ? g.dact_scale : 1.f;
        if (!HEAVY || g.dact == SFCVIT_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < NV; r++) v[r] = a[r] > 0.f ? v[r] * ds : 0.f;
        } else {
#pragma unroll
            for (int r = 0; r < NV; r++) v[r] = v[r] * gelu_erf_grad(a[r]) * ds;
        }
    }
    if (g.c_is_f32) {
        float *c = static_cast<float *>(g.c) + size_t(m) * g.ldc + n;
        *reinterpret_cast<f32x4 *>(c) = f32x4{v[0], v[1], v[2], v[3]};
        if (NV == 8) *reinterpret_cast<f32x4 *>(c + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
        uint16_t *c = static_cast<uint16_t *>(g.c) + size_t(m) * g.ldc + n;
        if (NV == 8) *reinterpret_cast<u32x4 *>(c) = u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
        else *reinterpret_cast<u32x2 *>(c) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    }
}

// Epilogue of one wave's (16*NI) x (16*NF) accumulator tile through LDS.
//
// The MFMA leaves a lane with 4 consecutive n of one m, so a direct store moves 8 bytes per lane
// into 16 different rows per instruction.  Here the wave transposes 32 rows at a time through a
// private fp32 LDS patch ([32][16*NF + 4], padded against bank conflicts) and then owns 8
// consecutive n per lane: bias / residual / aux reads and the C store are 16-byte vectors, 8
// lanes per 128-byte row segment.
// `patch` = this wave's LDS region of 32 * (16*NF + 4) * 4 bytes; the operand tiles must be dead
// (call after the k-loop's final barrier).  M, N bounds are honoured (N % 4 == 0).
// Code size matters here: the fused epilogue has a wide tree of wave-uniform options, and fully
// unrolling it per vector made the kernels ~230 KB of code that thrashed the instruction cache
// (the epilogue alone took 140 us on the QKV shape).  So the read/compute/store loop of a pass is
// a real loop (unroll 1), and the erf-GELU variants are compiled only into HEAVY instantiations.
template <int NI, int NF, bool HEAVY>
__device__ __forceinline__ void epilogue_tile(const sfcvit_gemm_args &g, const f32x4 (&acc)[NI][NF], float *patch,
                                              int m_base, int n_base, int lane) {
    constexpr int WC = 16 * NF, LD = WC + 4, VPR = WC / 8, VPL = 32 * VPR / 64;
    const int cg = lane % VPR;
    const int n = n_base + 8 * cg;
    const int nvalid = g.N - n;                      // >= 8: full vector, 4: half, <= 0: nothing
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (g.bias && nvalid >= 4) {
        const uint16_t *bp = static_cast<const uint16_t *>(g.bias) + n;
        const u32x2 b0 = *reinterpret_cast<const u32x2 *>(bp);
        bv[0] = bf2f(uint16_t(b0[0])); bv[1] = bf2f(uint16_t(b0[0] >> 16));
        bv[2] = bf2f(uint16_t(b0[1])); bv[3] = bf2f(uint16_t(b0[1] >> 16));
        if (nvalid >= 8) {
            const u32x2 b1 = *reinterpret_cast<const u32x2 *>(bp + 4);
            bv[4] = bf2f(uint16_t(b1[0])); bv[5] = bf2f(uint16_t(b1[0] >> 16));
            bv[6] = bf2f(uint16_t(b1[1])); bv[7] = bf2f(uint16_t(b1[1] >> 16));
        }
    }
#pragma unroll
    for (int p = 0; p < NI / 2; p++) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++)
#pragma unroll
            for (int j = 0; j < NF; j++)
                *reinterpret_cast<f32x4 *>(patch + (16 * ii + (lane & 15)) * LD + 16 * j + 4 * (lane >> 4)) = acc[2 * p + ii][j];
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < VPL; k++) {
            const int row = (k * 64 + lane) / VPR;
            const int m = m_base + 32 * p + row;
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(patch + row * LD + 8 * cg);
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(patch + row * LD + 8 * cg + 4);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (m < g.M) {
                if (nvalid >= 8) epilogue_vec<8, HEAVY>(g, m, n, v, bv);
                else if (nvalid >= 4) epilogue_vec<4, HEAVY>(g, m, n, v, bv);
            }
        }
        __syncthreads();
    }
}

}  // namespace gemm_core
}  // namespace sfcvit
