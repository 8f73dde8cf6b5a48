// HBM-bound row-wise kernels for gfx950: LayerNorm forward/backward, column sums
// (bias gradients), GELU, soft-target cross-entropy, sum of squares and the fused
// clip + AdamW step.  One 64-lane wave owns one row; every global access is a
// 16-byte vector (8 bf16 or 4 fp32) and reductions are wave butterflies.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <vector>
#include "common_host.h"
#include "device_common.h"

namespace sfcvit {
namespace {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;

__device__ __forceinline__ void unpack8(const u32x4 &v, float *f) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        f[2 * i] = bf2f(uint16_t(v[i]));
        f[2 * i + 1] = bf2f(uint16_t(v[i] >> 16));
    }
}
__device__ __forceinline__ u32x4 pack8(const float *f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
    return v;
}

// ---------------------------------------------------------------------------
// LayerNorm forward.  VPL = 16-byte vectors per lane (D <= VPL * 512).
// ---------------------------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(THREADS) void ln_fwd_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ gamma,
                                                         const uint16_t *__restrict__ beta, uint16_t *__restrict__ y,
                                                         float *__restrict__ mean, float *__restrict__ rstd, int M,
                                                         int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nvec = D >> 3;
    float v[VPL][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            unpack8(*reinterpret_cast<const u32x4 *>(x + size_t(row) * D + c * 8), v[i]);
#pragma unroll
            for (int j = 0; j < 8; j++) s += v[i][j];
        }
    }
    const float mu = wave_sum(s) / float(D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++)
        if (lane + 64 * i < nvec) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float d = v[i][j] - mu;
                q += d * d;
            }
        }
    const float rs = rsqrtf(wave_sum(q) / float(D) + eps);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            float g[8], b[8], o[8];
            unpack8(*reinterpret_cast<const u32x4 *>(gamma + c * 8), g);
            unpack8(*reinterpret_cast<const u32x4 *>(beta + c * 8), b);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
            *reinterpret_cast<u32x4 *>(y + size_t(row) * D + c * 8) = pack8(o);
        }
    }
}

// Two rows per wave (D = 768: 2 x 96 vectors = 3 per lane exactly, where one row leaves a third of the lanes idle on the second
// load; D = 1 024: 4 per lane): twice the bytes in flight per wave -- at one 1.5-KB row per wave a CU holds 48 KB of loads, about
// what 6 TB/s needs at this latency, and the kernel ran at 4.6 -- and the two rows' reductions overlap.  In the step (rocprofv3,
// same box, alternating): 33.4 -> 30.8-31.2 us per launch; alone both forms take 26.9 us (inputs warm in the memory-side cache).
// Nontemporal loads of x on top: 31.4-31.5 us, not kept.
template <int VPL>
__global__ __launch_bounds__(THREADS) void ln_fwd2_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ gamma,
                                                          const uint16_t *__restrict__ beta, uint16_t *__restrict__ y,
                                                          float *__restrict__ mean, float *__restrict__ rstd, int M,
                                                          int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = 2 * (blockIdx.x * WAVES + (threadIdx.x >> 6));
    if (row0 >= M) return;
    const int nvec = D >> 3;                                   // 2 * nvec <= 64 * VPL (host)
    const bool two = row0 + 1 < M;
    float v[VPL][8];
    int col[VPL];                                              // vector column, or -1; bit 30: second row
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int c = lane + 64 * i, r = c >= nvec ? 1 : 0, cc = c - r * nvec;
        col[i] = (c < 2 * nvec && (r == 0 || two)) ? (cc | (r << 30)) : -1;
        if (col[i] >= 0) {
            unpack8(*reinterpret_cast<const u32x4 *>(x + size_t(row0 + r) * D + cc * 8), v[i]);
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) t += v[i][j];
            if (r) s1 += t; else s0 += t;
        }
    }
    const float mu0 = wave_sum(s0) / float(D), mu1 = wave_sum(s1) / float(D);
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++)
        if (col[i] >= 0) {
            const bool r = col[i] >> 30;
            const float mu = r ? mu1 : mu0;
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float d = v[i][j] - mu;
                t += d * d;
            }
            if (r) q1 += t; else q0 += t;
        }
    const float rs0 = rsqrtf(wave_sum(q0) / float(D) + eps), rs1 = rsqrtf(wave_sum(q1) / float(D) + eps);
    if (lane == 0) {
        mean[row0] = mu0;
        rstd[row0] = rs0;
        if (two) {
            mean[row0 + 1] = mu1;
            rstd[row0 + 1] = rs1;
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; i++)
        if (col[i] >= 0) {
            const int r = col[i] >> 30, cc = col[i] & 0xFFFFFF;
            const float mu = r ? mu1 : mu0, rs = r ? rs1 : rs0;
            float g[8], b[8], o[8];
            unpack8(*reinterpret_cast<const u32x4 *>(gamma + cc * 8), g);
            unpack8(*reinterpret_cast<const u32x4 *>(beta + cc * 8), b);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
            *reinterpret_cast<u32x4 *>(y + size_t(row0 + r) * D + cc * 8) = pack8(o);
        }
}

// ---------------------------------------------------------------------------
// LayerNorm backward.  Each wave walks rows with a grid stride, keeps its lanes'
// columns of dgamma / dbeta in registers, and the block writes one partial row
// [2][D] to the workspace; ln_bwd_reduce sums the partials.
// ---------------------------------------------------------------------------
template <int VPL, bool ADD>
__global__ __launch_bounds__(THREADS) void ln_bwd_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x,
                                                         const float *__restrict__ mean, const float *__restrict__ rstd,
                                                         const uint16_t *__restrict__ gamma,
                                                         const uint16_t *__restrict__ dx_add, uint16_t *__restrict__ dx,
                                                         uint16_t *__restrict__ dx_drop, float drop_p, uint32_t drop_seed_arg,
                                                         const uint32_t *__restrict__ seed_off,
                                                         float *__restrict__ partial, int M, int D) {
    const uint32_t drop_seed = eff_seed(drop_seed_arg, seed_off);
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [WAVES][3][D] fp32
    const uint32_t drop_th = drop_thresh(drop_p);
    const float drop_sc = 1.f / (1.f - drop_p);
    float *red = reinterpret_cast<float *>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = D >> 3;
    float g[VPL][8], dg[VPL][8], db[VPL][8], dc[VPL][8];   // dc: column sums of the outgoing gradient
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int c = lane + 64 * i;
        if (c < nvec) unpack8(*reinterpret_cast<const u32x4 *>(gamma + c * 8), g[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) dg[i][j] = db[i][j] = dc[i][j] = 0.f;
    }
    // Rows are software-pipelined: the loads of this wave's next row are issued before the current row is reduced
    // (a row is load -> two wave reductions -> store, and only 8 waves per CU are resident: without the prefetch
    // the kernel sat at 2.7-3.3 TB/s).  Two rows ahead measured slower (54.6 vs 51.3 us at M 50 176, D 768).
    const int stride = gridDim.x * WAVES;
    u32x4 xr[VPL], dr[VPL], ar[VPL], xn[VPL], dn[VPL], an[VPL];
    float mu = 0.f, rs = 0.f, mun = 0.f, rsn = 0.f;
    auto load_row = [&](int row, u32x4 (&xq)[VPL], u32x4 (&dq)[VPL], u32x4 (&aq)[VPL], float &m, float &r) __attribute__((always_inline)) {
        m = mean[row];
        r = rstd[row];
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                xq[i] = *reinterpret_cast<const u32x4 *>(x + size_t(row) * D + c * 8);
                dq[i] = *reinterpret_cast<const u32x4 *>(dy + size_t(row) * D + c * 8);
                if (ADD) aq[i] = *reinterpret_cast<const u32x4 *>(dx_add + size_t(row) * D + c * 8);
            }
        }
    };
    int row = blockIdx.x * WAVES + wave;
    if (row < M) load_row(row, xr, dr, ar, mu, rs);
    for (; row < M; row += stride) {
        const bool more = row + stride < M;
        if (more) load_row(row + stride, xn, dn, an, mun, rsn);
        float xh[VPL][8], gy[VPL][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                float xv[8], dv[8];
                unpack8(xr[i], xv);
                unpack8(dr[i], dv);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    xh[i][j] = (xv[j] - mu) * rs;
                    gy[i][j] = dv[j] * g[i][j];
                    s1 += gy[i][j];
                    s2 += gy[i][j] * xh[i][j];
                    dg[i][j] += dv[j] * xh[i][j];
                    db[i][j] += dv[j];
                }
            }
        }
        const float c1 = wave_sum(s1) / float(D), c2 = wave_sum(s2) / float(D);
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] = rs * (gy[i][j] - c1 - xh[i][j] * c2);
                if (ADD) {
                    float a[8];
                    unpack8(ar[i], a);
#pragma unroll
                    for (int j = 0; j < 8; j++) o[j] += a[j];
                }
                *reinterpret_cast<u32x4 *>(dx + size_t(row) * D + c * 8) = pack8(o);
                if (!dx_drop) {
#pragma unroll
                    for (int j = 0; j < 8; j++) dc[i][j] += o[j];
                }
                if (dx_drop) {
                    const uint32_t rk = drop_row_key(drop_seed, uint64_t(row));
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        bool k0, k1;
                        drop_keep2(rk, uint32_t(c * 4 + q), drop_th, k0, k1);
                        o[2 * q] = k0 ? o[2 * q] * drop_sc : 0.f;
                        o[2 * q + 1] = k1 ? o[2 * q + 1] * drop_sc : 0.f;
                    }
                    *reinterpret_cast<u32x4 *>(dx_drop + size_t(row) * D + c * 8) = pack8(o);
#pragma unroll
                    for (int j = 0; j < 8; j++) dc[i][j] += o[j];
                }
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < VPL; i++) { xr[i] = xn[i]; dr[i] = dn[i]; ar[i] = an[i]; }
            mu = mun;
            rs = rsn;
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int c = lane + 64 * i;
        if (c < nvec) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                red[(wave * 3 + 0) * D + c * 8 + j] = dg[i][j];
                red[(wave * 3 + 1) * D + c * 8 + j] = db[i][j];
                red[(wave * 3 + 2) * D + c * 8 + j] = dc[i][j];
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * D; c += THREADS) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; w++) s += red[w * 3 * D + c];
        partial[size_t(blockIdx.x) * 3 * D + c] = s;
    }
}

// The same for D = 256 CH with CH odd (ViT-B: 768 = 3 x 256), where the 16-byte-vector form above leaves a quarter of the
// lanes idle in its second vector.  Measured (tools/bench_rowwise.py, M 50 176): that kernel issues ~450 VALU instructions
// per row and wave -- 49 rows per SIMD x 450 x 4 clocks = 46 us of its 51-66 us: it is VALU-bound, not HBM-bound (two rows
// of prefetch, or three waves per SIMD from a register diet, made it slower).  Here a lane owns CH chunks of 4 columns
// (8-byte loads, every lane busy) and all arithmetic is written on float pairs (v_pk_* instructions, no shuffles).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 unpack2(uint32_t w) { return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }

template <int CH, bool ADD>
__global__ __launch_bounds__(THREADS) void ln_bwd_cols_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x,
                                                              const float *__restrict__ mean, const float *__restrict__ rstd,
                                                              const uint16_t *__restrict__ gamma,
                                                              const uint16_t *__restrict__ dx_add, uint16_t *__restrict__ dx,
                                                              uint16_t *__restrict__ dx_drop, float drop_p, uint32_t drop_seed_arg,
                                                              const uint32_t *__restrict__ seed_off,
                                                              float *__restrict__ partial, int M) {
    constexpr int D = 256 * CH;
    const uint32_t drop_seed = eff_seed(drop_seed_arg, seed_off);
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [WAVES][3][D] fp32
    const uint32_t drop_th = drop_thresh(drop_p);
    const float drop_sc = 1.f / (1.f - drop_p);
    float *red = reinterpret_cast<float *>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane's columns: 256 k + 4 lane + 2 h + {0, 1}, k < CH, h < 2
    f32x2 g[CH][2], dg[CH][2], db[CH][2], dc[CH][2];
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const u32x2 w = *reinterpret_cast<const u32x2 *>(gamma + 256 * k + 4 * lane);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            g[k][h] = unpack2(w[h]);
            dg[k][h] = db[k][h] = dc[k][h] = f32x2{0.f, 0.f};
        }
    }
    const int stride = gridDim.x * WAVES;
    u32x2 xr[CH], dr[CH], ar[CH], xn[CH], dn[CH], an[CH];
    float mu = 0.f, rs = 0.f, mun = 0.f, rsn = 0.f;
    auto load_row = [&](int row, u32x2 (&xq)[CH], u32x2 (&dq)[CH], u32x2 (&aq)[CH], float &m, float &r) __attribute__((always_inline)) {
        m = mean[row];
        r = rstd[row];
        const size_t o = size_t(row) * D + 4 * lane;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            xq[k] = *reinterpret_cast<const u32x2 *>(x + o + 256 * k);
            dq[k] = *reinterpret_cast<const u32x2 *>(dy + o + 256 * k);
            if (ADD) aq[k] = *reinterpret_cast<const u32x2 *>(dx_add + o + 256 * k);
        }
    };
    int row = blockIdx.x * WAVES + wave;
    if (row < M) load_row(row, xr, dr, ar, mu, rs);
    for (; row < M; row += stride) {
        const bool more = row + stride < M;
        if (more) load_row(row + stride, xn, dn, an, mun, rsn);   // the next row's loads fly while this one is reduced
        f32x2 xh[CH][2], gy[CH][2];
        f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < CH; k++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const f32x2 xv = unpack2(xr[k][h]), dv = unpack2(dr[k][h]);
                xh[k][h] = (xv - mu) * rs;
                gy[k][h] = dv * g[k][h];
                s1 += gy[k][h];
                s2 += gy[k][h] * xh[k][h];
                dg[k][h] += dv * xh[k][h];
                db[k][h] += dv;
            }
        }
        const float c1 = wave_sum(s1[0] + s1[1]) / float(D), c2 = wave_sum(s2[0] + s2[1]) / float(D);
        const size_t o = size_t(row) * D + 4 * lane;
        const uint32_t rk = drop_row_key(drop_seed, uint64_t(row));
#pragma unroll
        for (int k = 0; k < CH; k++) {
            f32x2 ov[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                ov[h] = rs * (gy[k][h] - c1 - xh[k][h] * c2);
                if (ADD) ov[h] += unpack2(ar[k][h]);
            }
            *reinterpret_cast<u32x2 *>(dx + o + 256 * k) = u32x2{pack2bf(ov[0][0], ov[0][1]), pack2bf(ov[1][0], ov[1][1])};
            if (dx_drop) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    bool k0, k1;
                    drop_keep2(rk, uint32_t(128 * k + 2 * lane + h), drop_th, k0, k1);      // pair = column / 2
                    ov[h][0] = k0 ? ov[h][0] * drop_sc : 0.f;
                    ov[h][1] = k1 ? ov[h][1] * drop_sc : 0.f;
                }
                *reinterpret_cast<u32x2 *>(dx_drop + o + 256 * k) = u32x2{pack2bf(ov[0][0], ov[0][1]), pack2bf(ov[1][0], ov[1][1])};
            }
            dc[k][0] += ov[0];
            dc[k][1] += ov[1];
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < CH; k++) { xr[k] = xn[k]; dr[k] = dn[k]; ar[k] = an[k]; }
            mu = mun;
            rs = rsn;
        }
    }
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int c = 256 * k + 4 * lane;
        *reinterpret_cast<f32x4 *>(red + (wave * 3 + 0) * D + c) = f32x4{dg[k][0][0], dg[k][0][1], dg[k][1][0], dg[k][1][1]};
        *reinterpret_cast<f32x4 *>(red + (wave * 3 + 1) * D + c) = f32x4{db[k][0][0], db[k][0][1], db[k][1][0], db[k][1][1]};
        *reinterpret_cast<f32x4 *>(red + (wave * 3 + 2) * D + c) = f32x4{dc[k][0][0], dc[k][0][1], dc[k][1][0], dc[k][1][1]};
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * D; c += THREADS) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; w++) s += red[w * 3 * D + c];
        partial[size_t(blockIdx.x) * 3 * D + c] = s;
    }
}

// Sum of per-block partials, shared by ln_bwd_reduce and reduce_batched_kernel: a 1024-thread block owns 16 columns;
// its 64 thread rows each add every 64th partial row (<= 8 independent loads in flight per thread for 512 partials,
// where the 16-row form these kernels had serialised 32), then 4 thread rows add 16 sub-sums each and one adds those
// 4.  Fixed summation order, no float atomics: bit-reproducible.  The value is returned in thread row 0.
__device__ __forceinline__ void store_grad(void *p, int i, float v, int as_bf16) {
    if (as_bf16) static_cast<uint16_t *>(p)[i] = f2bf(v);
    else static_cast<float *>(p)[i] = v;
}

constexpr int RED_THREADS = 1024;

__device__ __forceinline__ float reduce_partials16(const float *__restrict__ part, int nparts, int ld, int c, bool ok) {
    __shared__ float red[64][17];
    __shared__ float red2[4][16];
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    float s = 0.f;
    if (ok) {
        int b = grp;
        for (; b + 192 < nparts; b += 256) {
            const float v0 = part[size_t(b) * ld + c], v1 = part[size_t(b + 64) * ld + c];
            const float v2 = part[size_t(b + 128) * ld + c], v3 = part[size_t(b + 192) * ld + c];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < nparts; b += 64) s += part[size_t(b) * ld + c];
    }
    red[grp][cl] = s;
    __syncthreads();
    if (grp < 4) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) t += red[grp * 16 + k][cl];
        red2[grp][cl] = t;
    }
    __syncthreads();
    return grp == 0 ? (red2[0][cl] + red2[1][cl]) + (red2[2][cl] + red2[3][cl]) : 0.f;
}

// [nblocks][3][D] partials -> dgamma | dbeta | column sums of the outgoing gradient.
__global__ __launch_bounds__(RED_THREADS) void ln_bwd_reduce(const float *__restrict__ partial, void *__restrict__ dgamma,
                                                            void *__restrict__ dbeta, void *__restrict__ dcol, int nblocks,
                                                            int D, int as_bf16) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const float t = reduce_partials16(partial, nblocks, 3 * D, c, c < 3 * D);
    if (threadIdx.x < 16 && c < 3 * D) {
        if (c < D) store_grad(dgamma, c, t, as_bf16);
        else if (c < 2 * D) store_grad(dbeta, c - D, t, as_bf16);
        else if (dcol) store_grad(dcol, c - 2 * D, t, as_bf16);
    }
}

// SFCVIT_LN_COLS=0: the 16-byte-vector kernel also at D = 768 / 1 024 (A/B in tools/bench_rowwise.py; at D = 1 024, ViT-L, the
// float-pair form gains 2-6 %: 45.3 -> 44.6 us, with dropout output 61.4 -> 57.9 us at M = 36 864)
bool ln_bwd_cols(int D) {
    static const bool on = [] { const char *e = getenv("SFCVIT_LN_COLS"); return !(e && e[0] == '0'); }();
    return on && (D == 768 || D == 1024);
}

// Workgroups of the backward kernel: two (16-byte-vector kernel, 190-222 registers) or three (column-chunk kernel, 145-156)
// resident waves per SIMD.  Measured at M 50 176, D 768: 52.1 / 56.4 us with 512 / 768 workgroups of the former, 47.5 / 45.8
// of the latter (with dropout output and column sums 70.3 / 77.2 and 59.0 / 57.3).
int ln_bwd_blocks(int M, int D) {
    const int want = (M + WAVES - 1) / WAVES;
    static const int env = [] { const char *e = getenv("SFCVIT_LN_BLOCKS"); return e ? atoi(e) : 0; }();   // tuning knob
    const int cap = env > 0 ? env : (ln_bwd_cols(D) && D == 768 ? 768 : 512);    // (D = 1 024: 179-194 registers, two waves per SIMD)
    return want < cap ? want : cap;
}

// ---------------------------------------------------------------------------
// Column sums of a bf16 [M, ld] matrix (bias gradients).
// Thread (cv, rl) owns 8 columns and every 8th row of the block's row range.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void colsum_kernel(const uint16_t *__restrict__ x, int M, int N, int ld,
                                                         int rows_per_block, float *__restrict__ out) {
    __shared__ float red[8][256];
    const int cv = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 256 + cv * 8;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        for (int r = r0 + rl; r < r1; r += 8) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4 *>(x + size_t(r) * ld + col), v);
#pragma unroll
            for (int j = 0; j < 8; j++) s[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) red[rl][cv * 8 + j] = s[j];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; r++) t += red[r][threadIdx.x];
        out[size_t(blockIdx.y) * N + c] = t;          // partial[row block][column]
    }
}

// Many such reductions in one launch (sfcvit_reduce_flush): a workgroup finds its item by the first-block table.
constexpr int RED_BATCH = 96;
struct ReduceItem {
    const float *part;
    void *out;
    int nparts, ld, ncols, out_bf16;
};
struct ReduceBatch {                 // by value in the kernel arguments: 96 x 32 + 98 x 4 bytes < 4 KiB
    ReduceItem it[RED_BATCH];
    int first[RED_BATCH + 1];        // first workgroup of item i; first[n] = grid size
    int n;
};
__global__ __launch_bounds__(RED_THREADS) void reduce_batched_kernel(const ReduceBatch b) {
    int lo = 0, hi = b.n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (b.first[mid] <= int(blockIdx.x)) lo = mid; else hi = mid;
    }
    const float *part = b.it[lo].part;
    void *out = b.it[lo].out;
    const int nparts = b.it[lo].nparts, ld = b.it[lo].ld, ncols = b.it[lo].ncols, as_bf16 = b.it[lo].out_bf16;
    const int c = (int(blockIdx.x) - b.first[lo]) * 16 + (threadIdx.x & 15);
    const float t = reduce_partials16(part, nparts, ld, c, c < ncols);
    if (threadIdx.x < 16 && c < ncols) store_grad(out, c, t, as_bf16);
}

// ---------------------------------------------------------------------------
// bf16 matrix transpose: dst[c][r] = src[r][c], 64 x 64 tiles through LDS (16-byte global
// accesses on both sides).  Used once per step per weight so that dX = dY W runs with both
// operands k-contiguous (the layout the LDS-DMA GEMM is fastest on).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void transpose_kernel(const uint16_t *__restrict__ src, int R, int C, int lds,
                                                            uint16_t *__restrict__ dst, int ldd) {
    __shared__ uint16_t tile[64][66];                       // 66: odd dword stride, conflict-free column reads
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; i++) {                           // 64 rows x 8 vectors
        const int v = t + THREADS * i, r = v >> 3, cv = (v & 7) * 8;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (r0 + r < R && c0 + cv < C) val = *reinterpret_cast<const u32x4 *>(src + size_t(r0 + r) * lds + c0 + cv);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            tile[r][cv + 2 * j] = uint16_t(val[j]);
            tile[r][cv + 2 * j + 1] = uint16_t(val[j] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int v = t + THREADS * i, c = v >> 3, rv = (v & 7) * 8;
        if (c0 + c < C && r0 + rv < R) {
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = uint32_t(tile[rv + 2 * j][c]) | (uint32_t(tile[rv + 2 * j + 1][c]) << 16);
            *reinterpret_cast<u32x4 *>(dst + size_t(c0 + c) * ldd + r0 + rv) = o;
        }
    }
}

// One launch for many matrices: a workgroup looks its 64 x 64 tile up in a device table (sfcvit_transpose_batched).
__global__ __launch_bounds__(THREADS) void transpose_batched_kernel(const uint16_t *__restrict__ src_base, uint16_t *__restrict__ dst_base,
                                                                    const sfcvit_transpose_tile *__restrict__ tiles) {
    __shared__ uint16_t tile[64][66];
    const sfcvit_transpose_tile d = tiles[blockIdx.x];
    const uint16_t *src = src_base + d.src_off;
    uint16_t *dst = dst_base + d.dst_off;
    const int R = d.R, C = d.C, r0 = d.r0, c0 = d.c0, t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int v = t + THREADS * i, r = v >> 3, cv = (v & 7) * 8;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (r0 + r < R && c0 + cv < C) val = *reinterpret_cast<const u32x4 *>(src + size_t(r0 + r) * C + c0 + cv);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            tile[r][cv + 2 * j] = uint16_t(val[j]);
            tile[r][cv + 2 * j + 1] = uint16_t(val[j] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int v = t + THREADS * i, c = v >> 3, rv = (v & 7) * 8;
        if (c0 + c < C && r0 + rv < R) {
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = uint32_t(tile[rv + 2 * j][c]) | (uint32_t(tile[rv + 2 * j + 1][c]) << 16);
            *reinterpret_cast<u32x4 *>(dst + size_t(c0 + c) * R + r0 + rv) = o;
        }
    }
}

// ---------------------------------------------------------------------------
// GELU (erf form)
// ---------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(THREADS) void gelu_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x,
                                                       uint16_t *__restrict__ out, int64_t nvec) {
    for (int64_t i = blockIdx.x * int64_t(THREADS) + threadIdx.x; i < nvec; i += int64_t(gridDim.x) * THREADS) {
        float xv[8], o[8];
        unpack8(*reinterpret_cast<const u32x4 *>(x + i * 8), xv);
        if (BWD) {
            float dv[8];
            unpack8(*reinterpret_cast<const u32x4 *>(dy + i * 8), dv);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = dv[j] * gelu_erf_grad(xv[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = gelu_erf(xv[j]);
        }
        *reinterpret_cast<u32x4 *>(out + i * 8) = pack8(o);
    }
}

// GELU followed by dropout on a [rows, cols] tensor (cols % 8 == 0); one 8-vector per thread-iteration.
template <bool BWD>
__global__ __launch_bounds__(THREADS) void gelu_drop_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x,
                                                            uint16_t *__restrict__ out, int rows, int cols, float p,
                                                            uint32_t seed_arg, const uint32_t *__restrict__ seed_off) {
    const uint32_t seed = eff_seed(seed_arg, seed_off);
    const uint32_t th = drop_thresh(p);
    const float sc = 1.f / (1.f - p);
    const int vpr = cols >> 3;
    const int64_t nvec = int64_t(rows) * vpr;
    for (int64_t i = blockIdx.x * int64_t(THREADS) + threadIdx.x; i < nvec; i += int64_t(gridDim.x) * THREADS) {
        const int64_t row = i / vpr;
        const int cv = int(i - row * vpr);
        float xv[8], o[8];
        unpack8(*reinterpret_cast<const u32x4 *>(x + i * 8), xv);
        if (BWD) {
            float dv[8];
            unpack8(*reinterpret_cast<const u32x4 *>(dy + i * 8), dv);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = dv[j] * gelu_erf_grad(xv[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = gelu_erf(xv[j]);
        }
        const uint32_t rk = drop_row_key(seed, uint64_t(row));
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bool k0, k1;
            drop_keep2(rk, uint32_t(cv * 4 + q), th, k0, k1);
            o[2 * q] = k0 ? o[2 * q] * sc : 0.f;
            o[2 * q + 1] = k1 ? o[2 * q + 1] * sc : 0.f;
        }
        *reinterpret_cast<u32x4 *>(out + i * 8) = pack8(o);
    }
}

__global__ __launch_bounds__(THREADS) void dropout_mask_kernel(uint16_t *__restrict__ out, int64_t rows, int cols, float p,
                                                               uint32_t seed) {
    const uint32_t th = drop_thresh(p);
    const uint16_t on = f2bf(1.f / (1.f - p));
    const int ppr = (cols + 1) >> 1;
    const int64_t npair = rows * ppr;
    for (int64_t i = blockIdx.x * int64_t(THREADS) + threadIdx.x; i < npair; i += int64_t(gridDim.x) * THREADS) {
        const int64_t row = i / ppr;
        const int c = int(i - row * ppr) * 2;
        bool k0, k1;
        drop_keep2(drop_row_key(seed, uint64_t(row)), uint32_t(c >> 1), th, k0, k1);
        out[row * cols + c] = k0 ? on : uint16_t(0);
        if (c + 1 < cols) out[row * cols + c + 1] = k1 ? on : uint16_t(0);
    }
}

// ---------------------------------------------------------------------------
// Soft-target cross entropy: one wave per row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void soft_ce_kernel(const uint16_t *__restrict__ logits,
                                                          const float *__restrict__ targets,
                                                          float *__restrict__ loss_rows, uint16_t *__restrict__ dlogits,
                                                          int B, int C, int ld, float gscale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (row >= B) return;
    const uint16_t *l = logits + size_t(row) * ld;
    const float *t = targets + size_t(row) * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, bf2f(l[c]));
    mx = wave_max(mx);
    float se = 0.f, st = 0.f, stl = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float z = bf2f(l[c]);
        se += __expf(z - mx);
        st += t[c];
        stl += t[c] * z;
    }
    se = wave_sum(se);
    st = wave_sum(st);
    stl = wave_sum(stl);
    const float lse = mx + __logf(se);
    if (lane == 0) loss_rows[row] = lse * st - stl;   // -sum t*(z - lse)
    if (dlogits) {
        uint16_t *d = dlogits + size_t(row) * ld;
        for (int c = lane; c < ld; c += 64) {
            float gr = 0.f;
            if (c < C) gr = (__expf(bf2f(l[c]) - lse) * st - t[c]) * gscale;
            d[c] = f2bf(gr);
        }
    }
}

// ---------------------------------------------------------------------------
// Sum of squares (gradient norm) and fused clip + AdamW
// ---------------------------------------------------------------------------
template <bool F32>
__global__ __launch_bounds__(THREADS) void sumsq_kernel(const void *__restrict__ g, int64_t n, float *__restrict__ out) {
    __shared__ float red[WAVES];
    float s = 0.f;
    const int64_t nvec = n >> 3;
    for (int64_t i = blockIdx.x * int64_t(THREADS) + threadIdx.x; i < nvec; i += int64_t(gridDim.x) * THREADS) {
        float v[8];
        if (F32) {
            const f32x4 a = reinterpret_cast<const f32x4 *>(g)[2 * i], b = reinterpret_cast<const f32x4 *>(g)[2 * i + 1];
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
            unpack8(reinterpret_cast<const u32x4 *>(g)[i], v);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) s += v[j] * v[j];
    }
    if (blockIdx.x == 0 && threadIdx.x < int(n & 7)) {   // tail (< 8 elements)
        const int64_t i = (nvec << 3) + threadIdx.x;
        const float v = F32 ? static_cast<const float *>(g)[i] : bf2f(static_cast<const uint16_t *>(g)[i]);
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; w++) t += red[w];
        out[blockIdx.x] = t;                           // partial[block]
    }
}

// *out += sum of the block partials in a fixed order (lane-strided, then the wave butterfly)
__global__ __launch_bounds__(64) void sumsq_reduce_kernel(const float *__restrict__ part, int nparts, float *__restrict__ out) {
    float t = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 64) t += part[i];
    t = wave_sum(t);
    if (threadIdx.x == 0) *out += t;
}

// 8 elements per thread-iteration: bf16 param / grad as one 16-byte vector, fp32 master / m / v
// as two each.  n_vec = n / 8; the (< 8 element) tail is handled by the first threads.
__device__ __forceinline__ void adamw_one(float &w, float &m, float &v, float gr, const sfcvit_adamw_args &a, float bc1,
                                          float rbc2) {
    w *= 1.f - a.lr * a.weight_decay;
    m = a.beta1 * m + (1.f - a.beta1) * gr;
    v = a.beta2 * v + (1.f - a.beta2) * gr * gr;
    const float denom = sqrtf(v) * rbc2 + a.eps;
    w -= (a.lr / bc1) * (m / denom);
}

template <int NT>      // bit 0: non-temporal loads, bit 1: non-temporal stores of the fp32 state
__global__ __launch_bounds__(THREADS) void adamw_kernel(sfcvit_adamw_args a, float bc1, float rbc2) {
    if (a.dev_state) {            // learning rate and bias corrections of THIS step from the device (sfcvit_step_advance)
        a.lr = a.dev_state[2];
        bc1 = a.dev_state[3];
        rbc2 = a.dev_state[4];
    }
    float gmul = a.grad_scale;
    if (a.sumsq) {
        const float norm = sqrtf(*a.sumsq) * a.grad_scale;
        gmul *= fminf(1.f, a.max_norm / (norm + 1e-6f));
    }
    uint16_t *p = static_cast<uint16_t *>(a.param);
    const uint16_t *g = static_cast<const uint16_t *>(a.grad);
    // A lane owns groups of FOUR parameters: one 8-byte load of the bf16 gradient and one 16-byte load of each fp32 buffer,
    // so that every wave instruction covers one contiguous 512-byte / 1-KiB span (with eight parameters per lane the fp32
    // loads were two 16-byte halves at a 32-byte lane stride: half-used lines per instruction, 4.0 TB/s; tools/bench_adamw.py).
    // Two groups per iteration keep 2 x 56 bytes per lane in flight.
    const int64_t nq = a.n >> 2, step = int64_t(gridDim.x) * THREADS;
    auto load = [&](int64_t i, u32x2 &gq, f32x4 &wv, f32x4 &mv, f32x4 &vv) __attribute__((always_inline)) {
        if (NT & 1) {                                               // every byte is touched once per step
            gq = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(g) + i);
            wv = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.master) + i);
            mv = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.m) + i);
            vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.v) + i);
        } else {
            gq = reinterpret_cast<const u32x2 *>(g)[i];
            wv = reinterpret_cast<const f32x4 *>(a.master)[i];
            mv = reinterpret_cast<const f32x4 *>(a.m)[i];
            vv = reinterpret_cast<const f32x4 *>(a.v)[i];
        }
    };
    auto update = [&](int64_t i, const u32x2 &gq, f32x4 wv, f32x4 mv, f32x4 vv) __attribute__((always_inline)) {
        const float gr[4] = {bf2f(uint16_t(gq[0])), bf2f(uint16_t(gq[0] >> 16)), bf2f(uint16_t(gq[1])), bf2f(uint16_t(gq[1] >> 16))};
        float w[4], m[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            w[j] = wv[j];
            m[j] = mv[j];
            v[j] = vv[j];
            adamw_one(w[j], m[j], v[j], gr[j] * gmul, a, bc1, rbc2);
        }
        if (NT & 2) {
            __builtin_nontemporal_store(f32x4{w[0], w[1], w[2], w[3]}, reinterpret_cast<f32x4 *>(a.master) + i);
            __builtin_nontemporal_store(f32x4{m[0], m[1], m[2], m[3]}, reinterpret_cast<f32x4 *>(a.m) + i);
            __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4 *>(a.v) + i);
        } else {
            reinterpret_cast<f32x4 *>(a.master)[i] = f32x4{w[0], w[1], w[2], w[3]};
            reinterpret_cast<f32x4 *>(a.m)[i] = f32x4{m[0], m[1], m[2], m[3]};
            reinterpret_cast<f32x4 *>(a.v)[i] = f32x4{v[0], v[1], v[2], v[3]};
        }
        reinterpret_cast<u32x2 *>(p)[i] = u32x2{pack2bf(w[0], w[1]), pack2bf(w[2], w[3])};   // (the next forward reads these)
    };
    int64_t i = blockIdx.x * int64_t(THREADS) + threadIdx.x;
    for (; i + step < nq; i += 2 * step) {
        u32x2 g0, g1;
        f32x4 w0, m0, v0, w1, m1, v1;
        load(i, g0, w0, m0, v0);
        load(i + step, g1, w1, m1, v1);
        update(i, g0, w0, m0, v0);
        update(i + step, g1, w1, m1, v1);
    }
    if (i < nq) {
        u32x2 g0;
        f32x4 w0, m0, v0;
        load(i, g0, w0, m0, v0);
        update(i, g0, w0, m0, v0);
    }
    if (blockIdx.x == 0 && threadIdx.x < int(a.n & 3)) {
        const int64_t t = (nq << 2) + threadIdx.x;
        float w = a.master[t], m = a.m[t], v = a.v[t];
        adamw_one(w, m, v, bf2f(g[t]) * gmul, a, bc1, rbc2);
        a.master[t] = w;
        a.m[t] = m;
        a.v[t] = v;
        p[t] = f2bf(w);
    }
}

int grid_for(int64_t work_items) {
    int64_t b = (work_items + THREADS - 1) / THREADS;
    if (b < 1) b = 1;
    return int(b > 2048 ? 2048 : b);
}

}  // namespace
}  // namespace sfcvit

using namespace sfcvit;

extern "C" int sfcvit_layernorm_fwd(const void *x, const void *gamma, const void *beta, void *y, float *mean,
                                    float *rstd, int M, int D, float eps, void *stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return fail(SFCVIT_EINVAL, "layernorm_fwd: null pointer");
    if (M <= 0 || D <= 0 || D % 8 || D > 4096) return fail(SFCVIT_EINVAL, "layernorm_fwd: M=%d D=%d (D %% 8 == 0, D <= 4096)", M, D);
    if (!aligned16(x) || !aligned16(y) || !aligned16(gamma) || !aligned16(beta)) return fail(SFCVIT_EINVAL, "layernorm_fwd: alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((M + WAVES - 1) / WAVES), block(THREADS);
    const auto *xp = static_cast<const uint16_t *>(x);
    const auto *gp = static_cast<const uint16_t *>(gamma);
    const auto *bp = static_cast<const uint16_t *>(beta);
    auto *yp = static_cast<uint16_t *>(y);
    static const int two_rows = [] { const char *e = getenv("SFCVIT_LN_FWD_TWO_ROWS"); return e ? atoi(e) : 1; }();
    if (two_rows && (D == 768 || D == 1024) && M >= 4096) {
        const dim3 grid2((M + 2 * WAVES - 1) / (2 * WAVES));
        if (D == 768) hipLaunchKernelGGL(ln_fwd2_kernel<3>, grid2, block, 0, s, xp, gp, bp, yp, mean, rstd, M, D, eps);
        else hipLaunchKernelGGL(ln_fwd2_kernel<4>, grid2, block, 0, s, xp, gp, bp, yp, mean, rstd, M, D, eps);
        return check_launch("layernorm_fwd");
    }
    if (D <= 512) hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, block, 0, s, xp, gp, bp, yp, mean, rstd, M, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, block, 0, s, xp, gp, bp, yp, mean, rstd, M, D, eps);
    else if (D <= 2048) hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, block, 0, s, xp, gp, bp, yp, mean, rstd, M, D, eps);
    else hipLaunchKernelGGL(ln_fwd_kernel<8>, grid, block, 0, s, xp, gp, bp, yp, mean, rstd, M, D, eps);
    return check_launch("layernorm_fwd");
}

extern "C" int64_t sfcvit_layernorm_bwd_ws(int M, int D) {
    if (M <= 0 || D <= 0) return 0;
    return int64_t(ln_bwd_blocks(M, D)) * 3 * D * int64_t(sizeof(float));
}

extern "C" int sfcvit_layernorm_bwd_drop(const void *dy, const void *x, const float *mean, const float *rstd,
                                         const void *gamma, const void *dx_add, void *dx, void *dx_drop, float p,
                                         uint32_t seed, const uint32_t *seed_off, void *dgamma, void *dbeta, void *dcol,
                                         int grads_bf16, int M, int D, void *ws, void *stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !ws)
        return fail(SFCVIT_EINVAL, "layernorm_bwd: null pointer");
    if (M <= 0 || D <= 0 || D % 8 || D > 2048) return fail(SFCVIT_EINVAL, "layernorm_bwd: M=%d D=%d (D %% 8 == 0, D <= 2048)", M, D);
    if (!aligned16(dy) || !aligned16(x) || !aligned16(dx) || !aligned16(gamma) || (dx_add && !aligned16(dx_add)) ||
        (dx_drop && !aligned16(dx_drop)))
        return fail(SFCVIT_EINVAL, "layernorm_bwd: alignment");
    if (dx_drop && !(p >= 0.f && p < 1.f)) return fail(SFCVIT_EINVAL, "layernorm_bwd: dropout p=%g", p);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = ln_bwd_blocks(M, D);
    dim3 grid(nb), block(THREADS);
    const size_t lds = size_t(WAVES) * 3 * D * sizeof(float);
    const auto *dyp = static_cast<const uint16_t *>(dy);
    const auto *xp = static_cast<const uint16_t *>(x);
    const auto *gp = static_cast<const uint16_t *>(gamma);
    const auto *ap = static_cast<const uint16_t *>(dx_add);
    auto *dxp = static_cast<uint16_t *>(dx);
    auto *ddp = static_cast<uint16_t *>(dx_drop);
    float *part = static_cast<float *>(ws);
#define LN_BWD(VPL) do { if (ap) hipLaunchKernelGGL((ln_bwd_kernel<VPL, true>), grid, block, lds, s, dyp, xp, mean, rstd, gp, ap, dxp, ddp, p, seed, seed_off, part, M, D); \
                         else hipLaunchKernelGGL((ln_bwd_kernel<VPL, false>), grid, block, lds, s, dyp, xp, mean, rstd, gp, ap, dxp, ddp, p, seed, seed_off, part, M, D); } while (0)
    const char *tf = ap ? "true" : "false";
    if (ln_bwd_cols(D)) note_rowwise_kernel("ln_bwd_cols_kernel<%d, %s>", D == 768 ? 3 : 4, tf);
    else note_rowwise_kernel("ln_bwd_kernel<%d, %s>", D <= 512 ? 1 : D <= 1024 ? 2 : 4, tf);
    if (ln_bwd_cols(D) && D == 768) {
        if (ap) hipLaunchKernelGGL((ln_bwd_cols_kernel<3, true>), grid, block, lds, s, dyp, xp, mean, rstd, gp, ap, dxp, ddp, p, seed, seed_off, part, M);
        else hipLaunchKernelGGL((ln_bwd_cols_kernel<3, false>), grid, block, lds, s, dyp, xp, mean, rstd, gp, ap, dxp, ddp, p, seed, seed_off, part, M);
    } else if (ln_bwd_cols(D)) {
        if (ap) hipLaunchKernelGGL((ln_bwd_cols_kernel<4, true>), grid, block, lds, s, dyp, xp, mean, rstd, gp, ap, dxp, ddp, p, seed, seed_off, part, M);
        else hipLaunchKernelGGL((ln_bwd_cols_kernel<4, false>), grid, block, lds, s, dyp, xp, mean, rstd, gp, ap, dxp, ddp, p, seed, seed_off, part, M);
    } else if (D <= 512) LN_BWD(1);
    else if (D <= 1024) LN_BWD(2);
    else LN_BWD(4);
#undef LN_BWD
    if (int rc = check_launch("layernorm_bwd")) return rc;
    if (reduce_deferring()) {            // queued for the batched launch at sfcvit_reduce_flush: partial rows are [3][D]
        reduce_cols(part, nb, 3 * D, D, dgamma, grads_bf16, stream);
        reduce_cols(part + D, nb, 3 * D, D, dbeta, grads_bf16, stream);
        if (dcol) reduce_cols(part + 2 * D, nb, 3 * D, D, dcol, grads_bf16, stream);
        return SFCVIT_OK;
    }
    hipLaunchKernelGGL(ln_bwd_reduce, dim3((3 * D + 15) / 16), dim3(RED_THREADS), 0, s, part, dgamma, dbeta, dcol, nb, D, grads_bf16);
    return check_launch("layernorm_bwd_reduce");
}

extern "C" int sfcvit_layernorm_bwd(const void *dy, const void *x, const float *mean, const float *rstd,
                                    const void *gamma, const void *dx_add, void *dx, float *dgamma, float *dbeta,
                                    int M, int D, void *ws, void *stream) {
    return sfcvit_layernorm_bwd_drop(dy, x, mean, rstd, gamma, dx_add, dx, nullptr, 0.f, 0u, nullptr, dgamma, dbeta, nullptr, 0, M, D, ws, stream);
}

namespace {
void colsum_plan(int M, int N, int &col_blocks, int &row_blocks, int &rpb) {
    col_blocks = (N + 255) / 256;
    row_blocks = 1024 / col_blocks;          // ~1024 workgroups in the first pass; the partials stay a few MB
    if (row_blocks > 256) row_blocks = 256;
    if (row_blocks < 1) row_blocks = 1;
    rpb = (M + row_blocks - 1) / row_blocks;
    rpb = ((rpb + 7) / 8) * 8;
    row_blocks = (M + rpb - 1) / rpb;
}
}  // namespace

namespace sfcvit {
namespace {
std::atomic<int> g_defer{0};
std::mutex g_reduce_mutex;
std::vector<ReduceItem> g_reduce_queue;
}  // namespace

bool reduce_deferring() { return g_defer.load(std::memory_order_relaxed) != 0; }

int reduce_cols(const float *part, int nparts, int ld, int ncols, void *out, int out_bf16, void *stream) {
    if (reduce_deferring()) {
        std::lock_guard<std::mutex> lock(g_reduce_mutex);
        g_reduce_queue.push_back(ReduceItem{part, out, nparts, ld, ncols, out_bf16});
        return SFCVIT_OK;
    }
    ReduceBatch b;
    b.it[0] = ReduceItem{part, out, nparts, ld, ncols, out_bf16};
    b.first[0] = 0;
    b.first[1] = (ncols + 15) / 16;
    b.n = 1;
    hipLaunchKernelGGL(reduce_batched_kernel, dim3(b.first[1]), dim3(RED_THREADS), 0, static_cast<hipStream_t>(stream), b);
    return check_launch("column reduce");
}

int launch_colsum_reduce(const float *part, int nparts, int N, void *out, int out_bf16, void *stream) {
    return reduce_cols(part, nparts, N, N, out, out_bf16, stream);
}
}  // namespace sfcvit

extern "C" int sfcvit_reduce_defer(int on) { return g_defer.exchange(on ? 1 : 0); }

extern "C" int sfcvit_reduce_pending(void) {
    std::lock_guard<std::mutex> lock(g_reduce_mutex);
    return int(g_reduce_queue.size());
}

extern "C" int sfcvit_reduce_discard(void) {
    std::lock_guard<std::mutex> lock(g_reduce_mutex);
    g_reduce_queue.clear();
    return SFCVIT_OK;
}

extern "C" int sfcvit_reduce_flush(void *stream) {
    std::vector<ReduceItem> items;
    {
        std::lock_guard<std::mutex> lock(g_reduce_mutex);
        items.swap(g_reduce_queue);
    }
    for (size_t i0 = 0; i0 < items.size(); i0 += RED_BATCH) {
        ReduceBatch b;
        b.n = int(std::min(items.size() - i0, size_t(RED_BATCH)));
        int blocks = 0;
        for (int i = 0; i < b.n; i++) {
            b.it[i] = items[i0 + i];
            b.first[i] = blocks;
            blocks += (b.it[i].ncols + 15) / 16;
        }
        b.first[b.n] = blocks;
        hipLaunchKernelGGL(reduce_batched_kernel, dim3(blocks), dim3(RED_THREADS), 0, static_cast<hipStream_t>(stream), b);
        if (int rc = check_launch("batched column reduce")) return rc;
    }
    return SFCVIT_OK;
}

extern "C" int64_t sfcvit_colsum_workspace(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    int cb, rb, rpb;
    colsum_plan(M, N, cb, rb, rpb);
    return int64_t(rb) * N * int64_t(sizeof(float));
}

extern "C" int sfcvit_colsum(const void *x, int M, int N, int ld, void *out, int out_bf16, void *workspace,
                             int64_t workspace_bytes, void *stream) {
    if (!x || !out) return fail(SFCVIT_EINVAL, "colsum: null pointer");
    if (M <= 0 || N <= 0 || N % 8 || ld % 8 || ld < N) return fail(SFCVIT_EINVAL, "colsum: M=%d N=%d ld=%d (N, ld %% 8 == 0)", M, N, ld);
    if (!aligned16(x)) return fail(SFCVIT_EINVAL, "colsum: alignment");
    const int64_t need = sfcvit_colsum_workspace(M, N);
    if (!workspace || workspace_bytes < need) return fail(SFCVIT_EINVAL, "colsum: workspace of %lld bytes needed", (long long)need);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int col_blocks, row_blocks, rpb;
    colsum_plan(M, N, col_blocks, row_blocks, rpb);
    float *part = static_cast<float *>(workspace);
    hipLaunchKernelGGL(colsum_kernel, dim3(col_blocks, row_blocks), dim3(THREADS), 0, s,
                       static_cast<const uint16_t *>(x), M, N, ld, rpb, part);
    if (int rc = check_launch("colsum")) return rc;
    return reduce_cols(part, row_blocks, N, N, out, out_bf16, stream);
}

extern "C" int sfcvit_transpose(const void *src, int R, int C, int lds, void *dst, int ldd, void *stream) {
    if (!src || !dst) return fail(SFCVIT_EINVAL, "transpose: null pointer");
    if (R <= 0 || C <= 0 || R % 8 || C % 8 || lds % 8 || ldd % 8 || lds < C || ldd < R)
        return fail(SFCVIT_EINVAL, "transpose: R=%d C=%d lds=%d ldd=%d (multiples of 8)", R, C, lds, ldd);
    if (!aligned16(src) || !aligned16(dst)) return fail(SFCVIT_EINVAL, "transpose: alignment");
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(THREADS), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(src), R, C, lds, static_cast<uint16_t *>(dst), ldd);
    return check_launch("transpose");
}

extern "C" int sfcvit_transpose_batched(const void *src_base, void *dst_base, const sfcvit_transpose_tile *tiles, int n_tiles,
                                        void *stream) {
    if (!src_base || !dst_base || !tiles || n_tiles <= 0) return fail(SFCVIT_EINVAL, "transpose_batched: null pointer or no tiles");
    if (!aligned16(src_base) || !aligned16(dst_base)) return fail(SFCVIT_EINVAL, "transpose_batched: alignment");
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(n_tiles), dim3(THREADS), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(src_base), static_cast<uint16_t *>(dst_base), tiles);
    return check_launch("transpose_batched");
}

extern "C" int sfcvit_gelu_fwd(const void *x, void *y, int64_t n, void *stream) {
    if (!x || !y || n <= 0 || n % 8) return fail(SFCVIT_EINVAL, "gelu_fwd: n=%lld must be a positive multiple of 8", (long long)n);
    hipLaunchKernelGGL(gelu_kernel<false>, dim3(grid_for(n / 8)), dim3(THREADS), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(nullptr), static_cast<const uint16_t *>(x), static_cast<uint16_t *>(y), n / 8);
    return check_launch("gelu_fwd");
}

extern "C" int sfcvit_gelu_bwd(const void *dy, const void *x, void *dx, int64_t n, void *stream) {
    if (!dy || !x || !dx || n <= 0 || n % 8) return fail(SFCVIT_EINVAL, "gelu_bwd: n=%lld must be a positive multiple of 8", (long long)n);
    hipLaunchKernelGGL(gelu_kernel<true>, dim3(grid_for(n / 8)), dim3(THREADS), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(dy), static_cast<const uint16_t *>(x), static_cast<uint16_t *>(dx), n / 8);
    return check_launch("gelu_bwd");
}

static int gelu_drop_check(const void *x, const void *y, int rows, int cols, float p, const char *what) {
    if (!x || !y || rows <= 0 || cols <= 0 || cols % 8) return fail(SFCVIT_EINVAL, "%s: rows=%d cols=%d (cols %% 8 == 0)", what, rows, cols);
    if (!(p >= 0.f && p < 1.f)) return fail(SFCVIT_EINVAL, "%s: dropout p=%g", what, p);
    return SFCVIT_OK;
}

extern "C" int sfcvit_gelu_drop_fwd(const void *x, void *y, int rows, int cols, float p, uint32_t seed, const uint32_t *seed_off,
                                    void *stream) {
    if (int rc = gelu_drop_check(x, y, rows, cols, p, "gelu_drop_fwd")) return rc;
    hipLaunchKernelGGL(gelu_drop_kernel<false>, dim3(grid_for(int64_t(rows) * cols / 8)), dim3(THREADS), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint16_t *>(nullptr),
                       static_cast<const uint16_t *>(x), static_cast<uint16_t *>(y), rows, cols, p, seed, seed_off);
    return check_launch("gelu_drop_fwd");
}

extern "C" int sfcvit_gelu_drop_bwd(const void *dy, const void *x, void *dx, int rows, int cols, float p, uint32_t seed,
                                    const uint32_t *seed_off, void *stream) {
    if (!dy) return fail(SFCVIT_EINVAL, "gelu_drop_bwd: null pointer");
    if (int rc = gelu_drop_check(x, dx, rows, cols, p, "gelu_drop_bwd")) return rc;
    hipLaunchKernelGGL(gelu_drop_kernel<true>, dim3(grid_for(int64_t(rows) * cols / 8)), dim3(THREADS), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint16_t *>(dy), static_cast<const uint16_t *>(x),
                       static_cast<uint16_t *>(dx), rows, cols, p, seed, seed_off);
    return check_launch("gelu_drop_bwd");
}

extern "C" int sfcvit_dropout_mask(void *out, int64_t rows, int cols, float p, uint32_t seed, void *stream) {
    if (!out || rows <= 0 || cols <= 0 || !(p >= 0.f && p < 1.f)) return fail(SFCVIT_EINVAL, "dropout_mask: bad argument");
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(rows * ((cols + 1) / 2))), dim3(THREADS), 0,
                       static_cast<hipStream_t>(stream), static_cast<uint16_t *>(out), rows, cols, p, seed);
    return check_launch("dropout_mask");
}

extern "C" int sfcvit_soft_ce(const void *logits, const float *targets, float *loss_rows, void *dlogits, int B, int C,
                              int ld, float gscale, void *stream) {
    if (!logits || !targets || !loss_rows) return fail(SFCVIT_EINVAL, "soft_ce: null pointer");
    if (B <= 0 || C <= 0 || ld < C) return fail(SFCVIT_EINVAL, "soft_ce: B=%d C=%d ld=%d", B, C, ld);
    hipLaunchKernelGGL(soft_ce_kernel, dim3((B + WAVES - 1) / WAVES), dim3(THREADS), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(logits), targets, loss_rows, static_cast<uint16_t *>(dlogits), B, C, ld, gscale);
    return check_launch("soft_ce");
}

extern "C" int sfcvit_sumsq_accum(const void *g, int64_t n, int is_f32, float *out, void *workspace, void *stream) {
    if (!g || !out || !workspace || n <= 0) return fail(SFCVIT_EINVAL, "sumsq: bad argument");
    if (!aligned16(g)) return fail(SFCVIT_EINVAL, "sumsq: alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int grid = grid_for((n + 7) / 8);
    const int max_parts = SFCVIT_SUMSQ_WORKSPACE_BYTES / int(sizeof(float));
    if (grid > max_parts) grid = max_parts;
    float *part = static_cast<float *>(workspace);
    if (is_f32) hipLaunchKernelGGL(sumsq_kernel<true>, dim3(grid), dim3(THREADS), 0, s, g, n, part);
    else hipLaunchKernelGGL(sumsq_kernel<false>, dim3(grid), dim3(THREADS), 0, s, g, n, part);
    if (int rc = check_launch("sumsq")) return rc;
    hipLaunchKernelGGL(sumsq_reduce_kernel, dim3(1), dim3(64), 0, s, part, grid, out);
    return check_launch("sumsq reduce");
}

extern "C" int sfcvit_adamw_step(const sfcvit_adamw_args *a, void *stream) {
    if (!a || !a->param || !a->master || !a->grad || !a->m || !a->v) return fail(SFCVIT_EINVAL, "adamw: null pointer");
    if (a->n <= 0 || (a->step < 1 && !a->dev_state)) return fail(SFCVIT_EINVAL, "adamw: n=%lld step=%d", (long long)a->n, a->step);
    const float bc1 = 1.f - powf(a->beta1, float(a->step));
    const float bc2 = 1.f - powf(a->beta2, float(a->step));
    if (!aligned16(a->param) || !aligned16(a->master) || !aligned16(a->grad) || !aligned16(a->m) || !aligned16(a->v))
        return fail(SFCVIT_EINVAL, "adamw: buffers must be 16-byte aligned");
    static const int nt = [] { const char *e = getenv("SFCVIT_ADAMW_NT"); return e ? atoi(e) & 3 : 3; }();
#define ADAMW(NT) hipLaunchKernelGGL(adamw_kernel<NT>, dim3(grid_for((a->n + 7) / 8)), dim3(THREADS), 0, static_cast<hipStream_t>(stream), *a, bc1, 1.f / sqrtf(bc2))
    if (nt == 3) ADAMW(3); else if (nt == 2) ADAMW(2); else if (nt == 1) ADAMW(1); else ADAMW(0);
#undef ADAMW
    return check_launch("adamw");
}

// ---------------------------------------------------------------------------
// Device-resident step state: [0] dropout seed offset (uint32)  [1] step count (int32)  [2] learning rate (float,
// written by the host, read by adamw)  [3] 1 - beta1^step  [4] 1 / sqrt(1 - beta2^step).  One thread advances it once
// per training step -- as the first node of a captured step graph, so that every replay draws new dropout masks and
// applies the right Adam bias correction without any by-value kernel argument changing.
// ---------------------------------------------------------------------------
namespace sfcvit {
namespace {
__global__ void step_advance_kernel(uint32_t *state, float beta1, float beta2, uint32_t seed_base) {
    if (threadIdx.x || blockIdx.x) return;
    const int step = int(state[1]) + 1;
    state[1] = uint32_t(step);
    state[0] = mix32(seed_base ^ (uint32_t(step) * 0x9E3779B1u));
    float *f = reinterpret_cast<float *>(state);
    f[3] = 1.f - powf(beta1, float(step));
    f[4] = 1.f / sqrtf(1.f - powf(beta2, float(step)));
}
}  // namespace
}  // namespace sfcvit

extern "C" int sfcvit_step_advance(void *state, float beta1, float beta2, uint32_t seed_base, void *stream) {
    using namespace sfcvit;
    if (!state || !aligned16(state)) return fail(SFCVIT_EINVAL, "step_advance: state must be a 16-byte aligned device buffer of 8 words");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<uint32_t *>(state), beta1,
                       beta2, seed_base);
    return check_launch("step_advance");
}
