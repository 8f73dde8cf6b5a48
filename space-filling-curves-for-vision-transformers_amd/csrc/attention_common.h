// Shared device helpers of the attention kernels (attention.hip: tiled, any N;
// attention_seq.hip: whole sequence resident in LDS, N <= 256).
#pragma once
#include "../../include/sfcvit.h"
#include "device_common.h"

namespace sfcvit {
namespace attn {

constexpr int HD = 64;
constexpr int THREADS = 256;
constexpr int BLK = 64;                 // rows (keys or queries) per LDS block
constexpr int IMG_BYTES = BLK * HD * 2;  // 8 KiB

// [64 rows][64 cols] bf16 image for transposed reads only: 32-B chunk ^ ((row >> 1) & 3).
__device__ __forceinline__ int vt_off(int row, int col) {
    return row * 128 + ((((col >> 4) ^ ((row >> 1) & 3))) << 5) + ((col & 15) << 1);
}

// Stage 64 rows x 64 cols from global (row stride `ld` elements) into an LDS image.
// Rows >= nvalid are zero-filled.  VT = false: "kc" layout, true: "vt" layout.
template <bool VT>
__device__ __forceinline__ void stage64(char *img, const uint16_t *__restrict__ src, int ld, int row0, int nvalid,
                                        int tid) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int v = tid + THREADS * i;
        const int r = v >> 3, c16 = v & 7;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (row0 + r < nvalid) val = *reinterpret_cast<const u32x4 *>(src + size_t(row0 + r) * ld + c16 * 8);
        const int off = VT ? vt_off(r, c16 * 8) : kc_off(r, c16);
        *reinterpret_cast<u32x4 *>(img + off) = val;
    }
}

// A-operand fragment of X^T for a 32-deep contraction over image rows:
// lane (g, i) gets X[rows r_lo+4g+{0..3}, r_hi+4g+{0..3}][col0 + i].
template <bool VT>
__device__ __forceinline__ bf16x8 tr_frag(const char *img, int r_lo, int r_hi, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int col = col0 + 4 * p;
    const int ra = r_lo + 4 * g + q, rb = r_hi + 4 * g + q;
    int oa, ob;
    if (VT) {
        oa = vt_off(ra, col);
        ob = vt_off(rb, col);
    } else {
        oa = kc_off(ra, col >> 3) + ((col & 7) << 1);
        ob = kc_off(rb, col >> 3) + ((col & 7) << 1);
    }
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFCVIT_LDS bf16x4 *)(img + oa));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFCVIT_LDS bf16x4 *)(img + ob));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// Row fragment straight from global: lane (g, i) gets X[row0 + i][32*kk + 8g .. +7].
__device__ __forceinline__ bf16x8 global_frag(const uint16_t *__restrict__ src, int ld, int row0, int nvalid, int kk,
                                              int lane) {
    const int r = row0 + (lane & 15);
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < nvalid) z = *reinterpret_cast<const bf16x8 *>(src + size_t(r) * ld + kk * 32 + 8 * (lane >> 4));
    return z;
}

__device__ __forceinline__ bf16x8 pack_frag(const f32x4 &a, const f32x4 &b) {
    const u32x4 w = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, w);
}

// Per-lane LDS byte offsets of the fragment reads, computed once per thread.  All images have
// 128-byte rows and every fragment starts at a row that is a multiple of 16, which leaves the
// swizzle terms lane-constant: a read is then `image + row0 * 128 + constant`.
struct LaneOff {
    int k[2];    // kc image, row fragment:        row (lane&15), 16-B chunk kk*4 + (lane>>4)
    int t[4];    // kc image, transposed fragment: row 4*(lane>>4) + ((lane&15)>>2), cols 16*hf + 4*(lane&3)
    int tv[4];   // vt image, transposed fragment
};
__device__ __forceinline__ LaneOff lane_offsets(int lane) {
    LaneOff o;
    const int r = lane & 15, g = lane >> 4, q = r >> 2, p = r & 3, tr = 4 * g + q;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) o.k[kk] = kc_off(r, kk * 4 + g);
#pragma unroll
    for (int hf = 0; hf < 4; hf++) {
        const int col = 16 * hf + 4 * p;
        o.t[hf] = kc_off(tr, col >> 3) + ((col & 7) << 1);
        o.tv[hf] = vt_off(tr, col);
    }
    return o;
}
__device__ __forceinline__ bf16x8 kc_frag_at(const char *img, int row0, int lane_off) {
    return *reinterpret_cast<const bf16x8 *>(img + row0 * 128 + lane_off);
}
// transposed fragment for a 32-deep contraction over rows r_lo..r_lo+15 and r_lo+16..r_lo+31
__device__ __forceinline__ bf16x8 tr_frag_at(const char *img, int r_lo, int lane_off) {
    const char *pa = img + r_lo * 128 + lane_off;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFCVIT_LDS bf16x4 *)pa);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFCVIT_LDS bf16x4 *)(pa + 16 * 128));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Start-up stagger of the whole-sequence kernels.  Their workgroups all take the same time and all begin with a load
// burst (K / V or Q / dO / K images of one (batch, head)): launched together they stay in lockstep for the whole kernel
// -- every round of workgroups hits HBM at once (~11 B/clock per CU) and then leaves it idle while it computes.  The
// workgroups of the FIRST round (linear id < first_round) therefore start `slot * ticks` late (ticks of the 100 MHz
// s_memrealtime clock), slot = id / per_slot: after that the rounds of the slots interleave load and compute phases
// for the rest of the kernel.  ticks == 0: off.
__device__ __forceinline__ void stagger_start(int first_round, int per_slot, int ticks) {
    if (ticks <= 0) return;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (lin >= first_round) return;
    const uint64_t until = __builtin_amdgcn_s_memrealtime() + uint64_t(lin / per_slot) * uint64_t(ticks);
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(32);
}

// reduce over the four 16-lane groups (same lane&15)
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// Store a transposed accumulator: acc[hf][r] = X[row = lane&15][col = 16hf + 4g + r] (g = lane >> 4), bf16, as TWO 16-byte
// stores per lane.  Packed, a lane holds four 8-byte pieces (one per hf) that lie 32 bytes apart in its row; the piece
// that continues it sits in the lane 16 further on.  v_permlane16_swap (odd 16-lane rows of the first operand <-> even rows
// of the second) on the pieces of an even / odd hf pair gives every lane 16 contiguous bytes:
//     g even: pieces (h_even, g), (h_even, g + 1) -> columns 16 h_even + 8 (g >> 1) .. + 7
//     g odd:  pieces (h_odd, g - 1), (h_odd, g)   -> columns 16 h_odd  + 8 (g >> 1) .. + 7
// so a store instruction writes 16 rows x 64 contiguous bytes instead of 16 rows x 4 pieces of 8 bytes (rounds 1-3: four
// 8-byte stores per lane).  Measured where it showed (round 4, stamped build of the one-pass backward): the 8 stores of a
// key wave's dK / dV took ~2 200 clocks to issue and the 3-4 key waves of a SIMD issued theirs one after the other -- 7 000
// clocks per (batch, head) in which nothing else ran.  All 64 lanes must be active (the swap crosses lanes); `valid` masks
// the stores only.
__device__ __forceinline__ void store_rows(uint16_t *__restrict__ dst, int ld, int row, bool valid, const f32x4 (&acc)[4],
                                           float mul, int lane) {
    uint32_t p[4][2];
#pragma unroll
    for (int hf = 0; hf < 4; hf++) {
        p[hf][0] = pack2bf(acc[hf][0] * mul, acc[hf][1] * mul);
        p[hf][1] = pack2bf(acc[hf][2] * mul, acc[hf][3] * mul);
    }
    const int g = lane >> 4;
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {
        const auto lo = __builtin_amdgcn_permlane16_swap(p[2 * pr][0], p[2 * pr + 1][0], false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(p[2 * pr][1], p[2 * pr + 1][1], false, false);
        const u32x4 o = {lo[0], hi[0], lo[1], hi[1]};
        if (valid) *reinterpret_cast<u32x4 *>(dst + size_t(row) * ld + 16 * (2 * pr + (g & 1)) + 8 * (g >> 1)) = o;
    }
}

// Dropout keep-factors of 4 consecutive keys (key0 % 4 == 0) of the mask row with key `row_key`.
__device__ __forceinline__ void drop_keep4(uint32_t row_key, int key0, uint32_t th, float sc, float (&keep)[4]) {
    bool k[4];
    drop_keep2(row_key, uint32_t(key0 >> 1), th, k[0], k[1]);
    drop_keep2(row_key, uint32_t(key0 >> 1) + 1, th, k[2], k[3]);
#pragma unroll
    for (int r = 0; r < 4; r++) keep[r] = k[r] ? sc : 0.f;
}

}  // namespace attn
}  // namespace sfcvit
