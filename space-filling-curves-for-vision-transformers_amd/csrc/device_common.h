// Device-side helpers for gfx950 (CDNA4): wave = 64 lanes, bf16 MFMA 16x16x32.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfcvit {

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define SFCVIT_LDS __attribute__((address_space(3)))

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(uint32_t(b) << 16); }

// Round-to-nearest-even fp32 -> bf16; a plain cast lets hipcc pick v_cvt_pk_bf16_f32
// (NaN stays NaN, MI355X_MICROARCH.md "Correctness boundaries").
__device__ __forceinline__ uint16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}
// Two at once: the vector conversion is ONE v_cvt_pk_bf16_f32; two scalar casts + shift + or compiled to two of them
// (each with a dummy second operand) plus three fix-up instructions per pair.
typedef __attribute__((ext_vector_type(2))) float f32x2_cvt;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_cvt;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const f32x2_cvt v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_cvt));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// ---- LDS images used by the MFMA kernels ------------------------------------
// "kc" image: [rows][64] bf16 (128-B rows), k contiguous.  16-B chunk index is XOR-ed with kc_swz(row) =
// 2 ((row >> 1) & 3): conflict-free for ds_write_b128 staging, for the ds_read_b128 fragment reads below AND for the
// transposed ds_read_b64_tr_b16 reads of the same image (attention backward reads Q, dO and K both ways)
// (tools/lds_conflicts.py).  Rounds 1-2 XOR-ed with (row >> 1) & 7: the same for the row reads, but every transposed
// read took two LDS passes (rows r and r + 2 of a 4-row block on the same banks) -- 36 % of the LDS-active cycles of the
// one-pass attention backward were bank conflicts (profiles/r2/README.md).
__device__ __forceinline__ int kc_swz(int row) { return ((row >> 1) & 3) << 1; }
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ kc_swz(row)) << 4); }

// Fragment of 16 rows x 32 k from a kc image: lane holds row (lane&15), k = 8*(lane>>4)+j.
__device__ __forceinline__ bf16x8 kc_frag(const char *img, int row0, int kk, int lane) {
    const int row = row0 + (lane & 15);
    return *reinterpret_cast<const bf16x8 *>(img + kc_off(row, kk * 4 + (lane >> 4)));
}

// "st" image: [64 k][128 cols] bf16 (256-B rows), cols contiguous (operand stored
// k-major in memory).  32-B chunk index XOR-ed with (k&3) | ((k>>3)&1)<<2:
// conflict-free for ds_write_b128 staging and for the transposed reads.
__device__ __forceinline__ int st_off(int krow, int col) {
    int c32 = (col >> 4) ^ ((krow & 3) | (((krow >> 3) & 1) << 2));
    return krow * 256 + (c32 << 5) + ((col & 15) << 1);
}

// Same fragment (16 cols x 32 k, lane holds col (lane&15), k = 8*(lane>>4)+j) from an
// st image via two ds_read_b64_tr_b16 (4 k-rows x 16 cols per 16-lane group each).
// EXEC must be all ones (cdna_hip_programming.md T10).
__device__ __forceinline__ bf16x8 st_frag(const char *img, int col0, int kk, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int k0 = kk * 32 + 8 * g + q;
    const int col = col0 + 4 * p;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFCVIT_LDS bf16x4 *)(img + st_off(k0, col)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFCVIT_LDS bf16x4 *)(img + st_off(k0 + 4, col)));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// ---- counter-based dropout mask ------------------------------------------------
// Element (row, col) of a [rows, cols] tensor is KEPT iff the top 16 bits of one 32-bit multiply-shift hash of
// (row key, col) are >= thresh = p * 65536:
//     keep(row, col) = ((row_key ^ col * 0x9E3779B1) * 0x2C1B3C6D) >= (thresh << 16)          [mod 2^32]
// with row_key = a murmur3-finalised hash of (seed, row) worked out once per row.  The same function is evaluated in
// forward and backward (nothing is stored) and by sfcvit_dropout_mask (tests).  Per element that is one xor, one 32-bit
// multiply and one compare on top of col * G, which is a lane constant plus a literal in every kernel.
// Rounds 1-3 drew TWO 16-bit uniforms from one xorshift-multiply-xorshift hash of (row key, col / 2): 6 vector
// instructions per pair -- but per ELEMENT in the one-pass attention backward, where a lane holds one key and four queries
// (different rows), and there the hash was 24 of the 50 vector instructions of a score tile (profiles/r4/attention_isa_budget.txt).
// Statistics of the new mask (tools/dropout_mask_stats.py, 37 632 rows x 3 072 columns, p = 0.1 and 0.5): drop rate exact
// to 1e-4, row and column rates binomially distributed, correlations at lags 1-2 along either axis and across seeds
// within sampling noise (1e-4) -- the same figures as the old hash.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t drop_row_key(uint32_t seed, uint64_t row) {
    return mix32(uint32_t(row) * 0x9E3779B1u + mix32(seed ^ (uint32_t(row >> 32) * 0x7FEB352Du)));
}
constexpr uint32_t DROP_G = 0x9E3779B1u, DROP_C = 0x2C1B3C6Du;
// hash of element `col` of the row with key `row_key`; col_g = col * DROP_G (callers that walk a row keep it as base + literal)
__device__ __forceinline__ uint32_t drop_hash_g(uint32_t row_key, uint32_t col_g) { return (row_key ^ col_g) * DROP_C; }
__device__ __forceinline__ uint32_t drop_hash(uint32_t row_key, uint32_t col) { return drop_hash_g(row_key, col * DROP_G); }
// Effective seed of a dropout site: the by-value seed of the call plus the device-resident per-step offset (NULL = 0).
// With the offset a captured hipGraph draws fresh masks on every replay (sfcvit_step_advance moves the offset), and
// forward and backward of one step still agree (both read it between two advances).
__device__ __forceinline__ uint32_t eff_seed(uint32_t seed, const uint32_t *off) { return off ? seed + *off : seed; }
// thresh: 16-bit (p * 65536, p < 1); the kernels compare the 32-bit hash with thresh << 16 (drop_thresh32)
__device__ __forceinline__ uint32_t drop_thresh(float p) { return uint32_t(p * 65536.f + 0.5f); }
__device__ __forceinline__ bool drop_keep(uint32_t row_key, uint32_t col, uint32_t thresh) { return drop_hash(row_key, col) >= (thresh << 16); }
// keep flags of the two elements (cols 2*pair, 2*pair+1) of a row (the interface of rounds 1-3: a pair per call)
__device__ __forceinline__ void drop_keep2(uint32_t row_key, uint32_t pair_in_row, uint32_t thresh, bool &k0, bool &k1) {
    const uint32_t g0 = pair_in_row * (2u * DROP_G), t32 = thresh << 16;
    k0 = drop_hash_g(row_key, g0) >= t32;
    k1 = drop_hash_g(row_key, g0 + DROP_G) >= t32;
}

// Wait states between the last MFMA of a sequence and the first VALU / LDS / store read of its
// accumulators.  hipcc (ROCm 7.2) pads this hazard inside a basic block but was observed to miss it
// across a taken branch (attention forward: `v_mfma ...; s_cbranch; v_max3 <acc>` read accumulators
// two instructions after the MFMA and returned pre-MFMA values -- a run-to-run varying row max).
// 16 states cover the 8-pass bf16 MFMAs; call it wherever control flow separates the two.
// The sched_barriers are what makes it a fence: an asm statement, volatile and memory-clobbering as it may be, orders
// memory operations, not MFMAs -- round 4 found three QK^T MFMAs of the forward kernel scheduled BEHIND the nops.  That round's
// failure (whole query fragments of attn_seq_fwd_kernel<13, true> wrong, run to run, lse intact; tools/attn_fwd_repeat_check.py,
// attn_fwd_race_probe.py) came and went with every change of the code between the MFMAs and the branches around them -- the
// per-fragment run-time key-mask tests, exec-masked loads in front of the QK^T MFMAs -- and was 50x rarer with
// -amdgpu-mfma-padding-ratio=100: hazards across basic-block boundaries next to MFMAs.  Rule kept since: no branch between
// an MFMA and the first reader or overwriter of its operands; real fences at the phase boundaries.
__device__ __forceinline__ void mfma_fence() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Compile-time loop: register arrays (MFMA accumulators, score rows) must only ever be indexed by constants; a loop
// the unroller gives up on turns the index into a runtime value and the array into scratch memory.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Sum over the 16 lanes of a DPP row (lanes that share lane >> 4), result in every lane of the row.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// 64-lane butterfly reductions.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace sfcvit
