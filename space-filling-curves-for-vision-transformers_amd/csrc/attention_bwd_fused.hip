// Fused attention backward for gfx950 (head dim 64, N <= 224: ViT-B/16 @ 224 has N = 196): dK, dV AND dQ of one
// (batch, head) in ONE pass over the scores -- five MFMA products per score tile (S, dP, dV, dK, dQ) instead of the
// seven of the two-kernel form (attention_seq.hip: a dK/dV kernel and a dQ kernel that each recompute S and dP), the
// exp / dropout-hash / dS arithmetic once instead of twice, and q, k, v, dO read from HBM once instead of twice.
//
//   * one workgroup of 16 waves per (batch, head); Q, dO and K of the whole sequence are staged once into LDS by
//     LDS-DMA (3 x 28 KiB at N = 196); every KEY wave (wave w < nf owns the 16 keys 16w .. 16w+15) fetches its K / V
//     fragments straight into registers, so V never touches LDS.
//   * the sequence is walked in chunks of 32 queries.  In chunk c a key wave computes S^T and dP^T of its keys against
//     the chunk's queries (key on the MFMA lane: the accumulators are the B operands of the dV and dK products without
//     any data movement), turns them into P and dS, adds P^T dO into dV and dS^T Q into dK -- exactly the loop body of
//     attn_seq_bwd_kv_kernel -- and additionally writes its [16 keys][32 queries] block of dS^T (bf16) into a
//     double-buffered exchange image in LDS.
//   * the last two waves are dQ waves: in chunk c they read the complete dS^T image of chunk c - 1 ([keys][32 queries],
//     transposed reads) and K^T (transposed reads of the K image) and produce dQ^T = K^T dS^T for those 32 queries,
//     32 head columns each, and store it.  One workgroup barrier per chunk orders writer and reader; the two halves of
//     the exchange image alternate.  Every dQ tile is summed by one wave in a fixed key order: results are bit-identical
//     from run to run (no atomics).
//   * delta = rowsum(dO . O) is worked out here as well, from the staged dO image and one 16-byte load of O per thread
//     (the two-kernel form runs a separate pass over dO and O for it), and -- when the caller asks -- the column sums
//     of dQ, dK, dV over the sequence (= this (batch, head)'s contribution to the in_proj bias gradient,
//     torch:nn/functional.py:5822-5833) leave the kernel as 192 floats instead of being re-read from the 231 MB dqkv
//     tensor by a column-sum pass.
//   * rows >= N: the DMA fills them with copies of row N - 1 (finite); padded queries get lse = +inf (P = dS = 0), padded
//     keys get dS = 0 before the exchange and their dK / dV rows are not stored; exchange rows of keys >= 16 nf are
//     zeroed once.
#include <algorithm>
#include "attention_common.h"
#include "common_host.h"

namespace sfcvit {
namespace {

using namespace attn;

constexpr int FT = 1024, FWAVES = 16, FMAXC = 7;   // threads, waves, 32-row chunks (N <= 224)
constexpr int FUSED_ROW_BYTES = 3 * 128 + 2 * 64 + 3 * 4;    // LDS bytes per padded sequence row
constexpr int FUSED_POST_BYTES = 16384;                      // reused after the loop (short sequences: the allocation's floor)
constexpr int FUSED_EXTRA = 256;                             // + 64 column sums of dQ

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// Stage npad rows x 64 cols (128-B rows) into a kc image by LDS-DMA; the bank swizzle goes on the SOURCE chunk
// (the DMA writes lane-linear); rows >= N copy row N - 1.
__device__ __forceinline__ void dma_rows(char *img, const uint16_t *__restrict__ src, int ld, int N, int npad, int tid) {
    for (int p = tid; p < npad * 8; p += FT) {
        const int row = p >> 3, cs = p & 7;
        const int c = cs ^ kc_swz(row);
        const uint16_t *g = src + size_t(min(row, N - 1)) * ld + c * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(img + p * 16), 16, 0, 0);
    }
}

// dS^T exchange image: [key rows][32 queries] bf16, 64-B rows; the two 32-B halves (query fragment 0 / 1 of the chunk)
// are swapped on rows with (row >> 2) & 1 set, which makes both the 8-byte writes of a key wave and the transposed
// reads of a dQ wave spread over all banks.
__device__ __forceinline__ bf16x8 ds_tr_frag(const char *slot, int key0, int lane_off) {
    const char *pa = slot + key0 * 64 + lane_off;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFCVIT_LDS bf16x4 *)pa);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFCVIT_LDS bf16x4 *)(pa + 16 * 64));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// NFC: number of 16-row fragments at compile time (0 = from N); DROP: dropout on the probabilities (compile-time: a
// runtime flag put a branch around every hash, each one a scheduling barrier between the MFMAs).
template <int NFC, bool DROP>
__global__ __launch_bounds__(FT) void attn_seq_bwd_fused_kernel(const sfcvit_attn_args a, int npad, int stag_round, int stag_per, int stag_ticks, int dq_sums) {

    extern __shared__ __attribute__((aligned(16))) char smem[];
    stagger_start(stag_round, stag_per, stag_ticks);
    char *qimg = smem, *doimg = smem + npad * 128, *kimg = smem + 2 * npad * 128, *dsb = smem + 3 * npad * 128;
    float *lse_s = reinterpret_cast<float *>(dsb + 2 * npad * 64), *del_s = lse_s + npad;
    uint32_t *rkey_s = reinterpret_cast<uint32_t *>(del_s + npad);   // dropout row key of every query
    // after the loop the images are dead and the first 16 KiB of LDS are reused: [FWAVES][128] column-sum staging and the
    // [32][64] partial dK / dV of a shared fragment; the dQ column sums live behind everything (FUSED_POST_BYTES)
    float *qcs = reinterpret_cast<float *>(smem + max(npad * FUSED_ROW_BYTES, FUSED_POST_BYTES));   // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, h = blockIdx.x, N = a.N, D = a.H * HD, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    const uint16_t *dop = static_cast<const uint16_t *>(a.dout) + size_t(b) * N * D + h * HD;
    const float *lse = a.lse + (size_t(b) * a.H + h) * N;
    const int nf = NFC ? NFC : (N + 15) >> 4, nc = NFC ? ((NFC + 1) >> 1) : npad >> 5;
    // Key fragments -> waves.  With nf <= 13 a wave is spare, and the LAST fragment (at N = 196: 4 valid keys of 16, a
    // whole wave's VALU work all the same) is shared by waves nf - 1 and nf, one query fragment of every chunk each: the
    // loop is VALU-bound per SIMD, waves go to SIMDs round-robin, and 13 whole fragments put 4 on one SIMD and 3 on the
    // others -- halves make it 3.5 / 3.5 / 3 / 3.  The two partial dK / dV meet in LDS after the loop.
    const bool split = nf <= FWAVES - 3;
    const int kfi = (split && wave == nf) ? nf - 1 : wave;             // the key fragment this wave works on
    const int tsel = !split ? -1 : wave == nf - 1 ? 0 : wave == nf ? 1 : -1;   // -1: both query fragments of a chunk
    const bool is_key = wave < nf || (split && wave == nf), is_dq = wave >= FWAVES - 2;
    bf16x8 kf[2], vf[2];                              // this wave's 16 keys: V from HBM here, K from the staged image below
#pragma unroll
    for (int kk = 0; kk < 2; kk++) vf[kk] = global_frag(vp, ld, 16 * kfi, is_key ? N : 0, kk, lane);
    // O, for delta: the 16-byte piece that pairs with this thread's piece(s) of the dO image (same row, same swizzled chunk)
    u32x4 opiece[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int p = tid + FT * i, row = p >> 3, c = (p & 7) ^ kc_swz(row);
        opiece[i] = u32x4{0u, 0u, 0u, 0u};
        if (row < N) opiece[i] = *reinterpret_cast<const u32x4 *>(static_cast<const uint16_t *>(a.out) + (size_t(b) * N + row) * D + h * HD + c * 8);
    }
    dma_rows(qimg, qp, ld, N, npad, tid);
    dma_rows(doimg, dop, D, N, npad, tid);
    dma_rows(kimg, kp, ld, N, npad, tid);
    const uint32_t dth = drop_thresh(a.dropout_p);
    for (int i = tid; i < npad; i += FT) {
        lse_s[i] = i < N ? lse[i] * 1.4426950408889634f : INFINITY;   // padded queries: p = exp2(-inf) = 0
        rkey_s[i] = drop_row_key(eff_seed(a.dropout_seed, a.seed_off), (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(i));
    }
    if (tid < 64) qcs[tid] = 0.f;
    {   // exchange rows no key wave writes (keys 16 nf .. npad - 1), both halves of the double buffer
        const int nz = (npad - 16 * nf) * 4;         // 16-byte pieces per half
        for (int i = tid; i < 2 * nz; i += FT) {
            char *dst = dsb + (i >= nz ? npad * 64 : 0) + 16 * nf * 64 + (i >= nz ? i - nz : i) * 16;
            *reinterpret_cast<u32x4 *>(dst) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();                                  // LDS-DMA pending: hipcc drains vmcnt(0) here
    // delta[q] = sum_c dO[q][c] O[q][c]: 8 products per thread and piece, summed over the 8 lanes that share a row
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int p = tid + FT * i, row = p >> 3;
        float part = 0.f;
        if (p < npad * 8) {
            const u32x4 dpiece = *reinterpret_cast<const u32x4 *>(doimg + p * 16);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                part += __uint_as_float(dpiece[j] << 16) * __uint_as_float(opiece[i][j] << 16);
                part += __uint_as_float(dpiece[j] & 0xFFFF0000u) * __uint_as_float(opiece[i][j] & 0xFFFF0000u);
            }
        }
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if ((tid & 7) == 0 && p < npad * 8) del_s[row] = part;          // rows >= N: O piece = 0 -> 0
    }
    __syncthreads();

    const float scale = a.scale, c2 = a.scale * 1.4426950408889634f;
    // lane offsets of the fragment reads: the XOR swizzle commutes with the fragment index, so ONE register per kind
    // (k_off ^ (kk << 6), t_off ^ (hf << 5)) instead of the six of LaneOff -- the register file is full
    const int k_off = lane_offsets(lane).k[0], t_off = lane_offsets(lane).t[0];
    const float dsc = 1.f / (1.f - a.dropout_p);
    uint16_t *dbase = static_cast<uint16_t *>(a.dqkv) + size_t(b) * N * ld + h * HD;
    const int g = lane >> 4, li = lane & 15;
    const int key = 16 * kfi + li;                    // key waves: the key this lane's accumulator columns belong to
    const uint32_t key_g = uint32_t(key) * DROP_G, dth32 = dth << 16;
    if (is_key) {                                     // rows >= N of the image copy row N - 1: those keys are masked below
#pragma unroll
        for (int kk = 0; kk < 2; kk++) kf[kk] = kc_frag_at(kimg, 16 * kfi, (k_off ^ (kk << 6)));
    } else {
        kf[0] = kf[1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int hf = 0; hf < 4; hf++) dk[hf] = dv[hf] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c <= nc; c++) {
        if (is_key && c < nc) {
            char *slot = dsb + (c & 1) * npad * 64;
            f32x4 p[2], ds[2];
            // exchange image, writer side: row key, query fragment t at half t ^ ((key >> 2) & 1), 8 bytes at 8 g
            const int ds_w = key * 64 + (((li >> 2) & 1) << 5) + 8 * g;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int qf = 2 * c + t;
                if (tsel == 1 - t) {                                  // the other wave of a shared fragment does this one
                    p[t] = ds[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    continue;
                }
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(qimg, 16 * qf, (k_off ^ (kk << 6))), kf[kk], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(doimg, 16 * qf, (k_off ^ (kk << 6))), vf[kk], dp, 0, 0, 0);
                }
                const int ql0 = 16 * qf + 4 * g;                     // this lane's 4 queries: one 16-B LDS read each
                const f32x4 lse4 = *reinterpret_cast<const f32x4 *>(lse_s + ql0);    // lse * log2(e)
                const f32x4 del4 = *reinterpret_cast<const f32x4 *>(del_s + ql0);
                const u32x4 rk4 = *reinterpret_cast<const u32x4 *>(rkey_s + ql0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float pv = fast_exp2(s[r] * c2 - lse4[r]);
                    float keep = 1.f;                                 // 1 / (1 - p) where the element is kept, else 0
                    if (DROP)     // element (query row, key): xor with the lane's key constant, one multiply, one compare
                        keep = drop_hash_g(rk4[r], key_g) >= dth32 ? dsc : 0.f;
                    p[t][r] = pv * keep;
                    ds[t][r] = pv * (dp[r] * keep - del4[r]);             // x scale at the stores of dK and dQ
                }
                if (16 * kfi + 16 > N) {                             // boundary fragment (wave-uniform): keys >= N carry no gradient
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (key >= N) ds[t][r] = 0.f;
                }
            }
            const bf16x8 pf = pack_frag(p[0], p[1]), dsf = pack_frag(ds[0], ds[1]);
            {   // dS^T for the dQ waves: 4 consecutive queries of fragment t = 8 bytes
                const u32x4 w = __builtin_bit_cast(u32x4, dsf);
                if (tsel != 1) *reinterpret_cast<u32x2 *>(slot + ds_w) = u32x2{w[0], w[1]};
                if (tsel != 0) *reinterpret_cast<u32x2 *>(slot + (ds_w ^ 32)) = u32x2{w[2], w[3]};
            }
#pragma unroll
            for (int hf = 0; hf < 4; hf++) {
                dv[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(doimg, 32 * c, (t_off ^ (hf << 5))), pf, dv[hf], 0, 0, 0);
                dk[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(qimg, 32 * c, (t_off ^ (hf << 5))), dsf, dk[hf], 0, 0, 0);
            }
        } else if (is_dq && c >= 1) {
            const char *slot = dsb + ((c - 1) & 1) * npad * 64;
            // (lane constants of the dQ role are worked out here, per chunk, rather than kept live across the key waves' loop:
            // every wave runs the same kernel and the register file is full)
            // reader side: rows 4 g + (li >> 2) (+ 16), query columns 16 qf + 4 (li & 3): half qf ^ (g & 1)
            const int ds_r = (4 * g + (li >> 2)) * 64 + ((g & 1) << 5) + 8 * (li & 3);
            const int dqw = wave - (FWAVES - 2);              // head columns 32 dqw .. 32 dqw + 31
            int kt_off[2];                                    // K^T fragments of those columns (kc image, transposed read)
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int col = 16 * (2 * dqw + hh) + 4 * (li & 3), tr = 4 * g + (li >> 2);
                kt_off[hh] = kc_off(tr, col >> 3) + ((col & 7) << 1);
            }
            f32x4 acc[2][2];                                         // [head-column fragment hh][query fragment qf]
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i >> 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < FMAXC; ks++) {
                if (ks < nc) {                                       // key steps of 32 (keys are padded like the queries)
                    const bf16x8 b0 = ds_tr_frag(slot, 32 * ks, ds_r), b1 = ds_tr_frag(slot, 32 * ks, ds_r ^ 32);
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const bf16x8 kt = tr_frag_at(kimg, 32 * ks, kt_off[hh]);
                        acc[hh][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, b0, acc[hh][0], 0, 0, 0);
                        acc[hh][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, b1, acc[hh][1], 0, 0, 0);
                    }
                }
            }
            mfma_fence();
            if (a.colsum_part && dq_sums) {
                // column sums of this chunk's dQ tiles, accumulated in this wave's 32 LDS words (one writer per word, chunk
                // after chunk: a fixed order; registers for running sums are not to be had).  Padded queries have dS = 0.
#pragma unroll
                for (int hh = 0; hh < 2; hh++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float sq = row16_sum(acc[hh][0][r] + acc[hh][1][r]);
                        if (li == 0) qcs[32 * dqw + 16 * hh + 4 * g + r] += sq;
                    }
            }
            // acc[hh][qf][r] = dQ[query 32 (c-1) + 16 qf + li][head column 32 dqw + 16 hh + 4 g + r]
#pragma unroll
            for (int qf = 0; qf < 2; qf++) {
                const int q = 32 * (c - 1) + 16 * qf + li;
                if (q < N) {
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const u32x2 o = {pack2bf(acc[hh][qf][0] * scale, acc[hh][qf][1] * scale),
                                         pack2bf(acc[hh][qf][2] * scale, acc[hh][qf][3] * scale)};
                        *reinterpret_cast<u32x2 *>(dbase + size_t(q) * ld + 32 * dqw + 16 * hh + 4 * g) = o;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (split) {                                      // the shared fragment: wave nf hands its partial dK / dV to wave nf - 1
        float *px = reinterpret_cast<float *>(smem) + FWAVES * 128;    // [32][64] lane-private words, behind the column-sum area
        if (wave == nf) {
            mfma_fence();
#pragma unroll
            for (int hf = 0; hf < 4; hf++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    px[(4 * hf + r) * 64 + lane] = dk[hf][r];
                    px[(16 + 4 * hf + r) * 64 + lane] = dv[hf][r];
                }
        }
        __syncthreads();
        if (wave == nf - 1) {
            mfma_fence();
#pragma unroll
            for (int hf = 0; hf < 4; hf++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    dk[hf][r] += px[(4 * hf + r) * 64 + lane];
                    dv[hf][r] += px[(16 + 4 * hf + r) * 64 + lane];
                }
        }
    }
    const bool owns = wave < nf;                      // the wave that holds a fragment's complete dK / dV
    if (owns) {
        mfma_fence();
        store_rows(dbase + D, ld, key, key < N, dk, scale, lane);
        store_rows(dbase + 2 * D, ld, key, key < N, dv, 1.f, lane);
    }
    if (a.colsum_part) {
        // column sums of this (batch, head)'s dQ | dK | dV block: lanes of a 16-lane row hold the 16 rows of a column
        // (DPP reduction), the 13 key waves / 2 dQ waves meet in LDS (the exchange image is free now) and are summed in
        // a fixed order: bit-reproducible.  Layout of the partials: [batch][q | k | v thirds of 3 D], summed over
        // the batch afterwards (launch_colsum_reduce).
        float *cs = reinterpret_cast<float *>(smem);             // [FWAVES][128]
        if (owns) {
#pragma unroll
            for (int hf = 0; hf < 4; hf++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float sk = row16_sum(key < N ? dk[hf][r] : 0.f), sv = row16_sum(key < N ? dv[hf][r] : 0.f);
                    if (li == 0) {
                        cs[wave * 128 + 16 * hf + 4 * g + r] = sk * scale;
                        cs[wave * 128 + 64 + 16 * hf + 4 * g + r] = sv;
                    }
                }
        }
        __syncthreads();
        if (tid < 192) {
            const int third = tid >> 6, c = tid & 63;
            float s = 0.f;
            if (third == 0) s = qcs[c] * scale;
            else
                for (int w = 0; w < nf; w++) s += cs[w * 128 + (third - 1) * 64 + c];
            a.colsum_part[size_t(b) * 3 * D + third * D + h * HD + c] = s;
        }
    }
}

constexpr int FUSED_MAX_N = 32 * FMAXC;
constexpr int FUSED_MAX_LDS = FUSED_MAX_N * FUSED_ROW_BYTES + FUSED_EXTRA;

template <int NFC, bool DROP>
int launch_fused(const sfcvit_attn_args &a, int npad, size_t lds, int round, int per, int ticks, int dq_sums, hipStream_t s) {
    if (int rc = raise_lds_limit(reinterpret_cast<const void *>(&attn_seq_bwd_fused_kernel<NFC, DROP>), FUSED_MAX_LDS, "attention_bwd_fused attribute")) return rc;
    hipLaunchKernelGGL((attn_seq_bwd_fused_kernel<NFC, DROP>), dim3(a.H, a.B), dim3(FT), lds, s, a, npad, round, per, ticks, dq_sums);
    return check_launch("attention_bwd_fused");
}

}  // namespace

// -1: not eligible (the caller falls back to the two-kernel form); else a status.
int attn_seq_bwd_fused(const sfcvit_attn_args &a, int dq_sums, hipStream_t s) {
    if (a.hd != HD || a.N > FUSED_MAX_N) return -1;
    const int npad = (a.N + 31) / 32 * 32;
    const bool nf13 = (a.N + 15) / 16 == 13, drop = a.dropout_p > 0.f;
    const size_t lds = size_t(std::max(npad * FUSED_ROW_BYTES, FUSED_POST_BYTES)) + FUSED_EXTRA;
    // Start-up stagger (attention_common.h): one workgroup per CU, every one of them opens with a 117 KiB load burst and they
    // all take the same time, so launched together they stay in lockstep.  Two slots 4.5 us apart: 270.8 -> 257.1 us at
    // ViT-B / 256 (3 or 4 slots, 2-8 us: 255.6-258.6).  SFCVIT_ATTN_STAGGER_BWD = "slots,ticks" (10 ns) overrides; "1,0" = off.
    int slots = 2, ticks = 450;
    if (const char *e = getenv("SFCVIT_ATTN_STAGGER_BWD")) sscanf(e, "%d,%d", &slots, &ticks);
    if (slots < 1) slots = 1;
    const int round = 256, per = (round + slots - 1) / slots;
    note_attn_kernel("attn_seq_bwd_fused_kernel<%d, %s>", nf13 ? 13 : 0, drop ? "true" : "false");
    if (nf13) return drop ? launch_fused<13, true>(a, npad, lds, round, per, ticks, dq_sums, s) : launch_fused<13, false>(a, npad, lds, round, per, ticks, dq_sums, s);
    return drop ? launch_fused<0, true>(a, npad, lds, round, per, ticks, dq_sums, s) : launch_fused<0, false>(a, npad, lds, round, per, ticks, dq_sums, s);
}

}  // namespace sfcvit
