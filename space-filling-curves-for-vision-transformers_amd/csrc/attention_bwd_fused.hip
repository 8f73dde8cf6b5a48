// Fused attention backward for gfx950 (head dim 64, N <= 224: ViT-B/16 @ 224 has N = 196): dK, dV AND dQ of one
// (batch, head) in ONE pass over the scores -- five MFMA products per score tile (S, dP, dV, dK, dQ) instead of the
// seven of the two-kernel form (attention_seq.hip: a dK/dV kernel and a dQ kernel that each recompute S and dP), the
// exp / dropout-hash / dS arithmetic once instead of twice, and q, k, v, dO read from HBM once instead of twice.
//
//   * one workgroup of 16 waves per (batch, head); Q, dO and K of the whole sequence are staged once into LDS by
//     LDS-DMA (3 x 28 KiB at N = 196); every KEY wave (wave w < nf owns the 16 keys 16w .. 16w+15) fetches its V
//     fragments straight into registers, so V never touches LDS (its K fragments are read from the K image where they are
//     used: held in registers for a whole item they were the eight registers the persistent loop did not have).
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
//     tensor by a column-sum pass (dQ's from the key waves: sum_q dQ[q, :] = scale sum_k (sum_q dS[q, k]) K[k, :]).
//   * PERSISTENT with a rolling prefetch (round 4): the grid is one workgroup per CU and a workgroup walks (batch, head)
//     items i, i + grid, ...  While it computes an item it stages the NEXT one behind itself: the Q / dO rows of chunk
//     c - 1 are dead once step c - 1 has passed its barrier, so step c issues the LDS-DMA of the next item's rows into the
//     same place; the next K goes into a second K image (all of it in step 1), and in the last step -- dQ waves only,
//     the key waves are idle -- the key waves fetch the next V fragments, the O pieces for delta, lse and the row keys.
//     Every barrier drains the DMA issued a whole step earlier.  The 117 KiB load burst that opened every workgroup
//     (28 % of the kernel: ~11 B/clock per CU whatever the other CUs do, nothing to overlap it with at one workgroup per
//     CU) is paid once per CU instead of once per item.  A grid of one workgroup per item is the former kernel.
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
constexpr int FUSED_ROW_BYTES = 4 * 128 + 2 * 64 + 6 * 4;    // LDS bytes per padded sequence row: Q, dO, two K images; dS exchange x 2; two sets of lse / delta / row key
constexpr int FUSED_CS_BYTES = 16 * 128 * 4;                   // column-sum staging [FWAVES][dK | dV][64] floats
constexpr int FUSED_POST_BYTES = 8192;                       // own scratch: the [32][64] partial dK / dV of a shared fragment (+ as much again for short sequences, whose K image is too small for the column-sum staging)
constexpr int FUSED_EXTRA = 256 + 64 + 14 * 64 * 4;          // (256 spare) + two item records (3 pointers each, 8-byte slots) + the key waves' shares of the dQ column sums [14][64]

// tools/attn_isa_budget.py compiles this file with -DSFCVIT_ISA_MARKERS and buckets the instructions between the marks
#ifdef SFCVIT_ISA_MARKERS
#define ISA_MARK(name) asm volatile("; ISA_MARK " name ::: "memory")
#else
#define ISA_MARK(name) do { } while (0)
#endif

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
#ifdef SFCVIT_ATTN_TRACE
__device__ unsigned long long g_attn_trace[512];
#define ATRACE(i) do { if (blockIdx.x == 5 && threadIdx.x == 0 && (i) < 512) g_attn_trace[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATRACE(i) do {} while (0)
#endif

// Stage npad rows x 64 cols (128-B rows) into a kc image by LDS-DMA; the bank swizzle goes on the SOURCE chunk
// (the DMA writes lane-linear); rows >= N copy row N - 1.
__device__ __forceinline__ void dma_rows(char *img, const uint16_t *__restrict__ src, int ld, int N, int npad, int tid, int nt = 0) {
    for (int p = tid; p < npad * 8; p += FT) {
        const int row = p >> 3, cs = p & 7;
        const int c = cs ^ kc_swz(row);
        const uint16_t *g = src + size_t(min(row, N - 1)) * ld + c * 8;
        if (nt) __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(img + p * 16), 16, 0, 2);      // nt: read once, do not allocate
        else __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(img + p * 16), 16, 0, 0);
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
__global__ __launch_bounds__(FT) void attn_seq_bwd_fused_kernel(const sfcvit_attn_args a, int npad, int stag_round, int stag_per, int stag_ticks, int dq_sums, int items, int nt, unsigned *__restrict__ qcnt) {

    extern __shared__ __attribute__((aligned(16))) char smem[];
    stagger_start(stag_round, stag_per, stag_ticks);
    char *qimg = smem, *doimg = smem + npad * 128, *kimg0 = smem + 2 * npad * 128, *dsb = smem + 4 * npad * 128;
    char *small0 = dsb + 2 * npad * 64;                               // two sets of [lse | delta | row key], npad words each
    // Item records: the global pointers a step needs (where the NEXT item's Q / K and dO rows come from, where the CURRENT item's
    // dQ goes), written once per item by thread 0 and read from LDS where they are used.  Formed from the item number in
    // scalar registers they cost every one of the 16 waves ~60 scalar instructions per step (an integer division among them)
    // and, kept across the loop, the scalar registers the kernel does not have (106 used, 48 spilled).
    unsigned long long *rec0 = reinterpret_cast<unsigned long long *>(smem + npad * FUSED_ROW_BYTES + 256);   // [2][4]
    float *qpart = reinterpret_cast<float *>(smem + npad * FUSED_ROW_BYTES + 256 + 64);                          // [14 key-side waves][64]
    char *own_scratch = smem + npad * FUSED_ROW_BYTES + FUSED_EXTRA;         // short sequences: a K image is smaller than the scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, D = a.H * HD, ld = 3 * D;
    const int nf = NFC ? NFC : (N + 15) >> 4, nc = NFC ? ((NFC + 1) >> 1) : npad >> 5;
    // Key fragments -> waves.  With nf <= 13 a wave is spare, and the LAST fragment (at N = 196: 4 valid keys of 16, a
    // whole wave's VALU work all the same) is shared by waves nf - 1 and nf, one query fragment of every chunk each: the
    // loop is VALU-bound per SIMD, waves go to SIMDs round-robin, and 13 whole fragments put 4 on one SIMD and 3 on the
    // others -- halves make it 3.5 / 3.5 / 3 / 3.  The two partial dK / dV meet in LDS after the loop.
    const bool split = nf <= FWAVES - 3;
    const int kfi = (split && wave == nf) ? nf - 1 : wave;             // the key fragment this wave works on
    const int tsel = !split ? -1 : wave == nf - 1 ? 0 : wave == nf ? 1 : -1;   // -1: both query fragments of a chunk
    const bool is_key = wave < nf || (split && wave == nf), is_dq = wave >= FWAVES - 2;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float scale = a.scale, c2 = a.scale * 1.4426950408889634f;
    // (lane offsets of the fragment reads are rebuilt inside key_step / dq_step from an opaque lane id: hoisted here they
    // were spilled and reloaded every step)
    const float dsc = 1.f / (1.f - a.dropout_p);
    const int g = lane >> 4, li = lane & 15;
    const int key = 16 * kfi + li;                    // key waves: the key this lane's accumulator columns belong to
    const uint32_t dth32 = dth << 16;

    // what an item needs in registers / the small arrays before its loop: V fragments (key waves), the O pieces that pair
    // with this thread's pieces of the dO image (delta), lse and the dropout row keys.  For the first item this runs in
    // front of the staging, for the others in the last step of the item before (key waves idle, loads under the dQ waves).
    bf16x8 vf[2];
    u32x4 opiece[2];
    auto fetch_item = [&](int it, char *small) __attribute__((always_inline)) {
        const int b = it / a.H, h = it - b * a.H;
        const uint16_t *vp = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD + 2 * D;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) vf[kk] = global_frag(vp, ld, 16 * kfi, is_key ? N : 0, kk, lane);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int p = tid + FT * i, row = p >> 3, c = (p & 7) ^ kc_swz(row);
            opiece[i] = u32x4{0u, 0u, 0u, 0u};
            if (row < N) {
                const u32x4 *op = reinterpret_cast<const u32x4 *>(static_cast<const uint16_t *>(a.out) + (size_t(b) * N + row) * D + h * HD + c * 8);
                opiece[i] = nt ? __builtin_nontemporal_load(op) : *op;
            }
        }
        const float *lse = a.lse + (size_t(b) * a.H + h) * N;
        float *lse_w = reinterpret_cast<float *>(small);
        uint32_t *rkey_w = reinterpret_cast<uint32_t *>(small) + 2 * npad;
        for (int i = tid; i < npad; i += FT) {
            lse_w[i] = i < N ? lse[i] * 1.4426950408889634f : INFINITY;   // padded queries: p = exp2(-inf) = 0
            rkey_w[i] = drop_row_key(eff_seed(a.dropout_seed, a.seed_off), (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(i));
        }
    };
    // delta[q] = sum_c dO[q][c] O[q][c]: 8 products per thread and piece, summed over the 8 lanes that share a row
    auto delta_item = [&](char *small) __attribute__((always_inline)) {
        float *del_w = reinterpret_cast<float *>(small) + npad;
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
            if ((tid & 7) == 0 && p < npad * 8) del_w[row] = part;          // rows >= N: O piece = 0 -> 0
        }
    };
    // One 16-byte piece of an image by LDS-DMA: piece p = (row, swizzled chunk) of [N, ld] source rows, rows >= N copy row
    // N - 1.  Issued from inline asm (the wave's LDS base in M0, lanes lane-linear behind it): hipcc, once it knows of an
    // LDS-DMA in flight, puts `s_waitcnt vmcnt(0)` in front of every transposed LDS read and every use of an ordinary load
    // -- each wave would wait for its prefetch at the top of the step that is supposed to hide it.  The kernel waits
    // itself: vmcnt(0) in front of the barrier that ends the step (dma_wait).
    const uint32_t lds_base = uint32_t(uintptr_t((lptr_t)smem));
    auto dma_piece = [&](const char *img, const uint16_t *src, int sld, int p) __attribute__((always_inline)) {
        const int row = p >> 3, cs = p & 7;
        const int c = cs ^ kc_swz(row);
        const uint16_t *gp = src + size_t(min(row, N - 1)) * sld + c * 8;
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base + uint32_t(img - smem) + uint32_t(p) * 16u);
        if (nt) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(gp), "s"(m0v) : "memory");
        else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(m0v) : "memory");
    };
    auto dma_wait = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    auto write_record = [&](int it, unsigned long long *rec) __attribute__((always_inline)) {     // thread 0 only
        const int b = it / a.H, h = it - b * a.H;
        rec[0] = reinterpret_cast<unsigned long long>(static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD);
        rec[1] = reinterpret_cast<unsigned long long>(static_cast<const uint16_t *>(a.dout) + size_t(b) * N * D + h * HD);
        rec[2] = reinterpret_cast<unsigned long long>(static_cast<uint16_t *>(a.dqkv) + size_t(b) * N * ld + h * HD);
    };
    // Items: the first is blockIdx.x; the others come from a counter (qcnt[0], + gridDim.x) that thread 0 draws ONE ITEM AHEAD
    // of where the number is needed and publishes in the next item's record, so that a workgroup which starts late -- RCCL
    // kernels hold some CUs while gradients are reduced beside backward -- just takes fewer items; with a fixed stride it would
    // run its whole list after the others had finished theirs (the case tools/bench_busy_cus.py makes for the GEMM).  qcnt[1]
    // counts finished workgroups, the last one zeroes both words for the next launch.  qcnt = nullptr: fixed stride.
    int item = blockIdx.x, cur = 0;
    unsigned drawn = 0;                               // thread 0: the draw in flight
    if (qcnt && tid == 0) drawn = atomicAdd(qcnt, 1u);
    {   // ---- the first item: everything staged up front ----
        const int b = item / a.H, h = item - b * a.H;
        const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD;
        fetch_item(item, small0);
        if (tid == 0) write_record(item, rec0);
        dma_rows(qimg, base, ld, N, npad, tid, nt);
        dma_rows(doimg, static_cast<const uint16_t *>(a.dout) + size_t(b) * N * D + h * HD, D, N, npad, tid, nt);
        dma_rows(kimg0, base + D, ld, N, npad, tid, nt);
        {   // exchange rows no key wave writes (keys 16 nf .. npad - 1), both halves of the double buffer; written once
            const int nz = (npad - 16 * nf) * 4;         // 16-byte pieces per half
            for (int i = tid; i < 2 * nz; i += FT) {
                char *dst = dsb + (i >= nz ? npad * 64 : 0) + 16 * nf * 64 + (i >= nz ? i - nz : i) * 16;
                *reinterpret_cast<u32x4 *>(dst) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();                                  // LDS-DMA pending: hipcc drains vmcnt(0) here
        delta_item(small0);
        __syncthreads();
    }

  for (;;) {                                          // ---- items of this workgroup ----
    const int b = item / a.H, h = item - b * a.H;
    int nxt = items;                                  // known to every thread after step 0's barrier (record word 3)
    bool has_next = false;
    char *kimg = kimg0 + cur * npad * 128, *knext = kimg0 + (cur ^ 1) * npad * 128;
    char *small = small0 + cur * 3 * npad * 4, *small_next = small0 + (cur ^ 1) * 3 * npad * 4;
    const float *lse_s = reinterpret_cast<const float *>(small), *del_s = lse_s + npad;
    const uint32_t *rkey_s = reinterpret_cast<const uint32_t *>(small) + 2 * npad;
    unsigned long long *rec = rec0 + cur * 4, *rec_next = rec0 + (cur ^ 1) * 4;
    if (tid == 0) {                                   // read from step 1 on (step 0's barrier in between)
        const int nx = qcnt ? int(gridDim.x + drawn) : item + int(gridDim.x);
        rec_next[3] = (unsigned long long)nx;
        if (nx < items) {
            write_record(nx, rec_next);
            if (qcnt) drawn = atomicAdd(qcnt, 1u);    // for the item after that one: consumed a whole item later
        }
    }
    // (the K fragments of this wave's keys are read from the K image where they are used, twice per step: held in registers
    // for the whole item they were the 8 registers the persistent loop did not have)
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int hf = 0; hf < 4; hf++) dk[hf] = dv[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // sum over the queries of dS[q, key] for this lane's key (its 4 query rows of every fragment): the column sums of dQ are
    // sum_q dQ[q, :] = scale sum_k (sum_q dS[q, k]) K[k, :] -- one multiply-add per score here and a 16 x 64 product per key
    // wave after the loop, instead of 8 cross-lane reductions + 8 LDS read-modify-writes per step on the two dQ waves, which
    // the step waits for (263 -> 2xx us with the in_proj bias gradient, profiles/r4/attention_bench_r4.txt)
    float ksum = 0.f;

    // one step of a key wave: S^T, dP^T of its keys against the 32 queries of chunk c, P and dS, dV += P^T dO, dK += dS^T Q,
    // and its [16 keys][32 queries] block of dS^T into the exchange image
    auto key_step = [&](int c) __attribute__((always_inline)) {
        // Lane constants rebuilt per step behind an empty asm: kept across the loop (and, hoisted, across the items) they and the
        // LDS addresses derived from them held 20-30 registers between them, and the V fragments went to scratch instead.
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int g = ln >> 4, li = ln & 15;
        const int k_off = lane_offsets(ln).k[0], t_off = lane_offsets(ln).t[0];
        const int key = 16 * kfi + li;
        const uint32_t key_g = uint32_t(key) * DROP_G;
        char *slot = dsb + (c & 1) * npad * 64;
        u32x2 pp[2], dd[2];                                       // P and dS of the two query fragments, packed to bf16 as soon as they exist
        // exchange image, writer side: row key, query fragment t at half t ^ ((key >> 2) & 1), 8 bytes at 8 g
        const int ds_w = key * 64 + (((li >> 2) & 1) << 5) + 8 * g;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int qf = 2 * c + t;
            // the other wave of a shared fragment does this one -- or the fragment has no query at all (N = 196: rows
            // 208 .. 223 of the last chunk; 1 / 14 of the loop's score work)
            if (tsel == 1 - t || 16 * qf >= N) {
                pp[t] = dd[t] = u32x2{0u, 0u};
                continue;
            }
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(qimg, 16 * qf, (k_off ^ (kk << 6))), kc_frag_at(kimg, 16 * kfi, (k_off ^ (kk << 6))), s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(doimg, 16 * qf, (k_off ^ (kk << 6))), vf[kk], dp, 0, 0, 0);
            }
            const int ql0 = 16 * qf + 4 * g;                     // this lane's 4 queries: one 16-B LDS read each
            const f32x4 lse4 = *reinterpret_cast<const f32x4 *>(lse_s + ql0);    // lse * log2(e)
            const f32x4 del4 = *reinterpret_cast<const f32x4 *>(del_s + ql0);
            const u32x4 rk4 = *reinterpret_cast<const u32x4 *>(rkey_s + ql0);
            float pr[4], dr[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float pv = fast_exp2(s[r] * c2 - lse4[r]);
                float keep = 1.f;                                 // 1 / (1 - p) where the element is kept, else 0
                if (DROP)     // element (query row, key): xor with the lane's key constant, one multiply, one compare
                    keep = drop_hash_g(rk4[r], key_g) >= dth32 ? dsc : 0.f;
                pr[r] = pv * keep;
                dr[r] = pv * (dp[r] * keep - del4[r]);                // x scale at the stores of dK and dQ
            }
            if (16 * kfi + 16 > N) {                             // boundary fragment (wave-uniform): keys >= N carry no gradient
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (key >= N) dr[r] = 0.f;
            }
            ksum += (dr[0] + dr[1]) + (dr[2] + dr[3]);
            pp[t] = u32x2{pack2bf(pr[0], pr[1]), pack2bf(pr[2], pr[3])};
            dd[t] = u32x2{pack2bf(dr[0], dr[1]), pack2bf(dr[2], dr[3])};
        }
        const bf16x8 pf = __builtin_bit_cast(bf16x8, u32x4{pp[0][0], pp[0][1], pp[1][0], pp[1][1]});
        const bf16x8 dsf = __builtin_bit_cast(bf16x8, u32x4{dd[0][0], dd[0][1], dd[1][0], dd[1][1]});
        // dS^T for the dQ waves: 4 consecutive queries of fragment t = 8 bytes
        if (tsel != 1) *reinterpret_cast<u32x2 *>(slot + ds_w) = dd[0];
        if (tsel != 0) *reinterpret_cast<u32x2 *>(slot + (ds_w ^ 32)) = dd[1];
#pragma unroll
        for (int hf = 0; hf < 4; hf++) {
            dv[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(doimg, 32 * c, (t_off ^ (hf << 5))), pf, dv[hf], 0, 0, 0);
            dk[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(qimg, 32 * c, (t_off ^ (hf << 5))), dsf, dk[hf], 0, 0, 0);
        }
    };
    // one step of a dQ wave: dQ^T = K^T dS^T for the 32 queries of chunk c - 1 from the exchange image the key waves filled one step ago
    auto dq_step = [&](int c) __attribute__((always_inline)) {
        int ln = lane;                                    // lane constants and the store addresses built from them: rebuilt per step (see key_step)
        asm volatile("" : "+v"(ln));
        const int g = ln >> 4, li = ln & 15;
        uint16_t *dq_base = reinterpret_cast<uint16_t *>(rec[2]);
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
        if (32 * (c - 1) + 16 < N) {
#pragma unroll
            for (int ks = 0; ks < FMAXC; ks++) {
                if (ks < nc) {                                   // key steps of 32 (keys are padded like the queries)
                    const bf16x8 b0 = ds_tr_frag(slot, 32 * ks, ds_r), b1 = ds_tr_frag(slot, 32 * ks, ds_r ^ 32);
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const bf16x8 kt = tr_frag_at(kimg, 32 * ks, kt_off[hh]);
                        acc[hh][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, b0, acc[hh][0], 0, 0, 0);
                        acc[hh][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, b1, acc[hh][1], 0, 0, 0);
                    }
                }
            }
        } else {                                                 // the chunk's second query fragment is past N (N = 196: the last chunk): half the step
#pragma unroll
            for (int ks = 0; ks < FMAXC; ks++) {
                if (ks < nc) {
                    const bf16x8 b0 = ds_tr_frag(slot, 32 * ks, ds_r);
#pragma unroll
                    for (int hh = 0; hh < 2; hh++)
                        acc[hh][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(kimg, 32 * ks, kt_off[hh]), b0, acc[hh][0], 0, 0, 0);
                }
            }
        }
        mfma_fence();
        // acc[hh][qf][r] = dQ[query 32 (c-1) + 16 qf + li][head column 32 dqw + 16 hh + 4 g + r]
        // one 16-byte store per lane and query fragment (the pieces of hh = 0 / 1 exchanged as in store_rows): 16 rows x 64 bytes
#pragma unroll
        for (int qf = 0; qf < 2; qf++) {
            const int q = 32 * (c - 1) + 16 * qf + li;
            const auto lo = __builtin_amdgcn_permlane16_swap(pack2bf(acc[0][qf][0] * scale, acc[0][qf][1] * scale),
                                                             pack2bf(acc[1][qf][0] * scale, acc[1][qf][1] * scale), false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap(pack2bf(acc[0][qf][2] * scale, acc[0][qf][3] * scale),
                                                             pack2bf(acc[1][qf][2] * scale, acc[1][qf][3] * scale), false, false);
            const u32x4 o = {lo[0], hi[0], lo[1], hi[1]};
            if (q < N) *reinterpret_cast<u32x4 *>(dq_base + size_t(q) * ld + 32 * dqw + 16 * (g & 1) + 8 * (g >> 1)) = o;
        }
    };
    // Rolling prefetch of step c >= 1: the Q / dO rows of chunk c - 1 (free since the last barrier), in step 1 also the whole
    // next K image.  Source pointers come from the item record in LDS (written at the top of the item, one barrier ago); the
    // thread index is made opaque so that the address arithmetic is rebuilt here: hoisted out of the loop it had no
    // registers and was reloaded from scratch on the path of every step (measured: 240 -> 340 us).
    auto prefetch = [&](int c) __attribute__((always_inline)) {
        int te = tid;
        asm volatile("" : "+v"(te));
        const uint16_t *nq = reinterpret_cast<const uint16_t *>(rec_next[0]);
        if (c == 1) {
            // the whole next K image, by the 14 waves that are not dQ waves: a dQ wave has stores in flight at every barrier and
            // must not have to wait for vmcnt(0) there
            if (te < FT - 128)
                for (int p = te; p < npad * 8; p += FT - 128) dma_piece(knext, nq + D, ld, p);
        }
        if (te < 512) {
            const int p = (c - 1) * 256 + (te & 255);
            if (te < 256) dma_piece(qimg, nq, ld, p);
            else dma_piece(doimg, reinterpret_cast<const uint16_t *>(rec_next[1]), D, p);
        }
    };
    [[maybe_unused]] int tr_i = ((item - int(blockIdx.x)) / int(gridDim.x)) * 16;
    ATRACE(tr_i);
    for (int c = 0; c < nc; c++) {
        if (has_next && c >= 1) prefetch(c);
        if (is_key) {
            ISA_MARK("key_step begin");
            key_step(c);
            ISA_MARK("key_step end");
        } else if (is_dq && c >= 1) {
            ISA_MARK("dq_step begin");
            dq_step(c);
            ISA_MARK("dq_step end");
        }
        if (!is_dq) dma_wait();                       // this step's DMA has landed (a key wave has no other vector-memory operation in flight)
        __syncthreads();
        if (c == 0) {
            nxt = __builtin_amdgcn_readfirstlane(int(rec_next[3]));
            has_next = nxt < items;
        }
        ATRACE(tr_i + 1 + c);
    }
    // The last step: dQ of the last chunk by the two dQ waves.  The key waves are idle and fetch what the next item needs in
    // registers; the wave that shares the last fragment hands over its half.  (Storing dK / dV here as well, under the dQ
    // waves, was measured: the same 40 000 clocks per item -- the 52 store instructions of an item are bound by the CU's
    // address path, ~4 500 clocks wherever they stand -- and 13 us MORE with the column sums behind them.)
    if (has_next) prefetch(nc);
    if (is_dq) dq_step(nc);                           // (before the fetch: a dQ wave's stores are then the OLDEST operations in its queue)
    if (has_next) fetch_item(nxt, small_next);
    ATRACE(tr_i + 8);
    if (a.colsum_part && dq_sums && is_key) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int g = ln >> 4, li = ln & 15, key = 16 * kfi + li;
        float sk = ksum;                                                  // lanes li, li + 16, li + 32, li + 48 hold the 4 quarters of key li
        {
            const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(sk), __float_as_uint(sk), false, false);
            sk = __uint_as_float(h[0]) + __uint_as_float(h[1]);
            const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(sk), __float_as_uint(sk), false, false);
            sk = __uint_as_float(q[0]) + __uint_as_float(q[1]);
        }
        if (key >= N) sk = 0.f;                                           // (rows of the K image past N repeat row N - 1)
#pragma unroll
        for (int hc = 0; hc < 2; hc++) {
            const u32x4 kv = *reinterpret_cast<const u32x4 *>(kimg + kc_off(key, 2 * g + hc));
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float q0 = row16_sum(sk * bf2f(uint16_t(kv[e]))), q1 = row16_sum(sk * bf2f(uint16_t(kv[e] >> 16)));
                if (li == 0) {                                            // this fragment's share, columns 16 g + 8 hc + 2 e (+ 1); read after two barriers
                    qpart[wave * 64 + 16 * g + 8 * hc + 2 * e] = q0 * scale;
                    qpart[wave * 64 + 16 * g + 8 * hc + 2 * e + 1] = q1 * scale;
                }
            }
        }
    }
    float *px = reinterpret_cast<float *>(own_scratch);               // [32][64] lane-private words: partial dK / dV of a shared fragment
    if (split && wave == nf) {                        // the shared fragment: wave nf hands its partial dK / dV to wave nf - 1
        mfma_fence();
#pragma unroll
        for (int hf = 0; hf < 4; hf++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                px[(4 * hf + r) * 64 + lane] = dk[hf][r];
                px[(16 + 4 * hf + r) * 64 + lane] = dv[hf][r];
            }
    }
    // Everything fetched is waited for HERE, tied to the registers: behind this point the key waves store dK / dV, and a wait
    // placed later (the compiler's own, at the first use of the V fragments or the O pieces) would wait for those stores to
    // be acknowledged as well -- 6 000 clocks per item in the first version of this kernel.
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(vf[0]), "+v"(vf[1]), "+v"(opiece[0]), "+v"(opiece[1]) :: "memory");
    __syncthreads();
    ATRACE(tr_i + 9);
#ifdef SFCVIT_ATTN_TRACE
    if (blockIdx.x == 5 && (threadIdx.x & 63) == 0 && tr_i == 16) g_attn_trace[288 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime();
#endif
    if (split && wave == nf - 1) {
        mfma_fence();
#pragma unroll
        for (int hf = 0; hf < 4; hf++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                dk[hf][r] += px[(4 * hf + r) * 64 + lane];
                dv[hf][r] += px[(16 + 4 * hf + r) * 64 + lane];
            }
    }
    const bool owns = wave < nf;                      // the wave that holds a fragment's complete dK / dV
    if (owns) {
        mfma_fence();
        uint16_t *dbase = reinterpret_cast<uint16_t *>(rec[2]);
        store_rows(dbase + D, ld, key, key < N, dk, scale, lane);
        store_rows(dbase + 2 * D, ld, key, key < N, dv, 1.f, lane);
    }
    if (a.colsum_part) {
        // column sums of this (batch, head)'s dQ | dK | dV block: lanes of a 16-lane row hold the 16 rows of a column
        // (DPP reduction), the 13 key waves / 2 dQ waves meet in LDS (the exchange image is free now) and are summed in
        // a fixed order: bit-reproducible.  Layout of the partials: [batch][q | k | v thirds of 3 D], summed over
        // the batch afterwards (launch_colsum_reduce).
        float *cs = reinterpret_cast<float *>(npad * 128 >= FUSED_CS_BYTES ? kimg : own_scratch + FUSED_POST_BYTES);   // [FWAVES][128]: the item's K image is dead by now
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
            if (third == 0) {
                if (dq_sums)
                    for (int w = 0; w < nf + (split ? 1 : 0); w++) s += qpart[w * 64 + c];
            } else {
                for (int w = 0; w < nf; w++) s += cs[w * 128 + (third - 1) * 64 + c];
            }
            if (third || dq_sums) a.colsum_part[size_t(b) * 3 * D + third * D + h * HD + c] = s;
        }
    }
    ATRACE(tr_i + 10);
    if (!has_next) break;
    // ---- item switch: the next item's images are resident (the last step's barrier drained their DMA) ----
#ifdef SFCVIT_ATTN_TRACE
    if (blockIdx.x == 5 && (threadIdx.x & 63) == 0 && tr_i == 16) g_attn_trace[256 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();                                  // scratch has been read
    ATRACE(tr_i + 12);
    delta_item(small_next);
    ATRACE(tr_i + 13);
    __syncthreads();
    ATRACE(tr_i + 11);
    item = nxt;
    cur ^= 1;
  }
    if (qcnt && tid == 0) {
        const unsigned old = atomicAdd(qcnt + 1, 1u);
        if (old == gridDim.x - 1u) { qcnt[0] = 0u; qcnt[1] = 0u; }
    }
}

constexpr int FUSED_MAX_N = 32 * FMAXC;
constexpr int FUSED_MAX_LDS = FUSED_MAX_N * FUSED_ROW_BYTES + FUSED_EXTRA + FUSED_POST_BYTES;

template <int NFC, bool DROP>
int launch_fused(const sfcvit_attn_args &a, int npad, size_t lds, int grid, int round, int per, int ticks, int dq_sums, hipStream_t s) {
    static const int nt = [] { const char *e = getenv("SFCVIT_ATTN_NT"); return e ? atoi(e) : 0; }();
    static const int fixed = [] { const char *e = getenv("SFCVIT_ATTN_BWD_QUEUE"); return e && e[0] == '0'; }();     // "0": fixed stride (A/B)
    unsigned *qcnt = nullptr;
    if (!fixed && grid < a.B * a.H) {
        unsigned *slot = stream_counters(s);
        if (slot) qcnt = slot + 12;
    }
    if (int rc = raise_lds_limit(reinterpret_cast<const void *>(&attn_seq_bwd_fused_kernel<NFC, DROP>), FUSED_MAX_LDS, "attention_bwd_fused attribute")) return rc;
    hipLaunchKernelGGL((attn_seq_bwd_fused_kernel<NFC, DROP>), dim3(grid), dim3(FT), lds, s, a, npad, round, per, ticks, dq_sums, a.B * a.H, nt & 1, qcnt);
    return check_launch("attention_bwd_fused");
}

}  // namespace

// -1: not eligible (the caller falls back to the two-kernel form); else a status.
int attn_seq_bwd_fused(const sfcvit_attn_args &a, int dq_sums, hipStream_t s) {
    if (a.hd != HD || a.N > FUSED_MAX_N) return -1;
    const int npad = (a.N + 31) / 32 * 32;
    const bool nf13 = (a.N + 15) / 16 == 13, drop = a.dropout_p > 0.f;
    const size_t lds = size_t(npad) * FUSED_ROW_BYTES + FUSED_EXTRA + FUSED_POST_BYTES + (npad * 128 >= FUSED_CS_BYTES ? 0 : FUSED_CS_BYTES);
    // One workgroup per CU walking the (batch, head) items with the next one staged behind the current (kernel header);
    // SFCVIT_ATTN_BWD_PERSIST=0: one workgroup per item, i.e. the kernel of rounds 2-3 (A/B).
    const int items = a.B * a.H, cus = device_cu_count();
    const char *pe = getenv("SFCVIT_ATTN_BWD_PERSIST");
    const int grid = (cus > 0 && !(pe && pe[0] == '0')) ? std::min(items, cus) : items;
    // Start-up stagger (attention_common.h): every workgroup opens with a 117 KiB load burst and they all take the same time,
    // so launched together they stay in lockstep.  Two slots 4.5 us apart: 270.8 -> 257.1 us at ViT-B / 256 with one workgroup
    // per item (3 or 4 slots, 2-8 us: 255.6-258.6).  SFCVIT_ATTN_STAGGER_BWD = "slots,ticks" (10 ns) overrides; "1,0" = off.
    // The persistent form (round 4) pays the burst once per 12 items and its workgroups drift apart on their own: the stagger
    // costs it 5 us (227.6 vs 222.7 us, profiles/r4/attention_bench_r4.txt), so it is on for one-workgroup-per-item launches only.
    int slots = grid < items ? 1 : 2, ticks = 450;
    if (const char *e = getenv("SFCVIT_ATTN_STAGGER_BWD")) sscanf(e, "%d,%d", &slots, &ticks);
    if (slots < 1) slots = 1;
    const int round = 256, per = (round + slots - 1) / slots;
    note_attn_kernel("attn_seq_bwd_fused_kernel<%d, %s>", nf13 ? 13 : 0, drop ? "true" : "false");
    if (nf13) return drop ? launch_fused<13, true>(a, npad, lds, grid, round, per, ticks, dq_sums, s) : launch_fused<13, false>(a, npad, lds, grid, round, per, ticks, dq_sums, s);
    return drop ? launch_fused<0, true>(a, npad, lds, grid, round, per, ticks, dq_sums, s) : launch_fused<0, false>(a, npad, lds, grid, round, per, ticks, dq_sums, s);
}

#ifdef SFCVIT_ATTN_TRACE
extern "C" int sfcvit_debug_attn_trace(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_trace), sizeof(unsigned long long) * 512) == hipSuccess ? 0 : 1;
}
#endif
}  // namespace sfcvit
