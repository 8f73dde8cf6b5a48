// Persistent 8-phase bf16 MFMA GEMM for gfx950: C[M,N] = epilogue(A[M,K] B[N,K]^T), both operands
// k-contiguous -- every forward GEMM of the encoder and (on the per-step transposed weight) every dX GEMM.
//
//   * tile (32*NI) x 256 x 64 with NI = 8, 7 or 6 (256, 224 or 192 rows: the height that wastes least of the last
//     round of tiles on 256 CUs; M = 50176 = 196 * 256 = 224 * 224, ViT-L's 36864 = 192 * 192; any M >= one tile: the last
//     row tile of a height that does not divide M overlaps the one before it), 8 waves = 2 wave groups (row halves) x 4 column
//     blocks, wave tile (16*NI) x 64, 16x16x32 MFMA.
//   * one workgroup per CU walks output tiles that it draws from a per-XCD atomic counter (tiles are dealt to the XCDs
//     in chunks of 32 consecutive ones, so the workgroups of an XCD share A row panels in its L2).  Drawing instead of
//     a fixed round-robin costs one atomic per tile and makes the kernel insensitive to CUs that are busy with
//     something else when it starts -- an RCCL all-reduce overlapped with backward occupies some: with fixed lists
//     the workgroups that must wait for those CUs start a whole kernel late and double its duration.  Which workgroup
//     computes a tile does not change the tile's result.  The k-tile stream does not stop
//     at a tile boundary: operands go HBM/L2 -> LDS by LDS-DMA into a ring of two 64 KiB k-tiles (four 16 KiB
//     half-tiles each: A rows of group 0 / group 1, B columns 0-127 / 128-255) and the ring keeps prefetching the
//     next tile's first k-tiles under the current tile's last phases and its epilogue -- no prologue per tile.
//   * a k-tile is 4 phases; a phase = {ds_read the fragments of one quadrant, issue the LDS-DMA of one half-tile,
//     barrier, 16 MFMAs, barrier}.  The two wave groups run one barrier apart, so one group's MFMAs cover the other
//     group's LDS reads and DMA issue (cdna_hip_programming.md §5, "The 256^2 8-phase template").  The four
//     half-tiles of a k-tile are cut so that each is read in exactly ONE phase -- A0 / A1 = the first 64 / the
//     remaining rows of BOTH wave groups, B0 / B1 = the first / second 32 columns of ALL four column blocks:
//         phase   reads (LDS)      MFMA (rows x cols)   LDS-DMA issued          counted wait before the barrier
//           0     A0, B0           a0 x b0              -                       -
//           1     A1               a1 x b0              -                       -
//           2     B1               a1 x b1              B0 of k-tile g+2        -
//           3     -                a0 x b1              A0 of k-tile g+2        vmcnt(4): all of k-tile g+1
//          end    -                -                    A1, B1 of k-tile g+2    -
//     ("end" = after the phase-3 MFMAs; at the last k-tile of a tile that is before the epilogue, so that everything
//     the next tile's first wait needs is older than the epilogue's stores: that wait is vmcnt(4 + stores).)
//     Every half-tile is overwritten at least two phases after its only read, so no early retirement of LDS reads is
//     needed; four to six half-tiles (64-96 KiB per CU) are in flight at any time; a buffer is read in the phases
//     after the wait that retires it.
//   * LDS image of a half-tile: [128 rows][64 k] bf16, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7)
//     (ds_read_b128 conflict-free, tools/lds_conflicts.py); LDS-DMA writes are lane-linear, so the swizzle is
//     applied to the per-lane SOURCE address.  B rows are stored in fragment order: LDS row 16j + l holds column
//     16(l >> 2) + 4j + (l & 3) of the wave's 64, which makes the accumulators of a lane 16 CONSECUTIVE columns of
//     one row -- the epilogue reads residual / aux as 16-byte vectors into the fragment layout; the packed result
//     rows then pass through a wave-private LDS patch so that every store instruction is row-contiguous.
//   * epilogue variants are compile-time (MASK) so that the fully unrolled per-row code stays small (DESIGN.md §5:
//     the instruction cache punished a runtime option tree).  Wave group 1 runs its epilogue before, group 0 after
//     the tile's last barrier: both run concurrently instead of one after the other.
//
// Anything not eligible (k-major operands, split-K, GELU / aux_out / fp32 output, shapes off the tile grid) returns
// -1 from gemm8p_dispatch and takes the older kernels (gemm256.hip, gemm.hip).
#include "common_host.h"
#include "gemm_core.h"
#include <mutex>
#include <type_traits>
#include <unordered_map>

namespace sfcvit {
namespace {

using namespace gemm_core;

namespace p8 {

constexpr int T = 512, HALF = 16384, KTB = 65536, LDS_BYTES = 2 * KTB;
// gemm8p_kernel's LDS beyond the ring: the 4-entry tile ring, one [16][128 B] store patch per wave, the bias vector
constexpr int LDS_TQ = LDS_BYTES, LDS_PATCH = LDS_TQ + 64, LDS_BIAS = LDS_PATCH + 8 * 2048, LDS_MAX = 160 * 1024;
constexpr int BIAS_MAX_N = (LDS_MAX - LDS_BIAS) / 2;
enum { RELU = 1, DROP = 2, RES = 4, DACT = 8, CSUM = 16, BITS = 32 };   // BITS: sfcvit_gemm_args.actmask written (RELU) / read (DACT)

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

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

struct Cursor {            // the k-tile being staged for one operand: index in the tile list, k offset, origin
    int tile, k0;
    const uint16_t *p;
};

// One output row segment of 16 consecutive columns, in the documented order: bias, act, dropout, residual, dact, store.
// `side` = the 16 bf16 of this row segment of the residual (RES) or of aux_in (DACT), loaded by the caller for ALL
// rows before the first use: hipcc puts one `s_waitcnt vmcnt(0)` in front of the first use of an ordinary load while
// LDS-DMA is in flight, so loading row by row paid one memory latency per row.
// Where a lane's packed output goes: the wave's [16 rows][64 columns] of one fragment row pass through a wave-private
// LDS patch ([16][128 B], 16-byte chunk index XOR-ed with row & 7) so that the global stores are row-contiguous.
struct StoreMap {
    int wa;          // LDS address of this lane's first 16 bytes in the fragment layout (row nl, chunk 2 q); second: wa ^ 16
    int ra;          // LDS address read back: row lane >> 3 (and + 8 at offset 1024), chunk lane & 7
    long coff;       // element offset of (row lane >> 3, column 8 (lane & 7)) relative to (row nl, column 16 q)
};

template <int MASK>
__device__ __forceinline__ void epilogue_math(const sfcvit_gemm_args &g, int m, int n, float (&v)[16], const float (&bv)[16],
                                              const u32x4 (&side)[2], uint32_t thresh, float keep_scale, float dact_scale,
                                              uint32_t rk_in0, uint32_t rk_in1, u32x4 &w0, u32x4 &w1) {
    if (!(MASK & DACT)) {                                       // gradient GEMMs carry no bias (the dispatcher checks)
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] += bv[r];
    }
    if (MASK & RELU) {
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = fmaxf(v[r], 0.f);
    }
    if (MASK & DROP) {
        // drop_row_key(seed, row) with its inner hash (a function of the seed and of row >> 32, which is 0 or 1 here:
        // m < 2^31, row_offset < 2^32) taken from the two wave-uniform values worked out once per kernel
        const uint64_t row = uint64_t(m) + uint64_t(uint32_t(g.row_offset));
        const uint32_t rk = mix32(uint32_t(row) * 0x9E3779B1u + ((row >> 32) ? rk_in1 : rk_in0));
#pragma unroll
        for (int p = 0; p < 8; p++) {
            bool k0, k1;
            drop_keep2(rk, uint32_t(n >> 1) + p, thresh, k0, k1);
            v[2 * p] = k0 ? v[2 * p] * keep_scale : 0.f;
            v[2 * p + 1] = k1 ? v[2 * p + 1] * keep_scale : 0.f;
        }
    }
    if (MASK & RES) {
        float rv[16];
        unpack8f(side[0], rv);
        unpack8f(side[1], rv + 8);
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] += rv[r];
    }
    if ((MASK & BITS) && (MASK & RELU)) {                       // the sign pattern of what is stored (no RES with RELU: dispatcher)
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) bits |= uint32_t(v[r] > 0.f) << r;
        uint8_t *mp = static_cast<uint8_t *>(g.actmask) + size_t(m) * g.ld_actmask + (n >> 3);
        asm volatile("global_store_short %0, %1, off\n\ts_nop 1" ::"v"(mp), "v"(bits) : "memory");
    }
    if (MASK & DACT) {
        if (MASK & BITS) {                                      // side[0][0] = the 16 mask bits of this row segment
            const uint32_t bits = side[0][0];
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = (bits >> r & 1u) ? v[r] * dact_scale : 0.f;
        } else {
            float a[16];
            unpack8f(side[0], a);
            unpack8f(side[1], a + 8);
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = a[r] > 0.f ? v[r] * dact_scale : 0.f;
        }
    }
    w0 = u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
    w1 = u32x4{pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15])};
}

// Fragment layout -> row-contiguous layout for NR packed rows of a wave, through the wave-private LDS patch.  In the
// fragment layout a store instruction touches 64 different 64-byte segments (16 rows per quarter-wave, 16 bytes each): the
// vector-memory pipe takes them one by one, and the next tile's LDS-DMA queues behind them; through the patch an instruction
// writes 8 rows x 128 contiguous bytes.  ONE asm statement for all rows, one wait at its end: LDS operations of a wave
// execute in order, so row i + 1's writes may follow row i's reads without a wait in between (the patch is reused row after
// row; the reads return into the registers the writes took their data from, which the LDS pipe has consumed by then), and
// the ~200-clock round trip is paid once per batch -- round 2 waited after every row, 7-8 exposed round trips per epilogue.
// Inline asm because hipcc would put `s_waitcnt vmcnt(0)` in front of ordinary LDS accesses while LDS-DMA is in flight.
#define SFCVIT_XROW(a, b) "ds_write_b128 %[wa], " a "\n\tds_write_b128 %[wb], " b "\n\tds_read_b128 " a ", %[ra]\n\tds_read_b128 " b ", %[ra] offset:1024\n\t"
template <int NR>
__device__ __forceinline__ void patch_exchange(u32x4 (&w)[NR][2], const StoreMap &sm) {
    static_assert(NR >= 1 && NR <= 8, "rows per batch");
    const int wa = sm.wa, wb = sm.wa ^ 16, ra = sm.ra;
    if constexpr (NR == 8)
        asm volatile(SFCVIT_XROW("%0", "%1") SFCVIT_XROW("%2", "%3") SFCVIT_XROW("%4", "%5") SFCVIT_XROW("%6", "%7")
                     SFCVIT_XROW("%8", "%9") SFCVIT_XROW("%10", "%11") SFCVIT_XROW("%12", "%13") SFCVIT_XROW("%14", "%15") "s_waitcnt lgkmcnt(0)"
                     : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[2][0]), "+v"(w[2][1]), "+v"(w[3][0]), "+v"(w[3][1]),
                       "+v"(w[4][0]), "+v"(w[4][1]), "+v"(w[5][0]), "+v"(w[5][1]), "+v"(w[6][0]), "+v"(w[6][1]), "+v"(w[7][0]), "+v"(w[7][1])
                     : [wa] "v"(wa), [wb] "v"(wb), [ra] "v"(ra) : "memory");
    else if constexpr (NR == 7)
        asm volatile(SFCVIT_XROW("%0", "%1") SFCVIT_XROW("%2", "%3") SFCVIT_XROW("%4", "%5") SFCVIT_XROW("%6", "%7")
                     SFCVIT_XROW("%8", "%9") SFCVIT_XROW("%10", "%11") SFCVIT_XROW("%12", "%13") "s_waitcnt lgkmcnt(0)"
                     : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[2][0]), "+v"(w[2][1]), "+v"(w[3][0]), "+v"(w[3][1]),
                       "+v"(w[4][0]), "+v"(w[4][1]), "+v"(w[5][0]), "+v"(w[5][1]), "+v"(w[6][0]), "+v"(w[6][1])
                     : [wa] "v"(wa), [wb] "v"(wb), [ra] "v"(ra) : "memory");
    else if constexpr (NR == 6)
        asm volatile(SFCVIT_XROW("%0", "%1") SFCVIT_XROW("%2", "%3") SFCVIT_XROW("%4", "%5") SFCVIT_XROW("%6", "%7")
                     SFCVIT_XROW("%8", "%9") SFCVIT_XROW("%10", "%11") "s_waitcnt lgkmcnt(0)"
                     : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[2][0]), "+v"(w[2][1]), "+v"(w[3][0]), "+v"(w[3][1]),
                       "+v"(w[4][0]), "+v"(w[4][1]), "+v"(w[5][0]), "+v"(w[5][1])
                     : [wa] "v"(wa), [wb] "v"(wb), [ra] "v"(ra) : "memory");
    else if constexpr (NR == 4)
        asm volatile(SFCVIT_XROW("%0", "%1") SFCVIT_XROW("%2", "%3") SFCVIT_XROW("%4", "%5") SFCVIT_XROW("%6", "%7") "s_waitcnt lgkmcnt(0)"
                     : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[2][0]), "+v"(w[2][1]), "+v"(w[3][0]), "+v"(w[3][1])
                     : [wa] "v"(wa), [wb] "v"(wb), [ra] "v"(ra) : "memory");
    else
        static_assert(NR == 8, "patch_exchange: batch sizes 4, 6, 7, 8");
}
#undef SFCVIT_XROW

// Streaming stores (system scope + non-temporal) of one exchanged row pair: C is far larger than the L2 and is next read
// by another kernel; written through, it does not push the B panel and the A rows the other workgroups are loading out of
// the L2 (measured with the cache-policy bits one by one, same process: plain 181 / 246 / 227 us for QKV / FFN1 / FFN2
// forward, `nt` 170 / 226 / 219, `sc0 sc1 nt` 170 / 219 / 215).  From asm because the builtin only has `nt`; s_nop: the
// store-data hazard (a VALU write of these registers right behind a wide store) is hipcc's to pad only for its own stores.
__device__ __forceinline__ void store_row_pair(const sfcvit_gemm_args &g, int m, int n, const StoreMap &sm, const u32x4 &w0, const u32x4 &w1) {
    uint16_t *c = static_cast<uint16_t *>(g.c) + size_t(m) * g.ldc + n + sm.coff;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(c), "v"(w0) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(c + size_t(8) * g.ldc), "v"(w1) : "memory");
}


// P2 = the two-phase schedule of a k-tile (see ktile2 below): 4 barriers per k-tile instead of 8.
template <int NI, int MASK, bool P2 = false>
__global__ __launch_bounds__(T) void gemm8p_kernel(const sfcvit_gemm_args g, unsigned *__restrict__ counters, int stag_slots, int stag_ticks, int walk) {
    constexpr int BM = 32 * NI, GR = 16 * NI;               // tile rows, rows per wave group
    extern __shared__ __attribute__((aligned(16))) char smem[];   // ONE array: ring of 2 x [A0 | A1 | B0 | B1]
    if (stag_ticks > 0) {   // start-up stagger (see launch()): the workgroups of an XCD start in `slots` groups, `ticks` x 10 ns apart
        const uint64_t until = __builtin_amdgcn_s_memrealtime() + uint64_t((blockIdx.x >> 3) % stag_slots) * uint64_t(stag_ticks);
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
    }
    // wid through readfirstlane: the compiler then knows that wr / wc (and the branches on them) are wave-uniform, and
    // keeps what those branches update -- the staging cursors -- in scalar registers
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wid >> 2, wc = wid & 3;
    const int q = lane >> 4, nl = lane & 15;
    // M need not be a multiple of the tile height: the LAST row tile then starts at row M - BM, i.e. overlaps the one before
    // it, and recomputes (bit-identically: same k order, masks are functions of the global row) and re-stores the rows
    // they share -- no predication anywhere; only the column sums have to leave the shared rows out (epilogue).
    const int K = g.K, NT = g.N / 256, ntiles = ((g.M + BM - 1) / BM) * NT, KT = K / 64, m_last = g.M - BM;
    const uint16_t *A = static_cast<const uint16_t *>(g.a);
    const uint16_t *B = static_cast<const uint16_t *>(g.b);
    // Tile queue.  XCD x (blockIdx % 8: where the dispatcher puts this workgroup, a locality heuristic only) owns the
    // tiles  (l / 32) * G + 32 x + l % 32  for l = 0, 1, ...; counters[x] is the next l.  A workgroup holds two tiles:
    // the one it computes and the next one (whose first k-tiles the ring prefetches); thread 0 draws the one after
    // that one epilogue ahead and publishes it in the next epilogue through a 4-entry ring in LDS (-1 = none left),
    // which every wave reads at least two k-tiles (K >= 256) and many barriers later.  counters[8] counts finished workgroups; the last one zeroes the counters for the next launch.
    const int G = gridDim.x, per_xcd = G >> 3, xcd = blockIdx.x & 7;
    int *const tq = reinterpret_cast<int *>(smem + LDS_TQ);
    auto tile_of = [&](unsigned l) __attribute__((always_inline)) {
        const long t = long(l / unsigned(per_xcd)) * G + xcd * per_xcd + int(l % unsigned(per_xcd));
        // published as (row tile << 16) | column tile: the divisions by NT are done here, by thread 0 off the critical
        // path, not by every wave when its staging cursor crosses into the tile (measured: ~1000 clocks per tile)
        if (t >= ntiles) return -1;
        if (walk > 0 && walk < NT) {
            // column windows of `walk` tiles: the 32 tiles an XCD works on at a time are then ~32 / walk row tiles x walk column
            // tiles instead of 32 / NT x NT -- fewer distinct operand panels per chunk (SFCVIT_GEMM_WALK, A/B)
            const unsigned MT = unsigned(ntiles / NT), full = unsigned(NT / walk) * MT * unsigned(walk), ut = unsigned(t);
            const unsigned wcur = ut < full ? unsigned(walk) : unsigned(NT % walk), base = ut < full ? (ut / (MT * walk)) * walk : unsigned(NT / walk) * walk;
            const unsigned tt = ut < full ? ut % (MT * walk) : ut - full;
            return int((tt / wcur) << 16 | (base + tt % wcur));
        }
        return int((unsigned(t) / unsigned(NT)) << 16 | (unsigned(t) % unsigned(NT)));
    };
    auto finish = [&]() __attribute__((always_inline)) {
        if (tid == 0) {
            const unsigned old = atomicAdd(counters + 8, 1u);
            if (old == unsigned(G) - 1u) {
#pragma unroll
                for (int i = 0; i < 9; i++) counters[i] = 0u;
            }
        }
    };
    if (tid == 0) {
        // two separate draws: every workgroup of the XCD first takes one tile of the first chunk of 32, then one of the
        // second.  (One draw of two would hand a workgroup two column tiles of the same A panel to compute one after
        // the other -- the panel is out of L2 by then: +40 % HBM-side traffic on the N = 768 GEMMs, measured.)
        const unsigned l0 = atomicAdd(counters + xcd, 1u);
        tq[0] = tile_of(l0);
        const unsigned l1 = atomicAdd(counters + xcd, 1u);
        tq[1] = tile_of(l1);
    }
    // the bias vector, once, into LDS: an epilogue then never waits for a global load on its account (vector-memory
    // operations retire in order, so such a wait also drains the LDS-DMA and the stores issued around it)
    const bool has_bias = !(MASK & DACT) && g.bias != nullptr;     // the DACT variants (gradient GEMMs) carry none
    if (has_bias)
        for (int i = tid; i < g.N / 8; i += T)
            *reinterpret_cast<u32x4 *>(smem + LDS_BIAS + i * 16) = static_cast<const u32x4 *>(g.bias)[i];
    __syncthreads();

    int tile_cur = __builtin_amdgcn_readfirstlane(tq[0]);   // the tile being computed (wave-uniform: keep it scalar)
    if (tile_cur < 0) { finish(); return; }
    // thread 0's draw in flight: issued in one epilogue (here for the first), read in the next, a whole tile later.
    // (Read in the epilogue that issues it, the wait for the atomic's return -- which, vector-memory operations
    // retiring in order, is also a wait for every LDS-DMA issued before it -- stalled wave 0, and through the
    // barriers the workgroup, once per tile.)  The exit path waits for vmcnt(0) before finish().
    unsigned drawn = 0;
    if (tid == 0) drawn = atomicAdd(counters + xcd, 1u);
    const int tile_first = tile_cur;
    int t = 0;                                  // tiles finished by this workgroup

    // --- staging: this thread's two 16-byte pieces of a half-tile (LDS rows r1 and r1 + 64) ---
    //   A0: LDS row 64 grp + w  <-  tile row grp*GR + w            (w < 64: fragments 0-3 of wave group grp)
    //   A1: LDS row 64 grp + w  <-  tile row grp*GR + 64 + w       (w < GR - 64; the other rows of the slot copy earlier ones, never read)
    //   Bh: LDS row 32 wc + 16 jj + l  <-  column 64 wc + 16 (l >> 2) + 4 (2h + jj) + (l & 3): fragment order, so that a
    //       lane's accumulators are 16 consecutive columns
    const int r1 = 8 * wid + (lane >> 3);
    const int sc = (lane & 7) ^ ((r1 >> 1) & 7);                 // source chunk that lands in LDS chunk lane & 7
    const int w1 = (r1 < GR - 64) ? r1 : r1 - (NI == 7 ? 16 : 32);   // NI = 6: rows 32..63 copy 0..31
    const int b_row = (r1 >> 5) * 64 + ((r1 & 15) >> 2) * 16 + ((r1 >> 4) & 1) * 4 + (r1 & 3);
    // per-thread byte offsets (32 bits: the dispatcher bounds 256 rows x ld), added to wave-uniform bases so that the
    // LDS-DMA takes its scalar-base + 32-bit-offset form: three offset registers and no 64-bit vector adds in the loop
    const uint32_t voffA0 = (uint32_t(r1) * uint32_t(g.lda) + sc * 8) * 2, voffA1 = (uint32_t(64 + w1) * uint32_t(g.lda) + sc * 8) * 2;
    const uint32_t voffB = (uint32_t(b_row) * uint32_t(g.ldb) + sc * 8) * 2;
    const size_t a_second = size_t(GR) * g.lda, b_second = size_t(128) * g.ldb, b_h1 = size_t(8) * g.ldb;

    auto cursor_at = [&](int seq, int tile, bool is_a) __attribute__((always_inline)) {
        Cursor c;
        c.tile = seq;                                          // position in this workgroup's sequence of tiles
        c.k0 = 0;
        if (tile < 0) tile = tile_first;                       // past the end: re-stage the first tile (never consumed)
        c.p = is_a ? A + size_t(min((tile >> 16) * BM, m_last)) * g.lda : B + size_t(tile & 0xFFFF) * 256 * g.ldb;
        return c;
    };
    // Where the cursors go when they leave the current tile: the operand origins of the workgroup's next tile, worked
    // out (queue entry from LDS, two 64-bit multiplies) during the first k-tile of every tile, behind that phase's
    // MFMAs -- done at the crossing itself it cost every wave ~800 clocks per tile (measured with the stamped lab build, tools/gemm_lab/gemm8p_lab.hip).
    const uint16_t *next_a = nullptr, *next_b = nullptr;
    auto set_next = [&](int seq) __attribute__((always_inline)) {
        const Cursor na = cursor_at(seq, __builtin_amdgcn_readfirstlane(tq[seq & 3]), true);
        const Cursor nb = cursor_at(seq, __builtin_amdgcn_readfirstlane(tq[seq & 3]), false);
        next_a = na.p;
        next_b = nb.p;
    };
    auto advance = [&](Cursor &c, bool is_a) __attribute__((always_inline)) {
        if (c.k0 + 64 == K) { c.tile++; c.k0 = 0; c.p = is_a ? next_a : next_b; }
        else { c.k0 += 64; c.p += 64; }
    };
    // The LDS-DMA is issued from inline asm (scalar base + 32-bit lane offset, LDS base of the wave in M0), not through
    // __builtin_amdgcn_global_load_lds: while hipcc (ROCm 7.2) knows of an LDS-DMA in flight, every s_waitcnt it
    // inserts for an ordinary load is vmcnt(0) -- in the epilogue that meant waiting for the stores just issued to be
    // acknowledged before the next batch of residual rows could be used.  Unaware of the DMA it counts (vmcnt(n) with
    // n = the younger operations it knows of), which is never too weak: vector-memory operations retire in order and
    // the DMA it does not count only add younger entries.  All waits that concern the DMA itself are explicit
    // (wait_vm<>), as before.
    const int m0_wave = wid * 1024;
    auto dma16 = [&](const uint16_t *base, uint32_t voff, int lds) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :
                     : "v"(voff), "s"(base), "s"(lds)
                     : "memory");
    };
    auto stage_a = [&](const Cursor &c, int buf, int h) __attribute__((always_inline)) {
        const int d = m0_wave + buf * KTB + h * HALF;
        const uint32_t vo = h ? voffA1 : voffA0;
        dma16(c.p, vo, d);
        dma16(c.p + a_second, vo, d + 8192);
    };
    auto stage_b = [&](const Cursor &c, int buf, int h) __attribute__((always_inline)) {
        const uint16_t *p = c.p + (h ? b_h1 : 0);
        const int d = m0_wave + buf * KTB + 2 * HALF + h * HALF;
        dma16(p, voffB, d);
        dma16(p + b_second, voffB, d + 8192);
    };

    // --- fragment reads: lane = (row nl, k chunk q) of a 16 x 32 fragment; the second k-step is the chunk ^ 4 ---
    const int o0 = nl * 128 + ((q ^ ((nl >> 1) & 7)) << 4);
    const int a_off0 = wr * 8192 + o0, a_off1 = wr * 8192 + (o0 ^ 64);                    // + h * HALF + 2048 i
    const int b_off0 = 2 * HALF + wc * 4096 + o0, b_off1 = 2 * HALF + wc * 4096 + (o0 ^ 64);   // + h * HALF + 2048 jj
    bf16x8 fa0[4][2], fa1[NI - 4][2], fb0[2][2], fb1[2][2];
    auto read_a0 = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            fa0[i][0] = *reinterpret_cast<const bf16x8 *>(smem + a_off0 + buf * KTB + i * 2048);
            fa0[i][1] = *reinterpret_cast<const bf16x8 *>(smem + a_off1 + buf * KTB + i * 2048);
        }
    };
    auto read_a1 = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NI - 4; i++) {
            fa1[i][0] = *reinterpret_cast<const bf16x8 *>(smem + a_off0 + buf * KTB + HALF + i * 2048);
            fa1[i][1] = *reinterpret_cast<const bf16x8 *>(smem + a_off1 + buf * KTB + HALF + i * 2048);
        }
    };
    auto read_b = [&](bf16x8 (&fb)[2][2], int buf, int sub) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            fb[j][0] = *reinterpret_cast<const bf16x8 *>(smem + b_off0 + buf * KTB + sub * HALF + j * 2048);
            fb[j][1] = *reinterpret_cast<const bf16x8 *>(smem + b_off1 + buf * KTB + sub * HALF + j * 2048);
        }
    };

    f32x4 acc[NI][4];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // acc[i][j][r] = C[m][n]: m = row0 + 16 i + nl, n = col0 + 16 q + 4 j + r  (B rows are in fragment order)
    auto mma0 = [&](const bf16x8 (&fb)[2][2], int bsub) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][bsub * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa0[i][kk], acc[i][bsub * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto mma1 = [&](const bf16x8 (&fb)[2][2], int bsub) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int i = 0; i < NI - 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[4 + i][bsub * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa1[i][kk], acc[4 + i][bsub * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    const uint32_t thresh = (MASK & DROP) ? drop_thresh(g.dropout_p) : 0u;
    const float keep_scale = (MASK & DROP) ? 1.f / (1.f - g.dropout_p) : 1.f;
    const uint32_t seed_e = (MASK & DROP) ? eff_seed(g.dropout_seed, g.seed_off) : 0u;
    const uint32_t rk_in0 = (MASK & DROP) ? mix32(seed_e) : 0u, rk_in1 = (MASK & DROP) ? mix32(seed_e ^ 0x7FEB352Du) : 0u;
    const float dact_scale = g.dact_scale != 0.f ? g.dact_scale : 1.f;
    auto draw = [&]() __attribute__((always_inline)) {        // wave group 0, before its epilogue (tid 0 is in it)
        if (tid == 0) {
            tq[(t + 2) & 3] = tile_of(drawn);
            drawn = atomicAdd(counters + xcd, 1u);
        }
    };
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int tile = tile_cur;
        // Everything lane-dependent is rebuilt here from the thread index behind an empty asm: left to itself the
        // compiler hoists it out of the k-loop, where there is no register to keep it in.
        int te = tid;
        asm volatile("" : "+v"(te));
        const int le = te & 63, qe = le >> 4, ne = le & 15;
        StoreMap sm;
        {
            const int patch = LDS_PATCH + wid * 2048, r0 = le >> 3, ch = le & 7;
            sm.wa = patch + ne * 128 + (((2 * qe) ^ (ne & 7)) << 4);
            sm.ra = patch + r0 * 128 + ((ch ^ r0) << 4);
            sm.coff = long(r0 - ne) * g.ldc + (8 * ch - 16 * qe);
        }
        const int m_new = (tile >> 16) * BM;                   // rows below it belong to the previous tile too (ragged M)
        const int m0 = min(m_new, m_last) + wr * GR + ne, n0 = (tile & 0xFFFF) * 256 + wc * 64 + qe * 16;
        mfma_fence();
        float bv[16];                                           // this lane's 16 bias values, from the LDS copy
        if (has_bias) {
            const char *bp = smem + LDS_BIAS + n0 * 2;
            unpack8f(*reinterpret_cast<const u32x4 *>(bp), bv);
            unpack8f(*reinterpret_cast<const u32x4 *>(bp + 16), bv + 8);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) bv[r] = 0.f;
        }
        // All side-operand loads (residual / aux_in rows) are issued before the first use and before any store: one
        // memory latency per tile, and no wait that has an older store in front of it.
        const uint16_t *sp = (MASK & RES) ? static_cast<const uint16_t *>(g.residual) : static_cast<const uint16_t *>(g.aux_in);
        const size_t ld_side = (MASK & RES) ? size_t(g.ldr) : size_t(g.ldaux);
        float cs[16];                                           // CSUM: this lane's 16 columns summed over its NI rows
#pragma unroll
        for (int r = 0; r < 16; r++) cs[r] = 0.f;
        auto batch = [&](auto i0c, auto i1c) __attribute__((always_inline)) {
            constexpr int i0 = decltype(i0c)::value, i1 = decltype(i1c)::value;
            u32x4 side[i1 - i0][2];
            if ((MASK & DACT) && (MASK & BITS)) {
#pragma unroll
                for (int i = i0; i < i1; i++)
                    side[i - i0][0][0] = *reinterpret_cast<const uint16_t *>(static_cast<const uint8_t *>(g.actmask) +
                                                                             size_t(m0 + 16 * i) * g.ld_actmask + (n0 >> 3));
            } else if (MASK & (RES | DACT)) {
#pragma unroll
                for (int i = i0; i < i1; i++) {
                    const uint16_t *p = sp + size_t(m0 + 16 * i) * ld_side + n0;
                    side[i - i0][0] = *reinterpret_cast<const u32x4 *>(p);
                    side[i - i0][1] = *reinterpret_cast<const u32x4 *>(p + 8);
                }
            }
            u32x4 pk[i1 - i0][2];                                   // the packed rows of this batch (half the accumulators' registers)
            static_for<i0, i1>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                float v[16] = {acc[i][0][0], acc[i][0][1], acc[i][0][2], acc[i][0][3], acc[i][1][0], acc[i][1][1],
                               acc[i][1][2], acc[i][1][3], acc[i][2][0], acc[i][2][1], acc[i][2][2], acc[i][2][3],
                               acc[i][3][0], acc[i][3][1], acc[i][3][2], acc[i][3][3]};
                epilogue_math<MASK>(g, m0 + 16 * i, n0, v, bv, side[i - i0], thresh, keep_scale, dact_scale, rk_in0, rk_in1,
                                    pk[i - i0][0], pk[i - i0][1]);
                if (MASK & CSUM) {
                    const bool own = m0 + 16 * i >= m_new;           // a row the overlapping last tile shares is summed once
#pragma unroll
                    for (int r = 0; r < 16; r++) cs[r] += own ? v[r] : 0.f;      // the fp32 values that are stored as bf16
                }
            });
            patch_exchange<i1 - i0>(pk, sm);
            static_for<i0, i1>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                store_row_pair(g, m0 + 16 * i, n0, sm, pk[i - i0][0], pk[i - i0][1]);
            });
        };
        if constexpr (NI == 8 && ((MASK & (DROP | RES)) == (DROP | RES) || (MASK & (DACT | CSUM | BITS)) == (DACT | CSUM))) {
            batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
            batch(std::integral_constant<int, 4>{}, std::integral_constant<int, NI>{});
        } else {
            batch(std::integral_constant<int, 0>{}, std::integral_constant<int, NI>{});
        }
        if (MASK & CSUM) {
            // column sums of this wave's GR x 64 block: across the 16 lanes that hold the same columns (one DPP row),
            // then lane nl == 0 of every row writes 16 floats of partial row (2 * tile row + wave group); a fixed-order
            // pass over the (M / GR) partial rows follows the kernel (no atomics)
#pragma unroll
            for (int r = 0; r < 16; r++) cs[r] = row16_sum(cs[r]);
            if (ne == 0) {
                float *pp = static_cast<float *>(g.workspace) + size_t(2 * (tile >> 16) + wr) * g.N + n0;
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
                    *reinterpret_cast<f32x4 *>(pp + 4 * r4) = f32x4{cs[4 * r4], cs[4 * r4 + 1], cs[4 * r4 + 2], cs[4 * r4 + 3]};
            }
        }
        zero_acc();
    };

    // --- prologue: the state the loop expects at phase 0 of k-tile 0 ---
    Cursor ca = cursor_at(0, tile_cur, true), cb = cursor_at(0, tile_cur, false);
    stage_b(cb, 0, 0); stage_a(ca, 0, 0); stage_a(ca, 0, 1); stage_b(cb, 0, 1);
    advance(ca, true);
    advance(cb, false);
    if (P2) {                                 // two-phase schedule: A1 of k-tile 1 is staged by the loop's first load section
        stage_b(cb, 1, 0); stage_a(ca, 1, 0); stage_b(cb, 1, 1);
        advance(cb, false);
        wait_vm<6>();                         // k-tile 0
    } else {
        stage_b(cb, 1, 0); stage_a(ca, 1, 0); stage_a(ca, 1, 1); stage_b(cb, 1, 1);
        advance(ca, true);
        advance(cb, false);
        wait_vm<8>();                         // k-tile 0
    }
    bar();
    if (wr == 1) bar();                       // wave group 1 runs one barrier behind group 0
    zero_acc();

    // One k-tile g in buffer `buf`.  On entry k-tile g+1 is issued in full and `ca` / `cb` are k-tile g+2.
    // One counted wait per k-tile (phase 3, vmcnt(4): everything but the two half-tiles issued last has landed, i.e.
    // the whole next k-tile).  Waiting per half-tile just before its first use (vmcnt(8) in phases 0, 1 and 3) measured
    // 2-5 % slower in the same process.  Also measured and not kept: issuing each
    // phase's fragment reads one phase early, under the previous phase's MFMAs (role-swapping register sets): no
    // gain (1051 / 813 / 977 / 1073 vs 1043 / 819 / 985 / 1116 TFLOP/s on the four forward shapes) and spills at
    // 256 registers -- LDS read latency is not what the load section of a phase waits for.
    // Tile boundary.  Vector-memory operations retire in order, so a counted wait right after an epilogue also waits
    // for the epilogue's stores to be acknowledged.  Therefore A1 / B1 of k-tile g+2 are issued at the END of k-tile g (both groups are
    // through with these halves by then), which at a tile boundary is before the epilogue: everything the first
    // k-tile of the next tile waits for is older than the stores, and its wait can leave them outstanding --
    // vmcnt(4 + NSTORE).  The next wait is a whole k-tile later.
    constexpr int NSTORE = 2 * NI + ((MASK & CSUM) ? 4 : 0) + (((MASK & BITS) && (MASK & RELU)) ? NI : 0);  // store instructions per wave and epilogue
    auto stage_next = [&](int buf) __attribute__((always_inline)) {
        stage_a(ca, buf, 1);
        advance(ca, true);
        stage_b(cb, buf, 1);
        advance(cb, false);
        __builtin_amdgcn_sched_barrier(0);    // the address temporaries die here, not under the next phase's fragment reads
    };
    auto ktile = [&](int buf, bool first, bool last) __attribute__((always_inline)) {
        // phase 0: a0 x b0
        read_b(fb0, buf, 0);
        read_a0(buf);
        bar();
        wait_lgkm<0>();
        mma0(fb0, 0);
        bar();
        // phase 1: a1 x b0
        read_a1(buf);
        bar();
        wait_lgkm<0>();
        mma1(fb0, 0);
        if (first) set_next(t + 1);           // entry t + 1 was published an epilogue and several barriers ago
        bar();
        // phase 2: a1 x b1
        read_b(fb1, buf, 1);
        stage_b(cb, buf, 0);
        bar();
        wait_lgkm<0>();
        mma1(fb1, 1);
        bar();
        // phase 3: a0 x b1
        stage_a(ca, buf, 0);
        if (first && t > 0) wait_vm<4 + NSTORE>();   // the whole next k-tile has landed; the stores may still be out
        else wait_vm<4>();
        bar();
        mma0(fb1, 1);
        // ONE copy of the epilogue code (group 1 runs it before, group 0 after the tile's last barrier), so that the
        // other k-tiles branch over it once, not twice: a taken branch across ~5 KB of code costs ~100 clocks
        // (measured with the stamped lab build: the k-tile pairs that contained the two jumps were 300 clocks longer)
        if (last) {
            if (wr == 0) {
                bar();
                draw();
            }
            stage_next(buf);
            epilogue();
            if (wr == 1) bar();
        } else {
            bar();
            stage_next(buf);
        }
    };
    // Two-phase schedule of a k-tile (P2): the same ring, the same half-tiles, HALF the barriers.
    //     section   reads (LDS)     LDS-DMA issued                         MFMA (rows x cols)    counted wait before the barrier
    //     X load    A0, B0, B1      A1 of k-tile g+1                       -                     vmcnt(8): A1 of k-tile g
    //     X mma     -               -                                      a0 x b0, a0 x b1      -
    //     Y load    A1              A0, B0, B1 of k-tile g+2 (this buffer)  -                     vmcnt(8): A0, B0, B1 of k-tile g+1
    //     Y mma     -               -                                      a1 x b0, a1 x b1      -
    // A load section ends with lgkmcnt(0) BEFORE its barrier (the other group's 24-32 MFMAs cover the LDS latency), so a
    // half-tile may be overwritten from the section after its (second) read on: A0 / B0 / B1 of k-tile g are read in the
    // X sections (barriers 0 and 1 of the k-tile, the two groups one barrier apart) and restaged in the Y sections
    // (barriers 2 and 3); A1 is read in the Y sections and restaged in the next k-tile's X sections.  Every DMA has one
    // whole k-tile between issue and the wait that retires it, as in the four-phase schedule.  At a tile boundary
    // everything the next k-tile's two waits retire was issued BEFORE the epilogue's stores: vmcnt(8 + NSTORE).
    auto ktile2 = [&](int buf, bool first, bool last) __attribute__((always_inline)) {
        // X: a0 x (b0, b1)
        read_b(fb0, buf, 0);
        read_b(fb1, buf, 1);
        read_a0(buf);
        stage_a(ca, buf ^ 1, 1);
        advance(ca, true);
        __builtin_amdgcn_sched_barrier(0);
        if (first && t > 0) wait_vm<8 + NSTORE>();
        else wait_vm<8>();
        wait_lgkm<0>();
        bar();
        mma0(fb0, 0);
        mma0(fb1, 1);
        if (first) set_next(t + 1);           // entry t + 1 was published an epilogue and several barriers ago
        bar();
        // Y: a1 x (b0, b1)
        read_a1(buf);
        stage_b(cb, buf, 0);
        stage_a(ca, buf, 0);
        stage_b(cb, buf, 1);
        advance(cb, false);
        __builtin_amdgcn_sched_barrier(0);
        if (first && t > 0) wait_vm<8 + NSTORE>();
        else wait_vm<8>();
        wait_lgkm<0>();
        bar();
        mma1(fb0, 0);
        mma1(fb1, 1);
        if (last) {
            if (wr == 0) {
                bar();
                draw();
            }
            epilogue();
            if (wr == 1) bar();
        } else {
            bar();
        }
    };
    int kt = 0;
    for (;;) {
        if (P2) ktile2(0, kt == 0, false);
        else ktile(0, kt == 0, false);
        kt += 2;
        const bool last = kt == KT;
        if (P2) ktile2(1, false, last);
        else ktile(1, false, last);
        if (last) {
            kt = 0;
            t++;
            tile_cur = __builtin_amdgcn_readfirstlane(tq[t & 3]);
            if (tile_cur < 0) break;
        }
    }
    if (wr == 0) bar();
    wait_vm<0>();
    finish();
}


// ----------------------------------------------------------------------------------------------------------------
// Weight-gradient form: C[M,N] = sum_k A[k][m] B[k][n] (both operands k-major: dW = dY^T X with k = token rows).
// Same 8 phases and ring; what changes is the LDS image and the fragment reads:
//   * half-tile = [64 k][128 cols] bf16 (256-B rows, the "st" image of device_common.h: 32-byte chunk index XOR-ed
//     with (k & 3) | ((k >> 3) & 1) << 2), filled by LDS-DMA in 4-row x 256-B pieces (two full cache lines per row)
//     with the swizzle on the source address;
//   * a 16 x 32 fragment is two ds_read_b64_tr_b16 (the hardware transposes 4 k-rows x 16 columns per 16-lane
//     group), conflict-free on that image.
// The output has few tiles (9-36 for ViT-B) and a very long k (batch x tokens), so the launch is tiles x splits
// workgroups, each with one k-range, writing its fp32 partial tile to slab z of the workspace; sfcvit_gemm's ordered
// split-K reduction sums the slabs (deterministic, no atomics).
// ----------------------------------------------------------------------------------------------------------------
template <bool P2>
__global__ __launch_bounds__(T) void gemm8p_km_kernel(const sfcvit_gemm_args g, int kt_per_split, int nsplits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 2, wc = wid & 3;
    const int q4 = lane >> 4, nl = lane & 15;
    const int NT = g.N / 256, KT = g.K / 64;
    const int tiles = (g.M / 256) * NT;
    // unit = (k-range z, tile), z major.  The workgroups of an XCD (blockIdx % 8) take a contiguous chunk of the unit
    // list: they run the same k-range at the same time, so every A / B panel is fetched from HBM once per XCD
    // (tile-minor round-robin instead re-read dY 3x and X 12x: 1.85 GB for 385 MB of operands).
    const int unit = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (unit >= tiles * nsplits) return;
    const int tile = unit % tiles, z = unit / tiles;
    const int m0 = (tile / NT) * 256, n0 = (tile % NT) * 256;
    const int kt_beg = z * kt_per_split;
    const int nkt = min(kt_per_split, KT - kt_beg);            // even, >= 2 (host)
    const uint16_t *A = static_cast<const uint16_t *>(g.a) + size_t(kt_beg) * 64 * g.lda + m0;
    const uint16_t *B = static_cast<const uint16_t *>(g.b) + size_t(kt_beg) * 64 * g.ldb + n0;

    // --- staging: this thread's two 16-byte pieces of a half-tile (k-rows kr and kr + 32, 16-byte chunk lane & 15) ---
    const int kr = 4 * wid + (lane >> 4);
    const int sw = (kr & 3) | (((kr >> 3) & 1) << 2);
    const int c16 = ((((lane & 15) >> 1) ^ sw) << 1) | (lane & 1);       // source chunk that lands in LDS chunk lane & 15
    const size_t offA = size_t(kr) * g.lda + c16 * 8, offB = size_t(kr) * g.ldb + c16 * 8;
    const size_t a32 = size_t(32) * g.lda, b32 = size_t(32) * g.ldb, a64 = size_t(64) * g.lda, b64 = size_t(64) * g.ldb;
    char *const lds_piece = smem + tid * 16;
    const uint16_t *ca = A, *cb = B;                            // k-tiles being staged
    int akt = 0, bkt = 0;
    auto advance_a = [&]() __attribute__((always_inline)) {
        if (++akt < nkt) ca += a64;                             // past the end: keep re-staging the last k-tile (never consumed)
    };
    auto advance_b = [&]() __attribute__((always_inline)) {
        if (++bkt < nkt) cb += b64;
    };
    auto stage_a = [&](int buf, int h) __attribute__((always_inline)) {
        const uint16_t *p = ca + offA + h * 128;
        char *d = lds_piece + buf * KTB + h * HALF;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(p + a32), (lptr_t)(d + 8192), 16, 0, 0);
    };
    auto stage_b = [&](int buf, int h) __attribute__((always_inline)) {
        const uint16_t *p = cb + offB + h * 128;
        char *d = lds_piece + buf * KTB + 2 * HALF + h * HALF;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(p + b32), (lptr_t)(d + 8192), 16, 0, 0);
    };

    // --- transposed fragment reads: lane group (q4, nl>>2) reads 8 bytes of k-row 8 q4 + (nl>>2) (and + 4) at
    //     columns 16 c + 4 (nl & 3); the chunk swizzle of that row is q | (q4 & 1) << 2 ---
    const int qq = nl >> 2, pp = nl & 3;
    const int lane_sw = qq | ((q4 & 1) << 2);
    const int row_off = (8 * q4 + qq) * 256 + 8 * pp;
    int a_addr[8], b_addr[4];
#pragma unroll
    for (int i = 0; i < 8; i++)      // fragment i = 4 asub + ii: columns 64 wr + 16 ii of A half asub
        a_addr[i] = (i >> 2) * HALF + row_off + (((wr * 4 + (i & 3)) ^ lane_sw) << 5);
#pragma unroll
    for (int j = 0; j < 4; j++)      // fragment j = 2 bsub + jj: columns 32 wc + 16 jj of B half bsub
        b_addr[j] = 2 * HALF + (j >> 1) * HALF + row_off + (((wc * 2 + (j & 1)) ^ lane_sw) << 5);
    // The reads are inline asm on purpose: for the ds_read_tr builtin hipcc (ROCm 7.2) cannot tell which LDS bytes
    // are read and puts `s_waitcnt vmcnt(0)` in front of every group of them while an LDS-DMA is in flight, which
    // drains the whole prefetch ring four times per k-tile (measured: 550 instead of 730 TFLOP/s).  The compiler
    // therefore does not know when the data arrives: every use sits behind wait_lgkm<0>() + sched_barrier.
    // LDS addresses carry the ring buffer in bit 16 and are flipped once per k-tile (DS offsets are 16 bits).
#define SFCVIT_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
    auto tr_frag2 = [&](bf16x8 (&f)[2], int addr) __attribute__((always_inline)) {      // both k-steps of one fragment
        bf16x4 l0, h0, l1, h1;
        SFCVIT_TR(l0, addr, 0);
        SFCVIT_TR(h0, addr, 1024);
        SFCVIT_TR(l1, addr, 8192);
        SFCVIT_TR(h1, addr, 9216);
        f[0] = __builtin_shufflevector(l0, h0, 0, 1, 2, 3, 4, 5, 6, 7);
        f[1] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 fa0[4][2], fa1[4][2], fb0[2][2], fb1[2][2];
    auto read_a2 = [&](bf16x8 (&fa)[4][2], int sub, int i0) __attribute__((always_inline)) {   // fragments i0, i0+1
#pragma unroll
        for (int i = i0; i < i0 + 2; i++) tr_frag2(fa[i], a_addr[sub * 4 + i]);
    };
    auto read_b = [&](bf16x8 (&fb)[2][2], int sub) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; j++) tr_frag2(fb[j], b_addr[sub * 2 + j]);
    };
    auto flip = [&]() __attribute__((always_inline)) {           // the next k-tile lives in the other ring buffer
#pragma unroll
        for (int i = 0; i < 8; i++) a_addr[i] ^= KTB;
#pragma unroll
        for (int j = 0; j < 4; j++) b_addr[j] ^= KTB;
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[2][2], int asub, int bsub) __attribute__((always_inline)) {
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

    if (P2) {
        // two-phase schedule (see gemm8p_kernel's ktile2): X = {reads A0, B0, B1; stages A1 of k-tile g+1}, a0 x (b0, b1);
        // Y = {reads A1; stages A0, B0, B1 of k-tile g+2 into this buffer}, a1 x (b0, b1); every load section ends with
        // lgkmcnt(0) and vmcnt(8) before its barrier.  One A cursor serves both halves: A0 of k-tile g+2 (Y of k-tile g) is
        // staged before A1 of k-tile g+2 (X of k-tile g+1), then the cursor advances.
        stage_b(0, 0); stage_a(0, 0); stage_a(0, 1); stage_b(0, 1);        // k-tile 0 in full
        advance_a();
        advance_b();
        stage_b(1, 0); stage_a(1, 0); stage_b(1, 1);                       // A0, B0, B1 of k-tile 1
        advance_b();
        wait_vm<6>();
        bar();
        if (wr == 1) bar();
        auto ktile2 = [&](int buf) __attribute__((always_inline)) {
            read_b(fb0, 0);
            read_b(fb1, 1);
            read_a2(fa0, 0, 0);
            read_a2(fa0, 0, 2);
            stage_a(buf ^ 1, 1);
            advance_a();
            wait_vm<8>();
            wait_lgkm<0>();
            bar();
            mma(fa0, fb0, 0, 0);
            mma(fa0, fb1, 0, 1);
            bar();
            read_a2(fa1, 1, 0);
            read_a2(fa1, 1, 2);
            stage_b(buf, 0);
            stage_a(buf, 0);
            stage_b(buf, 1);
            advance_b();
            wait_vm<8>();
            wait_lgkm<0>();
            bar();
            mma(fa1, fb0, 1, 0);
            mma(fa1, fb1, 1, 1);
            bar();
            flip();
        };
        for (int k = 0; k < nkt; k += 2) {
            ktile2(0);
            ktile2(1);
        }
    } else {
        stage_b(0, 0); stage_a(0, 0); stage_a(0, 1); stage_b(0, 1);
        advance_a();
        advance_b();
        stage_b(1, 0); stage_a(1, 0);
        wait_vm<4>();
        bar();
        if (wr == 1) bar();
        // same schedule as gemm8p_kernel (table at the top of the file); the half-tiles are contiguous 128-column blocks
        // here, and a wave takes its a0 / a1 (b0 / b1) fragments from the first / second of them
        auto ktile = [&](int buf) __attribute__((always_inline)) {
            read_b(fb0, 0);
            read_a2(fa0, 0, 0);
            read_a2(fa0, 0, 2);
            stage_a(buf ^ 1, 1);
            advance_a();
            bar();
            wait_lgkm<0>();
            mma(fa0, fb0, 0, 0);
            bar();
            read_a2(fa1, 1, 0);
            read_a2(fa1, 1, 2);
            stage_b(buf ^ 1, 1);
            advance_b();
            bar();
            wait_lgkm<0>();
            mma(fa1, fb0, 1, 0);
            bar();
            read_b(fb1, 1);
            stage_b(buf, 0);
            bar();
            wait_lgkm<0>();
            mma(fa1, fb1, 1, 1);
            bar();
            stage_a(buf, 0);
            wait_vm<4>();
            bar();
            mma(fa0, fb1, 0, 1);
            bar();
            flip();
        };
        for (int k = 0; k < nkt; k += 2) {
            ktile(0);
            ktile(1);
        }
}
    if (wr == 0) bar();
    wait_vm<0>();
    mfma_fence();
    // acc[4 as + ii][2 bs + jj][r] = C[m0 + 128 as + 64 wr + 16 ii + nl][n0 + 128 bs + 32 wc + 16 jj + 4 q4 + r]
    float *slab = static_cast<float *>(g.workspace) + size_t(z) * g.M * g.N + size_t(m0 + wr * 64 + nl) * g.N + n0 + wc * 32 + 4 * q4;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<f32x4 *>(slab + size_t((i >> 2) * 128 + (i & 3) * 16) * g.N + (j >> 1) * 128 + (j & 1) * 16) = acc[i][j];
}

// Nine zero-initialised counters per (device, stream) (tile queues of the 8 XCDs + finished workgroups); every launch
// leaves them zero again, and launches on one stream are ordered, so a (device, stream) pair can keep its slot for
// ever.  Rule (include/sfcvit.h): one launch at a time per slot, i.e. a captured graph holding these kernels must not
// be replayed on two streams at once.  Each device has its own pool, allocated at the first call on that device
// (which must therefore not sit inside a hipGraph capture: warm up first); handing a slot to a new stream -- a
// capture stream, say -- allocates nothing.  A launch that fails re-zeroes its slot (requeue_reset).
struct DeviceState {
    unsigned *pool = nullptr;
    int next = 0;
    int cus = 0;
    std::unordered_map<hipStream_t, unsigned *> per_stream;
};
constexpr int MAX_DEVICES = 64, SLOTS = 1024, SLOT_UINTS = 16;
std::mutex g_mu;
DeviceState g_dev[MAX_DEVICES];

DeviceState *device_state() {              // the CURRENT device's state (g_mu held by the caller)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
    DeviceState *d = &g_dev[dev];
    if (!d->cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
        d->cus = prop.multiProcessorCount;
    }
    return d;
}

int device_cus() {
    std::lock_guard<std::mutex> lock(g_mu);
    DeviceState *d = device_state();
    return d ? d->cus : 0;
}

unsigned *queue_counters(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_mu);
    DeviceState *d = device_state();
    if (!d) return nullptr;
    auto it = d->per_stream.find(s);
    if (it != d->per_stream.end()) return it->second;
    if (!d->pool) {
        if (hipMalloc(reinterpret_cast<void **>(&d->pool), size_t(SLOTS) * SLOT_UINTS * sizeof(unsigned)) != hipSuccess ||
            hipMemset(d->pool, 0, size_t(SLOTS) * SLOT_UINTS * sizeof(unsigned)) != hipSuccess) {
            d->pool = nullptr;
            return nullptr;
        }
    }
    if (d->next >= SLOTS) return nullptr;        // 1024 distinct streams on one device: not a case worth more code
    unsigned *p = d->pool + size_t(d->next++) * SLOT_UINTS;
    d->per_stream[s] = p;
    return p;
}

template <int NI, int MASK>
int launch(const sfcvit_gemm_args &a, int grid, hipStream_t s) {
    const int LDS_TOTAL = LDS_BIAS + (a.bias ? a.N * 2 : 0);
    unsigned *counters = queue_counters(s);
    if (!counters) return fail(SFCVIT_ELAUNCH, "gemm8p: could not allocate the tile-queue counters");
    // the two-phase k-tile schedule is the default; SFCVIT_GEMM_2PHASE=0 selects the four-phase one (A/B in one process)
    const char *e = getenv("SFCVIT_GEMM_2PHASE");
    note_gemm_kernel(1, NI, MASK, !(e && e[0] == '0'));
    // Start-up stagger: the 32 workgroups of an XCD start in 4 groups 2 us apart.  Uniform tiles keep the 256 workgroups of a
    // launch in lockstep, so all of them reach their epilogue in the same microsecond and 33 MB of C hit the memory system at
    // once (the epilogue section of the first tiles of a launch takes 10 000 clocks, 5 600 once the workgroups have drifted
    // apart: profiles/r3/gemm8p_ktile_trace.txt); the tile queue absorbs the late starts (late workgroups draw fewer tiles).
    // Measured alone, M = 50 176 (tools/gemm_lab/ab_two_phase.py with AB_STAGGER): QKV 148.9 -> 141.7 us, out-proj 74.4 ->
    // 68.6, linear1 233.7 -> 230.9, linear2 208.0 -> 205.3, linear2 dX 225.8 -> 221.4, linear1 dX 206.6 -> 203.1, in_proj dX
    // 162.0 -> 161.6; training step 33.26-33.38 -> 32.86-32.89 ms (three alternating pairs of runs on one box).
    // SFCVIT_GEMM_STAGGER="slots,ticks" (10 ns) overrides; "1,0" = off.
    // Tile walk: column windows of 6 tiles for the wide GEMMs (N >= 1 792).  An XCD's 32 concurrent tiles are then ~5 row tiles x
    // 6 column tiles instead of ~3 x 9-12, i.e. 11 distinct operand panels instead of 12-15: memory-side fetch of the N = 3 072 /
    // 2 304 forward GEMMs 427 -> 263 MB per launch, L2 hit rate 0.56 -> 0.63, time unchanged (profiles/r4/gemm_tile_walk_ab.txt).
    // SFCVIT_GEMM_WALK = window width, 0 = row-major walk (round 3).
    static const int walk_env = [] { const char *w = getenv("SFCVIT_GEMM_WALK"); return w ? atoi(w) : -1; }();
    const int walk = walk_env >= 0 ? walk_env : (a.N / 256 > 6 ? 6 : 0);
    int stag_slots = 4, stag_ticks = 200;
    if (const char *st = getenv("SFCVIT_GEMM_STAGGER")) sscanf(st, "%d,%d", &stag_slots, &stag_ticks);
    if (stag_slots < 1) stag_slots = 1;
    {   // a launch whose workgroups draw one tile each has no lockstep to break: the delay would only lengthen it
        const long ntiles = long((a.M + 32 * NI - 1) / (32 * NI)) * (a.N / 256);
        if (ntiles < 2L * grid) stag_ticks = 0;
    }
    if (!(e && e[0] == '0')) {
        if (int rc = raise_lds_limit(reinterpret_cast<const void *>(&gemm8p_kernel<NI, MASK, true>), LDS_MAX, "gemm8p attribute")) return rc;
        hipLaunchKernelGGL((gemm8p_kernel<NI, MASK, true>), dim3(grid), dim3(T), LDS_TOTAL, s, a, counters, stag_slots, stag_ticks, walk);
        const int rc = check_launch("gemm8p");
        if (rc) (void)hipMemsetAsync(counters, 0, SLOT_UINTS * sizeof(unsigned), s);
        return rc;
    }
    if (int rc = raise_lds_limit(reinterpret_cast<const void *>(&gemm8p_kernel<NI, MASK>), LDS_MAX, "gemm8p attribute")) return rc;
    hipLaunchKernelGGL((gemm8p_kernel<NI, MASK>), dim3(grid), dim3(T), LDS_TOTAL, s, a, counters, stag_slots, stag_ticks, walk);
    const int rc = check_launch("gemm8p");
    if (rc) (void)hipMemsetAsync(counters, 0, SLOT_UINTS * sizeof(unsigned), s);   // a launch that did not run leaves no debt
    return rc;
}

template <int NI>
int launch_mask(const sfcvit_gemm_args &a, int mask, int grid, hipStream_t s) {
    switch (mask) {
    case 0: return launch<NI, 0>(a, grid, s);
    case RES: return launch<NI, RES>(a, grid, s);
    case DROP | RES: return launch<NI, DROP | RES>(a, grid, s);
    case RELU: return launch<NI, RELU>(a, grid, s);
    case RELU | DROP: return launch<NI, RELU | DROP>(a, grid, s);
    case DACT: return launch<NI, DACT>(a, grid, s);
    case DACT | CSUM: return launch<NI, DACT | CSUM>(a, grid, s);
    case RELU | BITS: return launch<NI, RELU | BITS>(a, grid, s);
    case RELU | DROP | BITS: return launch<NI, RELU | DROP | BITS>(a, grid, s);
    case DACT | BITS: return launch<NI, DACT | BITS>(a, grid, s);
    case DACT | CSUM | BITS: return launch<NI, DACT | CSUM | BITS>(a, grid, s);
    default: return -1;
    }
}

}  // namespace p8
}  // namespace

// The (device, stream) slot of zero-initialised counters (16 words; this file's kernels use words 0-8 and leave them zero) for
// other persistent kernels of the library: attention_bwd_fused.hip deals its items from words 12-13.  nullptr: none to be had.
unsigned *stream_counters(void *stream) { return p8::queue_counters(static_cast<hipStream_t>(stream)); }

// Weight-gradient form (both operands k-major, split-K into the workspace slabs).  Returns -1 when not eligible,
// else a status; *splits_used = number of slabs written (the caller runs the ordered reduction over them).
// k need not be a multiple of 128 (k = batch x tokens: 19 600 rows at batch 100): the kernel takes the largest multiple,
// *k_done says how far it got, and the caller adds the remaining < 128 rows as one more slab (a slab is kept free for it).
int gemm8p_km_dispatch(const sfcvit_gemm_args &a, int splits_req, int *splits_used, int *k_done, hipStream_t s) {
    using namespace p8;
    if (!a.a_kmajor || !a.b_kmajor || splits_req < 2) return -1;
    if (a.M % 256 || a.N % 256 || a.K < 256 || a.lda % 8 || a.ldb % 8) return -1;
    const int cus = device_cus();
    if (!cus) return -1;
    const int Kb = a.K / 128 * 128, tail = a.K - Kb;
    const int tiles = (a.M / 256) * (a.N / 256), KT = Kb / 64;
    // SFCVIT_RESERVE_CUS=n (read per call): leave n CUs out of the split, for nodes where collectives run beside backward -- a
    // launch of one workgroup per CU takes twice as long when it does not fit on the CUs that are free
    int reserve = 0;
    if (const char *rv = getenv("SFCVIT_RESERVE_CUS")) reserve = atoi(rv);
    if (reserve < 0 || reserve > cus / 2) reserve = 0;
    int splits = (cus - reserve) / tiles;                         // one workgroup per CU
    if (splits > splits_req) splits = splits_req;
    const int64_t slabs_avail = a.workspace_bytes / (int64_t(a.M) * a.N * int64_t(sizeof(float))) - (tail ? 1 : 0);
    if (splits > slabs_avail) splits = int(slabs_avail);
    if (splits < 2) return -1;
    int kps = ((KT + splits - 1) / splits + 1) / 2 * 2;           // k-tiles per split, even
    splits = (KT + kps - 1) / kps;
    if (splits < 2) return -1;
    sfcvit_gemm_args body = a;
    body.K = Kb;
    const char *e = getenv("SFCVIT_GEMM_2PHASE");              // "0": the four-phase k-tile schedule (A/B in one process)
    note_gemm_kernel(2, !(e && e[0] == '0'));
    if (e && e[0] == '0') {
        if (int rc = raise_lds_limit(reinterpret_cast<const void *>(&gemm8p_km_kernel<false>), LDS_BYTES, "gemm8p_km attribute")) return rc;
        hipLaunchKernelGGL(gemm8p_km_kernel<false>, dim3((tiles * splits + 7) / 8 * 8), dim3(T), LDS_BYTES, s, body, kps, splits);
    } else {
        if (int rc = raise_lds_limit(reinterpret_cast<const void *>(&gemm8p_km_kernel<true>), LDS_BYTES, "gemm8p_km attribute")) return rc;
        hipLaunchKernelGGL(gemm8p_km_kernel<true>, dim3((tiles * splits + 7) / 8 * 8), dim3(T), LDS_BYTES, s, body, kps, splits);
    }
    *splits_used = splits;
    *k_done = Kb;
    return check_launch("gemm8p_km");
}

// Called by sfcvit_gemm after argument validation.  -1 = not eligible (the caller tries the older kernels).
int gemm8p_dispatch(const sfcvit_gemm_args &a, int splits, hipStream_t s) {
    using namespace p8;
    if (a.a_kmajor || a.b_kmajor || splits != 1 || a.c_is_f32 || a.aux_out) return -1;
    if (a.act == SFCVIT_ACT_GELU || a.dact == SFCVIT_ACT_GELU) return -1;
    if (a.N % 256 || a.K % 128 || a.K < 256 || a.lda % 8 || a.ldb % 8 || a.ldc % 8) return -1;
    if (a.lda >= (1 << 21) || a.ldb >= (1 << 21)) return -1;       // 32-bit byte offsets within a tile
    if (a.M / 192 >= 32768 || a.N / 256 >= 65536) return -1;       // (row tile, column tile) packed into one int
    if (a.residual && (a.ldr % 8 || (reinterpret_cast<uintptr_t>(a.residual) & 15))) return -1;
    if (a.dact && (a.ldaux % 8 || (reinterpret_cast<uintptr_t>(a.aux_in) & 15))) return -1;
    if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return -1;
    int mask = 0;
    if (a.act == SFCVIT_ACT_RELU) mask |= RELU;
    if (a.dropout_p > 0.f) mask |= DROP;
    if (a.residual) mask |= RES;
    if (a.dact == SFCVIT_ACT_RELU) mask |= DACT;
    if (a.colsum_out) mask |= CSUM;                          // built with DACT only; other combinations fall back
    if (a.actmask && (mask & (RELU | DACT)) && !(mask & RES)) mask |= BITS;   // with RELU + RES the bits come from the pass over C
    if ((mask & DACT) && a.bias) return -1;                  // the DACT variants leave the bias out (register room)
    if (a.bias && a.N > BIAS_MAX_N) return -1;               // the bias vector lives in LDS
    const int cus = device_cus() / 8 * 8;
    if (cus < 8) return -1;
    // Tile height: the one whose rounds of tiles cost least.  A tile's time is not proportional to its rows: the stamped
    // k-tile (profiles/r3/gemm8p_ktile_trace.txt) takes 2 663 clocks at 256 rows and 2 487 at 224 (0.934, not 0.875) -- the
    // per-section hand-off does not shrink with the tile; 192 rows extrapolated.  With these weights N = 3 072 at M = 50 176
    // takes 256-row tiles (10 rounds x 1 000 < 11 x 934; measured 234 vs 238 us and 219 vs 226 us), N = 768 stays at 224.
    const int nt = a.N / 256;
    long best = -1;
    int ni = 0;
    // Any M >= one tile: a height that does not divide M makes the last row tile overlap its predecessor (kernel header),
    // which costs that tile's share of recomputed rows, i.e. it is priced as one more tile.  The overlapping tile reads
    // residual / aux_in rows another workgroup may be storing to if C aliases them: refused then.
    const bool aliased = a.c == a.residual || a.c == a.aux_in;
    for (int cand : {8, 7, 6}) {
        if (a.M < 32 * cand || (a.M % (32 * cand) && aliased)) continue;
        const long tiles = long((a.M + 32 * cand - 1) / (32 * cand)) * nt;
        const long cost = ((tiles + cus - 1) / cus) * (cand == 8 ? 1000 : cand == 7 ? 934 : 870);
        if (best < 0 || cost < best) { best = cost; ni = cand; }
    }
    if (!ni) return -1;
    if (a.force_generic == 8) { if (a.M < 256 || (a.M % 256 && aliased)) return -1; ni = 8; }    // tests: pin the 256-row tile
    if (a.force_generic == 9) { if (a.M < 224 || (a.M % 224 && aliased)) return -1; ni = 7; }    // tests: pin the 224-row tile
    if (a.force_generic == 10) { if (a.M < 192 || (a.M % 192 && aliased)) return -1; ni = 6; }   // tests: pin the 192-row tile
    const int nparts = 2 * ((a.M + 32 * ni - 1) / (32 * ni));            // CSUM: one partial row per (row tile, wave group)
    if (mask & CSUM) {
        const int64_t need = int64_t(nparts) * a.N * int64_t(sizeof(float));
        if (!a.workspace || a.workspace_bytes < need || (reinterpret_cast<uintptr_t>(a.workspace) & 15)) return -1;
    }
    const int rc = ni == 8 ? launch_mask<8>(a, mask, cus, s) : ni == 7 ? launch_mask<7>(a, mask, cus, s) : launch_mask<6>(a, mask, cus, s);
    if (rc == 0 && (mask & CSUM))
        return launch_colsum_reduce(static_cast<const float *>(a.workspace), nparts, a.N, a.colsum_out, a.colsum_bf16, s);
    return rc;
}

}  // namespace sfcvit

