// Persistent 8-phase bf16 MFMA GEMM for gfx950: C[M,N] = epilogue(A[M,K] B[N,K]^T), both operands
// k-contiguous -- every forward GEMM of the encoder and (on the per-step transposed weight) every dX GEMM.
//
//   * tile (32*NI) x 256 x 64 with NI = 8 or 7 (256 or 224 rows: the height that wastes least of the last round
//     of tiles on 256 CUs; M = 50176 = 196 * 256 = 224 * 224), 8 waves = 2 wave groups (row halves) x 4 column
//     blocks, wave tile (16*NI) x 64, 16x16x32 MFMA.
//   * one workgroup per CU walks a list of output tiles (round r -> tile r*G + u).  The k-tile stream does not stop
//     at a tile boundary: operands go HBM/L2 -> LDS by LDS-DMA into a ring of two 64 KiB k-tiles (four 16 KiB
//     half-tiles each: A rows of group 0 / group 1, B columns 0-127 / 128-255) and the ring keeps prefetching the
//     next tile's first k-tiles under the current tile's last phases and its epilogue -- no prologue per tile.
//   * a k-tile is 4 phases; a phase = {ds_read the fragments of one quadrant, issue the LDS-DMA of one half-tile,
//     barrier, 16 MFMAs, barrier}.  The two wave groups run one barrier apart, so one group's MFMAs cover the other
//     group's LDS reads and DMA issue (cdna_hip_programming.md §5, "The 256^2 8-phase template"; the staging order
//     B0, A0, B1 of k-tile g+2 in phases 1-3 and A1 of k-tile g+1 in phase 0, one counted vmcnt(6) per k-tile in
//     phase 3, buffers read one phase after the wait that retires them, follow its rules).
//   * LDS image of a half-tile: [128 rows][64 k] bf16, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7)
//     (ds_read_b128 conflict-free, tools/lds_conflicts.py); LDS-DMA writes are lane-linear, so the swizzle is
//     applied to the per-lane SOURCE address.  B rows are stored in fragment order: LDS row 16j + l holds column
//     16(l >> 2) + 4j + (l & 3) of the wave's 64, which makes the accumulators of a lane 16 CONSECUTIVE columns of
//     one row -- the epilogue reads bias / residual / aux and writes C as 16-byte vectors straight from registers,
//     no LDS transpose.
//   * epilogue variants are compile-time (MASK) so that the fully unrolled per-row code stays small (DESIGN.md §5:
//     the instruction cache punished a runtime option tree).  Wave group 1 runs its epilogue before, group 0 after
//     the tile's last barrier: both run concurrently instead of one after the other.
//
// Anything not eligible (k-major operands, split-K, GELU / aux_out / fp32 output, shapes off the tile grid) returns
// -1 from gemm8p_dispatch and takes the older kernels (gemm256.hip, gemm.hip).
#include "common_host.h"
#include "gemm_core.h"
#include <type_traits>

namespace sfcvit {
namespace {

using namespace gemm_core;

namespace p8 {

constexpr int T = 512, HALF = 16384, KTB = 65536, LDS_BYTES = 2 * KTB;
enum { RELU = 1, DROP = 2, RES = 4, DACT = 8 };

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

// Compile-time loop: the accumulator array must only ever be indexed by constants (a loop the unroller gives up on
// -- the epilogue body is large -- turns acc[i] into a runtime index and the whole array into scratch memory).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct Cursor {            // the k-tile being staged: index in this workgroup's tile list, k offset, operand origins
    int tile, k0;
    const uint16_t *a, *b;
};

// One output row segment of 16 consecutive columns, in the documented order: bias, act, dropout, residual, dact, store.
template <int MASK>
__device__ __forceinline__ void epilogue_row(const sfcvit_gemm_args &g, int m, int n, float (&v)[16], const float (&bv)[16],
                                             uint32_t thresh, float keep_scale, float dact_scale) {
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] += bv[r];
    if (MASK & RELU) {
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = fmaxf(v[r], 0.f);
    }
    if (MASK & DROP) {
        const uint32_t rk = drop_row_key(g.dropout_seed, uint64_t(m) + uint64_t(uint32_t(g.row_offset)));
#pragma unroll
        for (int p = 0; p < 8; p++) {
            bool k0, k1;
            drop_keep2(rk, uint32_t(n >> 1) + p, thresh, k0, k1);
            v[2 * p] = k0 ? v[2 * p] * keep_scale : 0.f;
            v[2 * p + 1] = k1 ? v[2 * p + 1] * keep_scale : 0.f;
        }
    }
    if (MASK & RES) {
        const uint16_t *p = static_cast<const uint16_t *>(g.residual) + size_t(m) * g.ldr + n;
        float rv[16];
        unpack8f(*reinterpret_cast<const u32x4 *>(p), rv);
        unpack8f(*reinterpret_cast<const u32x4 *>(p + 8), rv + 8);
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] += rv[r];
    }
    if (MASK & DACT) {
        const uint16_t *p = static_cast<const uint16_t *>(g.aux_in) + size_t(m) * g.ldaux + n;
        float a[16];
        unpack8f(*reinterpret_cast<const u32x4 *>(p), a);
        unpack8f(*reinterpret_cast<const u32x4 *>(p + 8), a + 8);
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = a[r] > 0.f ? v[r] * dact_scale : 0.f;
    }
    uint16_t *c = static_cast<uint16_t *>(g.c) + size_t(m) * g.ldc + n;
    *reinterpret_cast<u32x4 *>(c) = u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
    *reinterpret_cast<u32x4 *>(c + 8) = u32x4{pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15])};
}

template <int NI, int MASK>
__global__ __launch_bounds__(T) void gemm8p_kernel(const sfcvit_gemm_args g) {
    constexpr int BM = 32 * NI, GR = 16 * NI;               // tile rows, rows per wave group
    extern __shared__ __attribute__((aligned(16))) char smem[];   // ONE array: ring of 2 x [A0 | A1 | B0 | B1]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 2, wc = wid & 3;
    const int q = lane >> 4, nl = lane & 15;
    const int K = g.K, NT = g.N / 256, ntiles = (g.M / BM) * NT, KT = K / 64;
    const uint16_t *A = static_cast<const uint16_t *>(g.a);
    const uint16_t *B = static_cast<const uint16_t *>(g.b);
    // The 32 workgroups of an XCD (blockIdx % 8) take 32 consecutive tiles of a round: they share A row panels
    // in that XCD's L2.
    const int G = gridDim.x;
    const int u = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    if (u >= ntiles) return;
    const int my_tiles = (ntiles - u + G - 1) / G;

    // --- staging: this thread's two 16-byte pieces of a half-tile (LDS rows r1 and r1 + 64) ---
    const int r1 = 8 * wid + (lane >> 3);
    const int sc = (lane & 7) ^ ((r1 >> 1) & 7);                 // source chunk that lands in LDS chunk lane & 7
    const int b_row = ((r1 & 15) >> 2) * 16 + ((r1 >> 4) & 3) * 4 + (r1 & 3);
    // NI = 7: a group owns 112 rows; LDS rows 112..127 are filled with a copy of rows 96..111 (never read).
    const int a_row2 = (r1 + 64 < GR) ? r1 + 64 : r1 + 64 - 16;
    const size_t offA1 = size_t(r1) * g.lda + sc * 8, offA2 = size_t(a_row2) * g.lda + sc * 8;
    const size_t offB1 = size_t(b_row) * g.ldb + sc * 8, offB2 = offB1 + size_t(64) * g.ldb;
    const size_t a_half = size_t(GR) * g.lda, b_half = size_t(128) * g.ldb;
    char *const lds_piece = smem + tid * 16;

    auto cursor_at = [&](int t) __attribute__((always_inline)) {
        Cursor c;
        c.tile = t;
        c.k0 = 0;
        const int tile = (t < my_tiles ? t : 0) * G + u;       // past the end: re-stage the first tile (never consumed)
        c.a = A + size_t(tile / NT) * BM * g.lda;
        c.b = B + size_t(tile % NT) * 256 * g.ldb;
        return c;
    };
    auto advance = [&](Cursor &c) __attribute__((always_inline)) {
        if (c.k0 + 64 == K) c = cursor_at(c.tile + 1);
        else { c.k0 += 64; c.a += 64; c.b += 64; }
    };
    auto stage_a = [&](const Cursor &c, int buf, int h) __attribute__((always_inline)) {
        const uint16_t *p = c.a + (h ? a_half : 0);
        char *d = lds_piece + buf * KTB + h * HALF;
        __builtin_amdgcn_global_load_lds((gptr_t)(p + offA1), (lptr_t)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(p + offA2), (lptr_t)(d + 8192), 16, 0, 0);
    };
    auto stage_b = [&](const Cursor &c, int buf, int h) __attribute__((always_inline)) {
        const uint16_t *p = c.b + (h ? b_half : 0);
        char *d = lds_piece + buf * KTB + 2 * HALF + h * HALF;
        __builtin_amdgcn_global_load_lds((gptr_t)(p + offB1), (lptr_t)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(p + offB2), (lptr_t)(d + 8192), 16, 0, 0);
    };

    // --- fragment reads: lane = (row nl, k chunk q) of a 16 x 32 fragment; the second k-step is the chunk ^ 4 ---
    const int o0 = nl * 128 + ((q ^ ((nl >> 1) & 7)) << 4);
    const int a_off0 = wr * HALF + o0, a_off1 = wr * HALF + (o0 ^ 64);
    const int b_off0 = 2 * HALF + (wc >> 1) * HALF + (wc & 1) * 8192 + o0, b_off1 = b_off0 ^ 64;
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
            fa1[i][0] = *reinterpret_cast<const bf16x8 *>(smem + a_off0 + buf * KTB + (4 + i) * 2048);
            fa1[i][1] = *reinterpret_cast<const bf16x8 *>(smem + a_off1 + buf * KTB + (4 + i) * 2048);
        }
    };
    auto read_b = [&](bf16x8 (&fb)[2][2], int buf, int sub) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            fb[j][0] = *reinterpret_cast<const bf16x8 *>(smem + b_off0 + buf * KTB + (sub * 2 + j) * 2048);
            fb[j][1] = *reinterpret_cast<const bf16x8 *>(smem + b_off1 + buf * KTB + (sub * 2 + j) * 2048);
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
    auto mma0 = [&](const bf16x8 (&fb)[2][2], int bsub) {
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
    auto mma1 = [&](const bf16x8 (&fb)[2][2], int bsub) {
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
    const float dact_scale = g.dact_scale != 0.f ? g.dact_scale : 1.f;
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        const int tile = t * G + u;
        const int m0 = (tile / NT) * BM + wr * GR + nl, n0 = (tile % NT) * 256 + wc * 64 + q * 16;
        mfma_fence();
        float bv[16];
        if (g.bias) {
            const uint16_t *bp = static_cast<const uint16_t *>(g.bias) + n0;
            unpack8f(*reinterpret_cast<const u32x4 *>(bp), bv);
            unpack8f(*reinterpret_cast<const u32x4 *>(bp + 8), bv + 8);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) bv[r] = 0.f;
        }
        static_for<0, NI>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            float v[16] = {acc[i][0][0], acc[i][0][1], acc[i][0][2], acc[i][0][3], acc[i][1][0], acc[i][1][1],
                           acc[i][1][2], acc[i][1][3], acc[i][2][0], acc[i][2][1], acc[i][2][2], acc[i][2][3],
                           acc[i][3][0], acc[i][3][1], acc[i][3][2], acc[i][3][3]};
            epilogue_row<MASK>(g, m0 + 16 * i, n0, v, bv, thresh, keep_scale, dact_scale);
        });
        zero_acc();
    };

    // --- prologue: k-tile 0 landed, B0 / A0 / B1 of k-tile 1 in flight ---
    Cursor cs = cursor_at(0);
    stage_b(cs, 0, 0); stage_a(cs, 0, 0); stage_b(cs, 0, 1); stage_a(cs, 0, 1);
    advance(cs);
    stage_b(cs, 1, 0); stage_a(cs, 1, 0); stage_b(cs, 1, 1);
    wait_vm<6>();
    bar();
    if (wr == 1) bar();                       // wave group 1 runs one barrier behind group 0
    zero_acc();

    // One k-tile.  `cs` is k-tile g+1 in phase 0 (its A1 is the only half-tile not issued yet) and g+2 afterwards.
    auto ktile = [&](int buf, bool last, int t) __attribute__((always_inline)) {
        // phase 0: rows 0-63 x columns 0-31 of the wave tile
        read_b(fb0, buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a0(buf);
        stage_a(cs, buf ^ 1, 1);
        advance(cs);
        wait_lgkm<8>();                       // B0 of this buffer has been read: it is restaged in phase 1
        bar();
        wait_lgkm<0>();
        mma0(fb0, 0);
        bar();
        // phase 1: rows 0-63 x columns 32-63
        read_b(fb1, buf, 1);
        stage_b(cs, buf, 0);
        bar();
        wait_lgkm<0>();
        mma0(fb1, 1);
        bar();
        // phase 2: rows 64.. x columns 32-63
        read_a1(buf);
        stage_a(cs, buf, 0);
        bar();
        wait_lgkm<0>();
        mma1(fb1, 1);
        bar();
        // phase 3: rows 64.. x columns 0-31; the other buffer (k-tile g+1) lands before anyone reads it in the next phase
        stage_b(cs, buf, 1);
        wait_vm<6>();
        bar();
        mma1(fb0, 0);
        if (last && wr == 1) epilogue(t);
        bar();
        if (last && wr == 0) epilogue(t);
    };
    int kt = 0, t = 0;
    const int total = my_tiles * KT;
    for (int gk = 0; gk < total; gk += 2) {
        ktile(0, false, t);
        kt += 2;
        const bool last = kt == KT;
        ktile(1, last, t);
        if (last) { kt = 0; t++; }
    }
    if (wr == 0) bar();
    wait_vm<0>();
}

template <int NI, int MASK>
int launch(const sfcvit_gemm_args &a, int grid, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm8p_kernel<NI, MASK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return check_launch("gemm8p attribute");
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm8p_kernel<NI, MASK>), dim3(grid), dim3(T), LDS_BYTES, s, a);
    return check_launch("gemm8p");
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
    default: return -1;
    }
}

}  // namespace p8
}  // namespace

// Called by sfcvit_gemm after argument validation.  -1 = not eligible (the caller tries the older kernels).
int gemm8p_dispatch(const sfcvit_gemm_args &a, int splits, hipStream_t s) {
    using namespace p8;
    if (a.a_kmajor || a.b_kmajor || splits != 1 || a.c_is_f32 || a.aux_out) return -1;
    if (a.act == SFCVIT_ACT_GELU || a.dact == SFCVIT_ACT_GELU) return -1;
    if (a.N % 256 || a.K % 128 || a.lda % 8 || a.ldb % 8 || a.ldc % 8) return -1;
    if (a.residual && (a.ldr % 8 || (reinterpret_cast<uintptr_t>(a.residual) & 15))) return -1;
    if (a.dact && (a.ldaux % 8 || (reinterpret_cast<uintptr_t>(a.aux_in) & 15))) return -1;
    if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return -1;
    int mask = 0;
    if (a.act == SFCVIT_ACT_RELU) mask |= RELU;
    if (a.dropout_p > 0.f) mask |= DROP;
    if (a.residual) mask |= RES;
    if (a.dact == SFCVIT_ACT_RELU) mask |= DACT;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        cus = prop.multiProcessorCount / 8 * 8;
        if (cus < 8) return -1;
    }
    // Tile height: the one whose last round of tiles wastes least (cost = rounds x rows per tile).
    const int nt = a.N / 256;
    long best = -1;
    int ni = 0;
    for (int cand : {8, 7}) {
        if (a.M % (32 * cand)) continue;
        const long tiles = long(a.M / (32 * cand)) * nt;
        const long cost = ((tiles + cus - 1) / cus) * cand;
        if (best < 0 || cost < best) { best = cost; ni = cand; }
    }
    if (!ni) return -1;
    if (a.force_generic == 8) ni = (a.M % 256 == 0) ? 8 : ni;          // tests: pin the 256-row tile
    if (a.force_generic == 9) { if (a.M % 224) return -1; ni = 7; }    // tests: pin the 224-row tile
    return ni == 8 ? launch_mask<8>(a, mask, cus, s) : launch_mask<7>(a, mask, cus, s);
}

}  // namespace sfcvit
