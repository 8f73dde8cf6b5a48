// bf16 MFMA GEMM with fused epilogue for gfx950 (see sfcvit_gemm in include/sfcvit.h).
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ),  fp32 accumulation.
//
// Tile machinery: gemm_core.h.  Operands are staged global -> registers -> LDS (two
// LDS buffers, one barrier per k-tile; the global loads of tile t+1 are issued before
// the MFMAs of tile t and written to LDS after them).  Each operand may be stored
// k-contiguous ("kc" LDS image, ds_read_b128 fragments) or k-major ("st" image,
// ds_read_b64_tr_b16 fragments), which covers y = x W^T, dx = dy W and dW = dy^T x
// without any transposed copy in HBM.
//
// Weight-gradient shapes (M, N small; K = batch * tokens) have too few output tiles
// to fill 256 CUs, so K can be split over blockIdx.z: every split stores its fp32
// tile into its own slab of a caller-provided workspace with plain 16-byte stores
// and splitk_reduce sums the slabs in a fixed order (bitwise reproducible, no
// atomics; MI355X_MICROARCH.md "Global float atomics" prices the alternative).
#include "common_host.h"
#include "gemm_core.h"

namespace sfcvit {
namespace {

using namespace gemm_core;

template <bool A_KM, bool B_KM, bool HEAVY>
__global__ __launch_bounds__(THREADS, 2) void gemm_kernel(const sfcvit_gemm_args g, int k_per_split, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + BN - 1) / BN;
    int tile, split;
    if (nsplit == 1) {
        tile = xcd_remap(blockIdx.x, gridDim.x);
        split = 0;
    } else {
        const int tiles = gridDim.x / nsplit;
        if (tiles < 64) {
            // Few output tiles, many k-ranges (dW of a 768 x 768 weight: 36 tiles x 24 ranges).
            // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share an L2): give every
            // XCD its own k-ranges and ALL tiles of them, so that the A / B k-slabs of a range are
            // fetched into one L2 only and shared there by the tiles running side by side
            // (measured 140 -> 107 us on that shape; larger outputs measured slower this way).
            // nsplit % 8 == 0 (host); speed only, never correctness.
            const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
            tile = idx % tiles;
            split = (idx / tiles) * 8 + xcd;
        } else {
            split = blockIdx.x / tiles;
            tile = xcd_remap(blockIdx.x - split * tiles, tiles);
        }
    }
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const uint16_t *A = static_cast<const uint16_t *>(g.a);
    const uint16_t *B = static_cast<const uint16_t *>(g.b);
    const int kbeg = split * k_per_split;
    const int kend = min(g.K, kbeg + k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x4 acc[4][4];
    zero_acc(acc);

    Stage sa, sb;
    load_tile<A_KM>(sa, A, g.lda, m0, g.M, kbeg, kend, tid);
    load_tile<B_KM>(sb, B, g.ldb, n0, g.N, kbeg, kend, tid);
    store_tile<A_KM>(sa, smem, tid);
    store_tile<B_KM>(sb, smem + TILE_BYTES, tid);
    __syncthreads();

    for (int kt = 0; kt < nk; kt++) {
        const char *ia = smem + (kt & 1) * 2 * TILE_BYTES;
        const bool more = kt + 1 < nk;
        if (more) {
            load_tile<A_KM>(sa, A, g.lda, m0, g.M, kbeg + (kt + 1) * BK, kend, tid);
            load_tile<B_KM>(sb, B, g.ldb, n0, g.N, kbeg + (kt + 1) * BK, kend, tid);
        }
        mma_tile<A_KM, B_KM>(acc, ia, ia + TILE_BYTES, wm, wn, lane);
        if (more) {
            char *oa = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
            store_tile<A_KM>(sa, oa, tid);
            store_tile<B_KM>(sb, oa + TILE_BYTES, tid);
        }
        __syncthreads();
    }

    mfma_fence();
    if (nsplit > 1) {
        store_partial(acc, static_cast<float *>(g.workspace) + size_t(split) * g.M * g.N, g.M, g.N, m0, n0, wm, wn, lane);
        return;
    }

    // Epilogue through LDS (operand tiles are dead after the loop's final barrier).
    epilogue_tile<4, 4, HEAVY>(g, acc, reinterpret_cast<float *>(smem) + wave * (32 * 68), m0 + wm * 64, n0 + wn * 64, lane);
}

// actmask fallback (kernels without the fused form): bit (n & 7) of byte n >> 3 of row m = C[m, n] > 0; 16 columns per thread.
__global__ __launch_bounds__(256) void relu_bits_kernel(const uint16_t *__restrict__ c, int ldc, int M, int N,
                                                       uint8_t *__restrict__ mask, int ldm) {
    const int64_t v = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int nv = N >> 4;
    if (v >= int64_t(M) * nv) return;
    const int m = int(v / nv), n = int(v % nv) * 16;
    const uint16_t *p = c + size_t(m) * ldc + n;
    uint32_t bits = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) bits |= uint32_t(bf2f(p[r]) > 0.f) << r;
    *reinterpret_cast<uint16_t *>(mask + size_t(m) * ldm + (n >> 3)) = uint16_t(bits);
}

// C[m, n] = sum_z slab[z][m][n]; 4 columns per thread.
__global__ __launch_bounds__(256) void splitk_reduce(const float *__restrict__ ws, int splits, int M, int N, void *c,
                                                    int ldc, int c_is_f32) {
    const int64_t v = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int nv = N >> 2;
    if (v >= int64_t(M) * nv) return;
    const int m = int(v / nv), n = int(v % nv) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splits; z++) s += *reinterpret_cast<const f32x4 *>(ws + (size_t(z) * M + m) * N + n);
    if (c_is_f32) {
        *reinterpret_cast<f32x4 *>(static_cast<float *>(c) + size_t(m) * ldc + n) = s;
    } else {
        u32x2 o = {pack2bf(s[0], s[1]), pack2bf(s[2], s[3])};
        *reinterpret_cast<u32x2 *>(static_cast<uint16_t *>(c) + size_t(m) * ldc + n) = o;
    }
}

}  // namespace

int gemm256_dispatch(const sfcvit_gemm_args &a, int splits, int k_per_split, hipStream_t s);   // gemm256.hip
int gemm8p_dispatch(const sfcvit_gemm_args &a, int splits, hipStream_t s);                        // gemm8p.hip
int gemm8p_km_dispatch(const sfcvit_gemm_args &a, int splits_req, int *splits_used, int *k_done, hipStream_t s);

}  // namespace sfcvit

extern "C" int64_t sfcvit_gemm_workspace(int M, int N, int splitk) {
    if (splitk <= 1 || M <= 0 || N <= 0) return 0;
    const int64_t slabs = (int64_t(splitk) + 7) / 8 * 8;       // sfcvit_gemm rounds the split up to one set per XCD
    return slabs * M * N * int64_t(sizeof(float));
}

extern "C" int64_t sfcvit_gemm_colsum_workspace(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    const int64_t fused = (int64_t(M) / 96 + 2) * N * int64_t(sizeof(float));       // partial rows of the 8-phase epilogue: 2 per row tile, <= 2 ceil(M / 192)
    const int64_t separate = sfcvit_colsum_workspace(M, N);
    return fused > separate ? fused : separate;
}

static int gemm_impl(const sfcvit_gemm_args *a, void *stream);

extern "C" int sfcvit_gemm(const sfcvit_gemm_args *a, void *stream) {
    using namespace sfcvit;
    if (a && a->colsum_out) {
        if (a->c_is_f32 || a->splitk > 1) return fail(SFCVIT_EINVAL, "gemm: colsum_out needs a bf16 C and no split-K");
        if (!a->workspace || a->workspace_bytes < sfcvit_gemm_colsum_workspace(a->M, a->N) || !aligned16(a->workspace))
            return fail(SFCVIT_EINVAL, "gemm: colsum_out needs sfcvit_gemm_colsum_workspace(M, N) bytes of workspace");
    }
    if (a && a->actmask) {
        if (a->N % 16 || a->ld_actmask % 2 || a->ld_actmask * 8 < a->N || (reinterpret_cast<uintptr_t>(a->actmask) & 1))
            return fail(SFCVIT_EINVAL, "gemm: actmask needs N %% 16 == 0, an even ld_actmask >= N / 8 and 2-byte alignment");
        if (a->act != SFCVIT_ACT_RELU && a->dact != SFCVIT_ACT_RELU)
            return fail(SFCVIT_EINVAL, "gemm: actmask goes with act = RELU (written) or dact = RELU (read)");
        if (a->c_is_f32 || a->splitk > 1) return fail(SFCVIT_EINVAL, "gemm: actmask not with fp32 C or split-K");
    }
    if (int rc = gemm_impl(a, stream)) return rc;
    if (a->actmask && a->act == SFCVIT_ACT_RELU && !gemm_fused_actmask()) {      // the kernel that ran does not write the bits
        const int64_t nv = int64_t(a->M) * (a->N / 16);
        hipLaunchKernelGGL(relu_bits_kernel, dim3(unsigned((nv + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const uint16_t *>(a->c), a->ldc, a->M, a->N, static_cast<uint8_t *>(a->actmask), a->ld_actmask);
        if (int rc = check_launch("gemm actmask pass")) return rc;
    }
    if (a->colsum_out && !gemm_fused_colsum())   // the kernel that ran had no fused column sums: one pass over the stored C
        return sfcvit_colsum(a->c, a->M, a->N, a->ldc, a->colsum_out, a->colsum_bf16, a->workspace, a->workspace_bytes, stream);
    return SFCVIT_OK;
}

static int gemm_impl(const sfcvit_gemm_args *a, void *stream) {
    using namespace sfcvit;
    if (!a || !a->a || !a->b || !a->c) return fail(SFCVIT_EINVAL, "gemm: null operand");
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return fail(SFCVIT_EINVAL, "gemm: M=%d N=%d K=%d", a->M, a->N, a->K);
    // 16-byte vectors along the contiguous dimension of every operand.
    if (a->K % 8 != 0 && (!a->a_kmajor || !a->b_kmajor))
        return fail(SFCVIT_EINVAL, "gemm: K=%d must be a multiple of 8 for a k-contiguous operand", a->K);
    if (a->a_kmajor && a->M % 8 != 0) return fail(SFCVIT_EINVAL, "gemm: M=%d must be a multiple of 8 for k-major A", a->M);
    if ((a->b_kmajor && a->N % 8 != 0) || a->N % 4 != 0)
        return fail(SFCVIT_EINVAL, "gemm: N=%d must be a multiple of 4 (8 for k-major B)", a->N);
    if (a->lda % 8 || a->ldb % 8 || a->ldc % 4) return fail(SFCVIT_EINVAL, "gemm: lda=%d ldb=%d ldc=%d alignment", a->lda, a->ldb, a->ldc);
    if (!aligned16(a->a) || !aligned16(a->b) || !aligned16(a->c)) return fail(SFCVIT_EINVAL, "gemm: operands must be 16-byte aligned");
    if (a->residual && (a->ldr % 4 || (reinterpret_cast<uintptr_t>(a->residual) & 7)))
        return fail(SFCVIT_EINVAL, "gemm: residual alignment");
    if ((a->aux_in || a->aux_out) && a->ldaux % 4) return fail(SFCVIT_EINVAL, "gemm: ldaux=%d alignment", a->ldaux);
    if (a->dact != SFCVIT_ACT_NONE && !a->aux_in) return fail(SFCVIT_EINVAL, "gemm: dact needs aux_in");
    if (a->act < 0 || a->act > 2 || a->dact < 0 || a->dact > 2) return fail(SFCVIT_EINVAL, "gemm: bad act/dact");
    if (a->bias && (reinterpret_cast<uintptr_t>(a->bias) & 7)) return fail(SFCVIT_EINVAL, "gemm: bias alignment");
    if (!(a->dropout_p >= 0.f && a->dropout_p < 1.f)) return fail(SFCVIT_EINVAL, "gemm: dropout_p=%g out of [0, 1)", a->dropout_p);
    int splits = a->splitk < 1 ? 1 : a->splitk;
    const int ktiles = (a->K + BK - 1) / BK;
    if (splits > ktiles) splits = ktiles;
    const int out_tiles = ((a->N + BN - 1) / BN) * ((a->M + BM - 1) / BM);
    const bool per_xcd = splits > 1 && out_tiles < 64;          // one set of k-ranges per XCD (see gemm_kernel)
    if (per_xcd) splits = (splits + 7) / 8 * 8;
    int k_per_split = ((ktiles + splits - 1) / splits) * BK;
    if (!per_xcd) splits = (a->K + k_per_split - 1) / k_per_split;   // drop empty trailing ranges
    if (splits > 1) {
        if (a->bias || a->residual || a->aux_out || a->act || a->dact || a->dropout_p > 0.f)
            return fail(SFCVIT_EINVAL, "gemm: split-K supports no epilogue");
        const int64_t need = int64_t(splits) * a->M * a->N * int64_t(sizeof(float));
        if (!a->workspace || a->workspace_bytes < need || !aligned16(a->workspace))
            return fail(SFCVIT_EINVAL, "gemm: split-K workspace too small (%lld bytes needed; use sfcvit_gemm_workspace)", (long long)need);
    }

    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a->force_generic == 0 || (a->force_generic >= 8 && a->force_generic <= 10)) {
        int p8 = gemm8p_dispatch(*a, splits, s);
        if (p8 >= 0) return p8;
        int used = 0, k_done = 0;
        p8 = gemm8p_km_dispatch(*a, splits, &used, &k_done, s);
        if (p8 > 0) return p8;
        if (p8 == 0 && k_done < a->K) {
            // the last K % 128 rows of both k-major operands: one more fp32 slab from the generic kernel, summed with the
            // others in the same fixed order
            sfcvit_gemm_args t = *a;
            t.a = static_cast<const uint16_t *>(a->a) + size_t(k_done) * a->lda;
            t.b = static_cast<const uint16_t *>(a->b) + size_t(k_done) * a->ldb;
            t.K = a->K - k_done;
            t.c = static_cast<float *>(a->workspace) + size_t(used) * a->M * a->N;
            t.ldc = a->N;
            t.c_is_f32 = 1;
            t.splitk = 1;
            t.workspace = nullptr;
            t.workspace_bytes = 0;
            t.force_generic = 1;
            if (int rc = gemm_impl(&t, stream)) return rc;
            {   // what ran is the 8-phase weight-gradient kernel (+ its tail)
                const char *e = getenv("SFCVIT_GEMM_2PHASE");
                note_gemm_kernel(2, !(e && e[0] == '0'));
            }
            used++;
        }
        if (p8 == 0) {
            const int64_t nvec = int64_t(a->M) * (a->N / 4);
            hipLaunchKernelGGL(splitk_reduce, dim3(unsigned((nvec + 255) / 256)), dim3(256), 0, s,
                               static_cast<const float *>(a->workspace), used, a->M, a->N, a->c, a->ldc, a->c_is_f32);
            return check_launch("gemm splitk_reduce");
        }
        if (a->force_generic != 0) return fail(SFCVIT_EINVAL, "gemm: shape / options not eligible for the persistent 8-phase kernel");
    }
    const int big = (a->force_generic == 1) ? -1 : gemm256_dispatch(*a, splits, k_per_split, s);
    if (big > 0) return big;
    dim3 grid(((a->N + BN - 1) / BN) * ((a->M + BM - 1) / BM) * splits), block(THREADS);
    const size_t lds = 4 * TILE_BYTES;
    const bool heavy = a->act == SFCVIT_ACT_GELU || a->dact == SFCVIT_ACT_GELU;
#define SFCVIT_GO(AK, BK)                                                                                   \
    do {                                                                                                    \
        if (heavy) hipLaunchKernelGGL((gemm_kernel<AK, BK, true>), grid, block, lds, s, *a, k_per_split, splits);   \
        else hipLaunchKernelGGL((gemm_kernel<AK, BK, false>), grid, block, lds, s, *a, k_per_split, splits);        \
    } while (0)
    if (big != 0) note_gemm_kernel(4, a->a_kmajor != 0, a->b_kmajor != 0, heavy);
    if (big == 0) {
    } else if (!a->a_kmajor && !a->b_kmajor) SFCVIT_GO(false, false);
    else if (!a->a_kmajor && a->b_kmajor) SFCVIT_GO(false, true);
    else if (a->a_kmajor && !a->b_kmajor) SFCVIT_GO(true, false);
    else SFCVIT_GO(true, true);
#undef SFCVIT_GO
    if (int rc = check_launch("gemm")) return rc;
    if (splits > 1) {
        const int64_t nvec = int64_t(a->M) * (a->N / 4);
        hipLaunchKernelGGL(splitk_reduce, dim3(unsigned((nvec + 255) / 256)), dim3(256), 0, s,
                           static_cast<const float *>(a->workspace), splits, a->M, a->N, a->c, a->ldc, a->c_is_f32);
        return check_launch("gemm splitk_reduce");
    }
    return SFCVIT_OK;
}
