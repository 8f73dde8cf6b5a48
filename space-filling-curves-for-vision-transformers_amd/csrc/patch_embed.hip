// Fused SFC gather + patchify + linear projection for gfx950 (see
// sfcvit_patch_embed_fwd / _bwd in include/sfcvit.h).
//
// The reference does this in three passes over the image (advanced-index gather,
// reshape copy, GEMM; hilbert_embedding1D.py:36-43) and keeps the gathered
// [B, N, P*C] tensor for backward.  Here the gather is the A-operand loader of the
// MFMA GEMM: token rows are assembled in registers from the image through the
// per-token pixel table and written straight into the LDS tile, so the token
// matrix never exists in HBM; backward re-gathers it the same way.
//
// The contraction index is reordered channel-major (f' = c*P + kk instead of the
// reference's kk*C + c) so that 8 consecutive features are 8 consecutive curve
// positions of one channel: one 32-byte read of the pixel table + 8 image reads
// per LDS vector.  W is permuted to that order once per call into the workspace
// (D*K bf16, L2-resident); dW is permuted back by the split-K reduction.
#include "common_host.h"
#include "gemm_core.h"

namespace sfcvit {
namespace {

using namespace gemm_core;

struct Geo {
    const void *x;
    const int32_t *pix;
    int B, C, HW, N, P, M, K, Kp;   // M = B*N token rows, K = C*P features, Kp = K rounded up to 8 (zero padded)
};

// 8 consecutive channel-major features f..f+7 of token row m, as bf16x8 bits.
template <bool XBF16>
__device__ __forceinline__ u32x4 gather8(const Geo &g, int m, int f) {
    const int b = m / g.N, t = m - b * g.N;
    float v[8];
    if ((g.P & 7) == 0) {
        const int c = f / g.P, kk = f - c * g.P;
        const int32_t *pp = g.pix + size_t(t) * g.P + kk;
        const u32x4 i0 = *reinterpret_cast<const u32x4 *>(pp), i1 = *reinterpret_cast<const u32x4 *>(pp + 4);
        const uint32_t idx[8] = {i0[0], i0[1], i0[2], i0[3], i1[0], i1[1], i1[2], i1[3]};
        const size_t plane = (size_t(b) * g.C + c) * g.HW;
#pragma unroll
        for (int j = 0; j < 8; j++)
            v[j] = XBF16 ? bf2f(static_cast<const uint16_t *>(g.x)[plane + idx[j]])
                         : static_cast<const float *>(g.x)[plane + idx[j]];
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int ff = f + j;
            float val = 0.f;
            if (ff < g.K) {
                const int c = ff / g.P, kk = ff - c * g.P;
                const size_t off = (size_t(b) * g.C + c) * g.HW + g.pix[size_t(t) * g.P + kk];
                val = XBF16 ? bf2f(static_cast<const uint16_t *>(g.x)[off]) : static_cast<const float *>(g.x)[off];
            }
            v[j] = val;
        }
    }
    return u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
}

// Token tile for the forward GEMM: 128 token rows x 64 features ("kc" image).
template <bool XBF16>
__device__ __forceinline__ void load_tokens_kc(Stage &s, const Geo &g, int m0, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = tid + THREADS * i;
        const int m = m0 + (v >> 3), f = k0 + ((v & 7) << 3);
        s.v[i] = (m < g.M && f < g.Kp) ? gather8<XBF16>(g, m, f) : u32x4{0u, 0u, 0u, 0u};
    }
}

// Token tile for the weight-gradient GEMM: 64 token rows (contraction) x 128 features ("st" image).
template <bool XBF16>
__device__ __forceinline__ void load_tokens_st(Stage &s, const Geo &g, int f0, int m0, int mend, int tid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = tid + THREADS * i;
        const int m = m0 + (v >> 4), f = f0 + ((v & 15) << 3);
        s.v[i] = (m < mend && f < g.Kp) ? gather8<XBF16>(g, m, f) : u32x4{0u, 0u, 0u, 0u};
    }
}

// W'[d][c*P + kk] = W[d][kk*C + c]; rows padded with zeros to Kp features
__global__ __launch_bounds__(256) void permute_w_kernel(const uint16_t *__restrict__ w, uint16_t *__restrict__ wp, int D,
                                                        int C, int P, int Kp) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int K = C * P;
    if (i >= int64_t(D) * Kp) return;
    const int d = int(i / Kp), f = int(i % Kp);
    const int c = f / P, kk = f % P;
    wp[i] = f < K ? w[size_t(d) * K + kk * C + c] : uint16_t(0);
}

// y[m, d] = sum_f' tokens'[m, f'] W'[d, f'] + bias[d]
template <bool XBF16>
__global__ __launch_bounds__(THREADS, 2) void pe_fwd_kernel(const Geo g, const uint16_t *__restrict__ wp,
                                                            const uint16_t *__restrict__ bias, uint16_t *__restrict__ y,
                                                            int D) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (D + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int nk = (g.Kp + BK - 1) / BK;
    f32x4 acc[4][4];
    zero_acc(acc);
    Stage sa, sb;
    load_tokens_kc<XBF16>(sa, g, m0, 0, tid);
    load_tile<false>(sb, wp, g.Kp, n0, D, 0, g.Kp, tid);
    store_tile<false>(sa, smem, tid);
    store_tile<false>(sb, smem + TILE_BYTES, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const char *ia = smem + (kt & 1) * 2 * TILE_BYTES;
        const bool more = kt + 1 < nk;
        if (more) {
            load_tokens_kc<XBF16>(sa, g, m0, (kt + 1) * BK, tid);
            load_tile<false>(sb, wp, g.Kp, n0, D, (kt + 1) * BK, g.Kp, tid);
        }
        mma_tile<false, false>(acc, ia, ia + TILE_BYTES, wm, wn, lane);
        if (more) {
            char *oa = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
            store_tile<false>(sa, oa, tid);
            store_tile<false>(sb, oa + TILE_BYTES, tid);
        }
        __syncthreads();
    }
    mfma_fence();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
        if (n >= D) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
            const u32x2 b2 = *reinterpret_cast<const u32x2 *>(bias + n);
            bv[0] = bf2f(uint16_t(b2[0])); bv[1] = bf2f(uint16_t(b2[0] >> 16));
            bv[2] = bf2f(uint16_t(b2[1])); bv[3] = bf2f(uint16_t(b2[1] >> 16));
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = m0 + wm * 64 + i * 16 + (lane & 15);
            if (m >= g.M) continue;
            u32x2 o = {pack2bf(acc[i][j][0] + bv[0], acc[i][j][1] + bv[1]), pack2bf(acc[i][j][2] + bv[2], acc[i][j][3] + bv[3])};
            *reinterpret_cast<u32x2 *>(y + size_t(m) * D + n) = o;
        }
    }
}

// slab[z][d, f'] = sum_{m in split z} dY[m, d] tokens'[m, f']
template <bool XBF16>
__global__ __launch_bounds__(THREADS, 2) void pe_bwd_kernel(const Geo g, const uint16_t *__restrict__ dy,
                                                            float *__restrict__ slabs, int D, int m_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // 1-D grid, XCD-aware: logical index = (split z, feature tile, d tile) with the d tile fastest, and consecutive
    // logical indices on one XCD -- the D / 128 workgroups that re-gather the same token tile then share it in that
    // XCD's L2 (with a plain 3-D grid they were dealt round-robin over the XCDs: 1.79 GB of HBM-side traffic per launch
    // for 0.23 GB of operands, rocprofv3 FETCH_SIZE).
    const int tiles_d = (D + BM - 1) / BM, tiles_f = (g.Kp + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int dz = lid % tiles_d, fz = (lid / tiles_d) % tiles_f, zz = lid / (tiles_d * tiles_f);
    const int d0 = dz * BM, f0 = fz * BN;
    const int mbeg = zz * m_per_split, mend = min(g.M, mbeg + m_per_split);
    const int nk = (mend - mbeg + BK - 1) / BK;
    f32x4 acc[4][4];
    zero_acc(acc);
    Stage sa, sb;
    load_tile<true>(sa, dy, D, d0, D, mbeg, mend, tid);
    load_tokens_st<XBF16>(sb, g, f0, mbeg, mend, tid);
    store_tile<true>(sa, smem, tid);
    store_tile<true>(sb, smem + TILE_BYTES, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const char *ia = smem + (kt & 1) * 2 * TILE_BYTES;
        const bool more = kt + 1 < nk;
        if (more) {
            load_tile<true>(sa, dy, D, d0, D, mbeg + (kt + 1) * BK, mend, tid);
            load_tokens_st<XBF16>(sb, g, f0, mbeg + (kt + 1) * BK, mend, tid);
        }
        mma_tile<true, true>(acc, ia, ia + TILE_BYTES, wm, wn, lane);
        if (more) {
            char *oa = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
            store_tile<true>(sa, oa, tid);
            store_tile<true>(sb, oa + TILE_BYTES, tid);
        }
        __syncthreads();
    }
    mfma_fence();
    store_partial(acc, slabs + size_t(zz) * D * g.Kp, D, g.Kp, d0, f0, wm, wn, lane);
}

// dW[d][kk*C + c] = sum_z slab[z][d][c*P + kk].  A 1024-thread block owns 16 consecutive slab elements; its 64 thread
// rows each add every 64th slab (a narrow tokenizer level -- K = 48, D = 256 -- has 2 output tiles and therefore 512
// slabs: one thread per element summed them serially in 119 us), then 4 thread rows add 16 sub-sums each and one adds
// those 4.  Fixed order: bit-reproducible.
__global__ __launch_bounds__(1024) void pe_bwd_reduce(const float *__restrict__ slabs, int splits, float *__restrict__ dw,
                                                      int D, int C, int P, int Kp) {
    __shared__ float red[64][17];
    __shared__ float red2[4][16];
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t i = int64_t(blockIdx.x) * 16 + cl, total = int64_t(D) * Kp;
    float s = 0.f;
    if (i < total)
        for (int z = grp; z < splits; z += 64) s += slabs[size_t(z) * total + i];
    red[grp][cl] = s;
    __syncthreads();
    if (grp < 4) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) t += red[grp * 16 + k][cl];
        red2[grp][cl] = t;
    }
    __syncthreads();
    if (grp == 0 && i < total) {
        const int K = C * P, d = int(i / Kp), f = int(i % Kp);
        if (f < K) {
            const int c = f / P, kk = f % P;
            dw[size_t(d) * K + kk * C + c] = (red2[0][cl] + red2[1][cl]) + (red2[2][cl] + red2[3][cl]);
        }
    }
}

int bwd_splits(int M, int D, int K) {
    const int tiles = ((D + BM - 1) / BM) * ((K + BN - 1) / BN);
    int splits = (1024 + tiles - 1) / tiles;
    const int ktiles = (M + BK - 1) / BK;
    if (splits > ktiles) splits = ktiles;
    if (splits < 1) splits = 1;
    return splits;
}

int check_args(const sfcvit_patch_embed_args *a, const char *what) {
    if (!a || !a->x || !a->pix || !a->y) return fail(SFCVIT_EINVAL, "%s: null pointer", what);
    if (a->B <= 0 || a->C <= 0 || a->HW <= 0 || a->N <= 0 || a->P <= 0 || a->D <= 0)
        return fail(SFCVIT_EINVAL, "%s: non-positive dimension", what);
    if (int64_t(a->N) * a->P != a->HW) return fail(SFCVIT_EINVAL, "%s: N*P=%lld must equal H*W=%d", what, (long long)a->N * a->P, a->HW);
    if (a->D % 8 != 0) return fail(SFCVIT_EINVAL, "%s: D=%d must be a multiple of 8", what, a->D);
    if (int64_t(a->B) * a->N > 0x7fffffff / 2) return fail(SFCVIT_EINVAL, "%s: B*N too large", what);
    if (!aligned16(a->x) || !aligned16(a->pix) || !aligned16(a->y)) return fail(SFCVIT_EINVAL, "%s: alignment", what);
    return SFCVIT_OK;
}

Geo make_geo(const sfcvit_patch_embed_args *a) {
    return Geo{a->x, a->pix, a->B, a->C, a->HW, a->N, a->P, a->B * a->N, a->C * a->P, (a->C * a->P + 7) / 8 * 8};
}

}  // namespace
}  // namespace sfcvit

namespace sfcvit {
int pe2_fwd(const sfcvit_patch_embed_args &a, hipStream_t s);      // patch_embed_tiled.hip; -1 = not eligible
int pe2_bwd(const sfcvit_patch_embed_args &a, hipStream_t s);
int64_t pe2_bwd_workspace(int B, int C, int N, int D);
}

using namespace sfcvit;

// ---------------------------------------------------------------------------------------------------------------------
// The gather alone (sfcvit_tokens_gather): one workgroup per (token n, group of IMG images).  The token's P pixel offsets
// are read once; per image every thread loads its pixels' C channel values (4-byte / 2-byte loads that between them
// use every byte of the token's image lines), rounds to bf16 into an LDS row in the reference's feature order
// kk * C + c, and the row leaves as 16-byte stores.
// ---------------------------------------------------------------------------------------------------------------------
namespace sfcvit {
namespace {
constexpr int GT = 256, GIMG = 8;

template <bool XBF16>
__global__ __launch_bounds__(GT) void tokens_gather_kernel(const void *__restrict__ x, const int32_t *__restrict__ pix,
                                                           uint16_t *__restrict__ tokens, int B, int C, int HW, int N, int P, int ld) {
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    uint16_t *row = reinterpret_cast<uint16_t *>(smem_g);                 // [ld]
    const int n = blockIdx.x, b0 = blockIdx.y * GIMG, tid = threadIdx.x, K = P * C;
    for (int i = K + tid; i < ld; i += GT) row[i] = 0;                     // padding columns
    for (int b = b0; b < min(B, b0 + GIMG); b++) {
        const size_t img = size_t(b) * C * HW;
        for (int kk = tid; kk < P; kk += GT) {
            const int off = pix[size_t(n) * P + kk];
            for (int c = 0; c < C; c++) {
                const size_t src = img + size_t(c) * HW + off;
                row[kk * C + c] = XBF16 ? static_cast<const uint16_t *>(x)[src] : f2bf(static_cast<const float *>(x)[src]);
            }
        }
        __syncthreads();
        uint16_t *dst = tokens + (size_t(b) * N + n) * ld;
        for (int v = tid; v < ld / 8; v += GT) *reinterpret_cast<u32x4 *>(dst + v * 8) = *reinterpret_cast<const u32x4 *>(row + v * 8);
        __syncthreads();
    }
}

// P <= 256 (every reference tokenizer: 16, 64 or 256 pixels per token), C <= 4: one pixel per thread, the loads of all GIMG
// images in flight at once, one barrier per workgroup -- and TWO tokens per workgroup, taken from `order`: a 16 x 16 tile's
// rows are 64 bytes of fp32, half a 128-byte line whose other half belongs to the tile next to it; with one token per
// workgroup every line of the image was fetched twice (62 us at ViT-B / 256 images: 154 MB of image read as 308).  The
// caller orders the tokens by their lowest pixel offset, which puts horizontal neighbours side by side.
template <bool XBF16, int CMAX>
__global__ __launch_bounds__(2 * GT) void tokens_gather_p256_kernel(const void *__restrict__ x, const int32_t *__restrict__ pix,
                                                                    const int32_t *__restrict__ order, uint16_t *__restrict__ tokens,
                                                                    int B, int C, int HW, int N, int P, int ld) {
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    const int half = threadIdx.x >> 8, tid = threadIdx.x & (GT - 1);
    const int slot = 2 * blockIdx.x + half;
    uint16_t *rows = reinterpret_cast<uint16_t *>(smem_g) + size_t(half) * GIMG * ld;   // [2][GIMG][ld]
    const int b0 = blockIdx.y * GIMG, K = P * C;
    const int nb = min(GIMG, B - b0);
    const bool live = slot < N;
    const int n = live ? (order ? order[slot] : slot) : 0;
    uint16_t v[GIMG][CMAX];
    if (live && tid < P) {
        const int off = pix[size_t(n) * P + tid];
#pragma unroll
        for (int i = 0; i < GIMG; i++) {
            if (i < nb) {
                const size_t img = size_t(b0 + i) * C * HW + off;
#pragma unroll
                for (int c = 0; c < CMAX; c++)
                    if (c < C) v[i][c] = XBF16 ? static_cast<const uint16_t *>(x)[img + size_t(c) * HW] : f2bf(static_cast<const float *>(x)[img + size_t(c) * HW]);
            }
        }
#pragma unroll
        for (int i = 0; i < GIMG; i++)
            if (i < nb)
#pragma unroll
                for (int c = 0; c < CMAX; c++)
                    if (c < C) rows[i * ld + tid * C + c] = v[i][c];
    }
    if (live)
        for (int i = K + tid; i < ld; i += GT)
            for (int r = 0; r < nb; r++) rows[r * ld + i] = 0;
    __syncthreads();
    if (!live) return;
    const int vpr = ld / 8;
    for (int j = tid; j < nb * vpr; j += GT) {
        const int r = j / vpr, vv = j - r * vpr;
        *reinterpret_cast<u32x4 *>(tokens + (size_t(b0 + r) * N + n) * ld + vv * 8) = *reinterpret_cast<const u32x4 *>(rows + r * ld + vv * 8);
    }
}


// Tokens that are 16 x 16 pixel tiles of an fp32 image (every Hilbert / Z tokenizer at 256 pixels per token): the image is
// read in WHOLE 128-byte lines -- a line is one tile row of two horizontally adjacent tiles -- with 16-byte loads, eight
// lines per wave instruction, and the curve order is applied on the way OUT of LDS.  One wave owns a unit = (tile pair,
// image): 2 C wave loads, the bf16 pair image [C][16 rows][32 pixels] in a wave-private LDS region, then every lane
// collects the 8 features of each of its C output vectors with 2-byte LDS reads at addresses worked out once per
// workgroup (feature j of token t: channel j % C of curve pixel j / C), and the two token rows leave as contiguous
// 16-byte stores.  Against the per-pixel kernel above: a quarter of the load instructions, half the lines touched per
// byte (a 4-byte-per-lane load of 64 curve-consecutive pixels touches 8 lines for 256 bytes), no workgroup barrier
// between a unit's loads and its stores.  Workgroup = 4 waves x U images (U = 1: 78 registers, six workgroups per CU, 6 272
// short workgroups at ViT-B / 256 images -- the tail of the launch is what U = 2 lost: 45.7 vs 44.0 us in the step); the
// image is read and the tokens are written with nontemporal accesses (see the launch code).  A third form -- one workgroup
// per 16-row strip of the image, read front to back -- was built and dropped: 56 us warm against 38 (two barriers per image,
// 132 registers).
constexpr int TG_THREADS = 256;

template <int C, int NT, int U>
__global__ __launch_bounds__(TG_THREADS) void tokens_gather_tiles_kernel(const float *__restrict__ x, const int32_t *__restrict__ pix,
                                                                         const int32_t *__restrict__ order, const int32_t *__restrict__ origin,
                                                                         uint16_t *__restrict__ tokens, int B, int HW, int W,
                                                                         uint32_t wmagic, int N) {
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    uint16_t *pos = reinterpret_cast<uint16_t *>(smem_g);                  // [2 tokens][256 curve pixels] -> pixel of the pair image
    constexpr int UNIT = C * 16 * 64;                                      // bytes of one bf16 pair image
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int slot = 2 * blockIdx.x;
    const bool two = slot + 1 < N;
    const int n0 = order ? order[slot] : slot;
    const int n1 = two ? (order ? order[slot + 1] : slot + 1) : n0;
    const int o0 = origin[n0], o1 = origin[n1];
    for (int i = tid; i < 512; i += TG_THREADS) {
        const int t = i >> 8, kk = i & 255;
        const uint32_t d = uint32_t(pix[size_t(t ? n1 : n0) * 256 + kk] - (t ? o1 : o0));     // < 16 W
        const uint32_t r = __umulhi(d, wmagic);                           // d / W: exact for d W < 2^32
        pos[i] = uint16_t(r * 32 + 16 * t + (d - r * W));
    }
    // this wave's two images: every line of the pair's rows, all loads in flight
    const int b0 = blockIdx.y * (4 * U) + U * wave;
    const int chunk = lane & 7, lt = chunk >> 2;
    const size_t src0 = size_t(lt ? o1 : o0) + 4 * (chunk & 3);
    f32x4 v[U][2 * C];
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (b0 + u < B) {
            const float *im = x + size_t(b0 + u) * C * HW + src0;
#pragma unroll
            for (int i = 0; i < 2 * C; i++) {
                const int line = 8 * i + (lane >> 3);
                const f32x4 *src = reinterpret_cast<const f32x4 *>(im + size_t(line >> 4) * HW + size_t(line & 15) * W);
                v[u][i] = (NT & 1) ? __builtin_nontemporal_load(src) : *src;
            }
        }
    }
    __syncthreads();
    // byte addresses of the features of this lane's output vectors (vector vv of the pair = lane + 64 i: token vv / (32 C),
    // features 8 (vv % (32 C)) ... + 8)
    uint32_t addr[C][8];
#pragma unroll
    for (int i = 0; i < C; i++) {
        const int vv = lane + 64 * i, t = vv / (32 * C), j0 = (vv - t * 32 * C) * 8;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int j = j0 + e, kk = j / C, c = j - kk * C;
            addr[i][e] = uint32_t(c) * 1024 + 2 * uint32_t(pos[t * 256 + kk]);
        }
    }
    char *mine = smem_g + 1024 + size_t(U * wave) * UNIT;
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (b0 + u >= B) break;                                            // wave-uniform
        char *im = mine + u * UNIT;
#pragma unroll
        for (int i = 0; i < 2 * C; i++) {
            const int line = 8 * i + (lane >> 3);
            u32x2 pk;
            pk[0] = pack2bf(v[u][i][0], v[u][i][1]);
            pk[1] = pack2bf(v[u][i][2], v[u][i][3]);
            *reinterpret_cast<u32x2 *>(im + line * 64 + chunk * 8) = pk;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < C; i++) {
            const int vv = lane + 64 * i, t = vv / (32 * C), j0 = (vv - t * 32 * C) * 8;
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t lo = *reinterpret_cast<const uint16_t *>(im + addr[i][2 * q]);
                const uint32_t hi = *reinterpret_cast<const uint16_t *>(im + addr[i][2 * q + 1]);
                o[q] = lo | (hi << 16);
            }
            if (t == 0 || two) {
                u32x4 *dst = reinterpret_cast<u32x4 *>(tokens + (size_t(b0 + u) * N + (t ? n1 : n0)) * (256 * C) + j0);
                if (NT & 2) __builtin_nontemporal_store(o, dst);
                else *dst = o;
            }
        }
    }
}

}  // namespace
}  // namespace sfcvit

extern "C" int sfcvit_tokens_gather_tiles(const void *x, const int32_t *pix, const int32_t *order, const int32_t *origin, int B, int C,
                                          int H, int W, int N, void *tokens, int ld, void *stream) {
    using namespace sfcvit;
    if (!x || !pix || !origin || !tokens) return fail(SFCVIT_EINVAL, "tokens_gather_tiles: null pointer");
    if (B <= 0 || C < 1 || C > 4 || H <= 0 || W <= 0 || N <= 0 || int64_t(N) * 256 != int64_t(H) * W || (W & 7) || (H & 15))
        return fail(SFCVIT_EINVAL, "tokens_gather_tiles: B=%d C=%d H=%d W=%d N=%d (1 <= C <= 4, W %% 8 == 0, H %% 16 == 0, N * 256 == H * W)", B, C, H, W, N);
    if (ld != 256 * C) return fail(SFCVIT_EINVAL, "tokens_gather_tiles: ld=%d (must be 256 * C = %d)", ld, 256 * C);
    if (!aligned16(x) || !aligned16(tokens)) return fail(SFCVIT_EINVAL, "tokens_gather_tiles: x and tokens must be 16-byte aligned");
    if (int64_t(16) * W * W >= (int64_t(1) << 32)) return fail(SFCVIT_EINVAL, "tokens_gather_tiles: W=%d too wide", W);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const auto *xp = static_cast<const float *>(x);
    auto *tp = static_cast<uint16_t *>(tokens);
    const int HW = H * W;
    static const int upw = [] { const char *e = getenv("SFCVIT_GATHER_U"); return e && atoi(e) == 2 ? 2 : 1; }();   // images per wave (A/B: 2 = fewer, longer workgroups)
    const dim3 grid((N + 1) / 2, (B + 4 * upw - 1) / (4 * upw));
    if (grid.y > 65535) return fail(SFCVIT_EINVAL, "tokens_gather_tiles: batch %d too large", B);
    const uint32_t wmagic = uint32_t((uint64_t(1) << 32) / uint32_t(W)) + 1;
    const size_t lds = 1024 + size_t(4 * upw) * C * 1024;
    // nontemporal loads (1) and stores (2).  The gather is the first kernel of a step: L2 and the memory-side cache are full
    // of the optimizer's dirty lines, and every line an ordinary load allocates evicts one of them -- a write-back that
    // shares HBM with the gather (75.8 us against 42.1 us with clean caches; a plain fp32 -> bf16 cast of the image suffers
    // the same: 69 vs 37 us).  Streaming loads do not allocate: 46.0 us; with streaming stores 43.2 us alone, 44.0 us in
    // the step (profiles/r4/gather_ab.txt).  SFCVIT_GATHER_NT=0..3 for the A/B.
    static const int nt = [] { const char *e = getenv("SFCVIT_GATHER_NT"); return e ? atoi(e) : 3; }();
#define TILES(CC, NTV) do { if (upw == 2) hipLaunchKernelGGL((tokens_gather_tiles_kernel<CC, NTV, 2>), grid, dim3(TG_THREADS), lds, s, xp, pix, order, origin, tp, B, HW, W, wmagic, N); \
                            else hipLaunchKernelGGL((tokens_gather_tiles_kernel<CC, NTV, 1>), grid, dim3(TG_THREADS), lds, s, xp, pix, order, origin, tp, B, HW, W, wmagic, N); } while (0)
#define TILES_C(CC) do { if (nt == 1) TILES(CC, 1); else if (nt == 2) TILES(CC, 2); else if (nt == 3) TILES(CC, 3); else TILES(CC, 0); } while (0)
    switch (C) {
    case 1: TILES_C(1); break;
    case 2: TILES_C(2); break;
    case 3: TILES_C(3); break;
    default: TILES_C(4); break;
    }
#undef TILES_C
#undef TILES
    return check_launch("tokens_gather_tiles");
}

extern "C" int sfcvit_tokens_gather(const void *x, int x_is_bf16, const int32_t *pix, const int32_t *order, int B, int C, int HW, int N,
                                    int P, void *tokens, int ld, void *stream) {
    using namespace sfcvit;
    if (!x || !pix || !tokens) return fail(SFCVIT_EINVAL, "tokens_gather: null pointer");
    if (B <= 0 || C <= 0 || HW <= 0 || N <= 0 || P <= 0 || int64_t(N) * P != HW)
        return fail(SFCVIT_EINVAL, "tokens_gather: B=%d C=%d HW=%d N=%d P=%d (N * P must equal H * W)", B, C, HW, N, P);
    if (ld < P * C || ld % 8 || ld > 32768) return fail(SFCVIT_EINVAL, "tokens_gather: ld=%d (>= P * C = %d, multiple of 8, <= 32768)", ld, P * C);
    if (!aligned16(tokens)) return fail(SFCVIT_EINVAL, "tokens_gather: tokens must be 16-byte aligned");
    const dim3 grid(N, (B + GIMG - 1) / GIMG);
    if (grid.y > 65535) return fail(SFCVIT_EINVAL, "tokens_gather: batch %d too large", B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (P <= GT && C <= 4 && size_t(2) * GIMG * ld * 2 <= 64 * 1024) {
        const dim3 grid2((N + 1) / 2, grid.y);
        const size_t lds = size_t(2) * GIMG * ld * 2;
        if (x_is_bf16)
            hipLaunchKernelGGL((tokens_gather_p256_kernel<true, 4>), grid2, dim3(2 * GT), lds, s, x, pix, order, static_cast<uint16_t *>(tokens), B, C, HW, N, P, ld);
        else
            hipLaunchKernelGGL((tokens_gather_p256_kernel<false, 4>), grid2, dim3(2 * GT), lds, s, x, pix, order, static_cast<uint16_t *>(tokens), B, C, HW, N, P, ld);
        return check_launch("tokens_gather");
    }
    if (x_is_bf16)
        hipLaunchKernelGGL(tokens_gather_kernel<true>, grid, dim3(GT), size_t(ld) * 2, s, x, pix, static_cast<uint16_t *>(tokens), B, C, HW, N, P, ld);
    else
        hipLaunchKernelGGL(tokens_gather_kernel<false>, grid, dim3(GT), size_t(ld) * 2, s, x, pix, static_cast<uint16_t *>(tokens), B, C, HW, N, P, ld);
    return check_launch("tokens_gather");
}

extern "C" int64_t sfcvit_patch_embed_workspace(int B, int C, int N, int P, int D, int bwd) {
    if (B <= 0 || C <= 0 || N <= 0 || P <= 0 || D <= 0) return 0;
    const int64_t K = (int64_t(C) * P + 7) / 8 * 8;
    if (!bwd) return ((int64_t(D) * K * 2 * 8 + 15) / 16) * 16;      // up to 8 class-permuted copies of W (tiled forward)
    const int64_t slabs = int64_t(bwd_splits(B * N, D, int(K))) * D * K * int64_t(sizeof(float));
    const int64_t bias_ws = sfcvit_colsum_workspace(B * N, D);   // dbias reuses the buffer after the slabs are reduced
    const int64_t tiled = (P == 256 && D % 256 == 0) ? pe2_bwd_workspace(B, C, N, D) : 0;
    const int64_t m = slabs > bias_ws ? slabs : bias_ws;
    return m > tiled ? m : tiled;
}

extern "C" int sfcvit_patch_embed_fwd(const sfcvit_patch_embed_args *a, void *stream) {
    if (int rc = check_args(a, "patch_embed_fwd")) return rc;
    if (!a->w) return fail(SFCVIT_EINVAL, "patch_embed_fwd: null weight");
    const int64_t need = sfcvit_patch_embed_workspace(a->B, a->C, a->N, a->P, a->D, 0);
    if (!a->workspace || a->workspace_bytes < need || !aligned16(a->workspace))
        return fail(SFCVIT_EINVAL, "patch_embed_fwd: workspace of %lld bytes needed", (long long)need);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = pe2_fwd(*a, s); rc >= 0) return rc;                 // tiles / strips: the coalesced kernel
    const Geo g = make_geo(a);
    uint16_t *wp = static_cast<uint16_t *>(a->workspace);
    const int64_t nw = int64_t(a->D) * g.Kp;
    hipLaunchKernelGGL(permute_w_kernel, dim3(unsigned((nw + 255) / 256)), dim3(256), 0, s,
                       static_cast<const uint16_t *>(a->w), wp, a->D, a->C, a->P, g.Kp);
    if (int rc = check_launch("patch_embed_fwd permute")) return rc;
    dim3 grid(((a->D + BN - 1) / BN) * ((g.M + BM - 1) / BM)), block(THREADS);
    const size_t lds = 4 * TILE_BYTES;
    if (a->x_is_bf16)
        hipLaunchKernelGGL(pe_fwd_kernel<true>, grid, block, lds, s, g, wp, static_cast<const uint16_t *>(a->bias),
                           static_cast<uint16_t *>(a->y), a->D);
    else
        hipLaunchKernelGGL(pe_fwd_kernel<false>, grid, block, lds, s, g, wp, static_cast<const uint16_t *>(a->bias),
                           static_cast<uint16_t *>(a->y), a->D);
    return check_launch("patch_embed_fwd");
}

extern "C" int sfcvit_patch_embed_bwd(const sfcvit_patch_embed_args *a, void *stream) {
    if (int rc = check_args(a, "patch_embed_bwd")) return rc;
    if (!a->dw) return fail(SFCVIT_EINVAL, "patch_embed_bwd: null dw");
    const int64_t need = sfcvit_patch_embed_workspace(a->B, a->C, a->N, a->P, a->D, 1);
    if (!a->workspace || a->workspace_bytes < need || !aligned16(a->workspace))
        return fail(SFCVIT_EINVAL, "patch_embed_bwd: workspace of %lld bytes needed", (long long)need);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = pe2_bwd(*a, s); rc >= 0) {                          // tiles / strips: the coalesced kernel
        if (rc || !a->dbias) return rc;
        return sfcvit_colsum(a->y, a->B * a->N, a->D, a->D, a->dbias, 0, a->workspace, a->workspace_bytes, stream);
    }
    const Geo g = make_geo(a);
    const int splits = bwd_splits(g.M, a->D, g.Kp);
    const int ktiles = (g.M + BK - 1) / BK;
    const int m_per_split = ((ktiles + splits - 1) / splits) * BK;
    const int zs = (g.M + m_per_split - 1) / m_per_split;
    float *slabs = static_cast<float *>(a->workspace);
    dim3 grid(((g.Kp + BN - 1) / BN) * ((a->D + BM - 1) / BM) * zs), block(THREADS);
    const size_t lds = 4 * TILE_BYTES;
    if (a->x_is_bf16)
        hipLaunchKernelGGL(pe_bwd_kernel<true>, grid, block, lds, s, g, static_cast<const uint16_t *>(a->y), slabs, a->D, m_per_split);
    else
        hipLaunchKernelGGL(pe_bwd_kernel<false>, grid, block, lds, s, g, static_cast<const uint16_t *>(a->y), slabs, a->D, m_per_split);
    if (int rc = check_launch("patch_embed_bwd")) return rc;
    const int64_t nw = int64_t(a->D) * g.Kp;
    hipLaunchKernelGGL(pe_bwd_reduce, dim3(unsigned((nw + 15) / 16)), dim3(1024), 0, s, slabs, zs,
                       static_cast<float *>(a->dw), a->D, a->C, a->P, g.Kp);
    if (int rc = check_launch("patch_embed_bwd reduce")) return rc;
    if (a->dbias)
        return sfcvit_colsum(a->y, g.M, a->D, a->D, a->dbias, 0, a->workspace, a->workspace_bytes, stream);
    return SFCVIT_OK;
}
