// Fused SFC gather + patchify + projection, TILED form (gfx950) -- the BASELINE tokenizers.
//
// At every BASELINE size (32, 224, 384 px; 256 curve pixels per token) each token of the Hilbert / Z order is ONE
// 16 x 16 pixel tile of the image, visited in the curve's order, with one of <= 4 pixel orders
// inside the tile ("classes"; 1 for Z); a raster token (zigzag_embedding1D.py:30-39) is a strip of 256 consecutive
// pixels.  So a token's 768 features are 48 contiguous row segments of 16 pixels, and the curve is two small tables:
// the tile origin per token and the intra-tile permutation per class.  The permutation is folded into the WEIGHT
// (W'[class][d][c*256 + r*16 + x] = W[d][curve_pos_class(r, x)*C + c], one tiny kernel per call), after which
//     tokens(b, n) . W^T  ==  image_tile(b, n) (raster order inside the tile) . W'[class(n)]^T            exactly,
// i.e. the same products summed in a different order.  The gather is then the A-operand loader of an MFMA GEMM:
//
//   * forward: one workgroup = 128 token rows (all of one class; rows = (token, image) pairs) x 256 output columns;
//     per k-tile (64 features = 4 tile rows of one channel) every thread loads ONE 16-pixel row segment (64 B fp32 /
//     32 B bf16, 16-byte vector loads: whole segments, fully used cache lines), converts it to bf16 and writes it into
//     the swizzled LDS tile; W' streams through a register-staged double buffer (LDS-DMA into a ring of three k-tiles
//     was measured: 174 vs 168 us, no gain -- the bound is what the CUs can take in, not how); 16x16x32 bf16 MFMA, fp32 accumulate; the
//     bias is added and the rows are stored straight to out[b, n, :].  The three column tiles of a row tile run
//     back to back on one XCD, so the image is fetched from HBM once and re-read from that XCD's L2.
//   * the per-pixel table of the generic kernel (patch_embed.hip: 8 scalar 2-byte loads per LDS vector, every
//     128-column tile of D re-gathering its tokens) is not touched on this path; tokenizers that are not tiles or
//     strips (SFCEmbedding1D with other p / g, Peano at 27 px, ...) keep using the generic kernel.
//
// Replaces HilbertEmbedding1D / MortonEmbedding1D / RasterScan1DEmbedding .forward
// (src/tokenizers/_1D/hilbert_embedding1D.py:30-44, morton_embedding1D.py:30-44, zigzag_embedding1D.py:30-39).
#include "common_host.h"
#include "device_common.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace sfcvit {
namespace {

constexpr int TM = 128, TN = 256, BK = 64, PT = 512;       // token rows, output columns, k per step, threads
constexpr int A_BYTES = TM * 128, B_BYTES = TN * 128;       // one LDS k-tile of each operand (128-byte rows)
using sfcvit::tile_desc::MAXCLS;
using sfcvit::tile_desc::DESC_HDR;      // layout: tile_descriptors.cpp

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// W'[cls][d][c*256 + j] = W[d][perm[cls][j]*C + c]   (j = pixel of the tile in raster order, perm = its curve position)
__global__ void pe2_permute_w_kernel(const uint16_t *__restrict__ w, const int32_t *__restrict__ perm, uint16_t *__restrict__ wp,
                                     int D, int C, int ncls) {
    const int K = C * 256;
    const int64_t total = int64_t(ncls) * D * K;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int f = int(i % K), d = int((i / K) % D), cls = int(i / (int64_t(K) * D));
        const int c = f >> 8, j = f & 255;
        wp[i] = w[size_t(d) * K + perm[cls * 256 + j] * C + c];
    }
}

struct RowInfo {              // one token row of the workgroup's tile
    long long src;            // element offset of the tile origin in x (b * C*HW + origin[n])
    int out_row;              // b * N + n, or -1 for a padding row
    int pad;
};

template <bool XBF16>
__global__ __launch_bounds__(PT) void pe2_fwd_kernel(const void *__restrict__ x, const int32_t *__restrict__ desc,
                                                     const uint16_t *__restrict__ wp, const uint16_t *__restrict__ bias,
                                                     uint16_t *__restrict__ y, int B, int C, int HW, int D, int n_row_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *abuf = smem, *bbuf = smem + 2 * A_BYTES;
    RowInfo *rows = reinterpret_cast<RowInfo *>(smem + 2 * A_BYTES + 2 * B_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 2, wc = wid & 3;
    const int K = C * 256, KT = K / BK, NCT = D / TN;
    // XCD-aware tile order: the NCT column tiles of a row tile run consecutively on one XCD (blocks b, b + 8, ... share one)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, ct = idx % NCT, rt = (idx / NCT) * 8 + xcd;
    if (rt >= n_row_tiles) return;
    const int mode_ncls = desc[1], kstep = desc[2], sstep = desc[3], N = desc[4];
    const int32_t *toks = desc + DESC_HDR, *origin = desc + DESC_HDR + N;
    // class of this row tile: every class is padded to whole tiles of TM rows
    int cls = 0, tile0 = 0, cnt = 0, tok0 = 0;
    for (int c = 0; c < mode_ncls; c++) {
        const int t0 = desc[6 + c], n_c = desc[6 + c + 1] - t0, tiles_c = (n_c * B + TM - 1) / TM;
        if (rt >= tile0 && rt < tile0 + tiles_c) { cls = c; cnt = n_c; tok0 = t0; break; }
        tile0 += tiles_c;
    }
    if (tid < TM) {
        const int ml = (rt - tile0) * TM + tid;                 // row within the class: (token index, image), image fastest
        const bool valid = ml < cnt * B;
        const int mc = valid ? ml : 0;
        const int n = toks[tok0 + mc / B], b = mc % B;
        rows[tid].src = (long long)(b) * C * HW + origin[n];
        rows[tid].out_row = valid ? b * N + n : -1;
    }
    __syncthreads();

    // --- staging through registers, TWO k-tiles deep ---------------------------------------------------------------
    // A: this thread's 16-pixel row segment (row tid >> 2, segment tid & 3) of every k-tile; B: four 16-byte pieces of
    // the W'[cls] panel.  Both are ordinary vector loads (no LDS-DMA), so hipcc's own counted vmcnt waits apply and a
    // __syncthreads() is a bare barrier: the loads of k-tile t + 2 are issued before the MFMAs of k-tile t and are
    // first touched two barriers later (128 KiB in flight per CU).  Measured at ViT-B / 256 images (profiles/r2): 160-168 us
    // = 5.8 TB/s of operands into the CUs with one or with two k-tiles in flight -- the kernel is bound by operand
    // delivery (786 KB per 128 x 256 tile: the fp32 segments and a W' panel that the streaming image pushes out of the
    // 4 MiB L2: 403 MB memory-side fetch for 154 MB of image), not by latency; the generic kernel took 338 us + a 32 us cast.
    const int arow = tid >> 2, aseg = tid & 3;
    const long long asrc = rows[arow].src + aseg * sstep;
    const int aw0 = kc_off(arow, 2 * aseg), aw1 = kc_off(arow, 2 * aseg + 1);
    const uint16_t *wbase = wp + (size_t(cls) * D + size_t(ct) * TN) * K;
    constexpr int NA = XBF16 ? 2 : 4;
    struct Stage { u32x4 a[NA]; u32x4 b[4]; };
    auto load_stage = [&](Stage &st, int kt) __attribute__((always_inline)) {
        const long long off = asrc + (long long)(kt >> 2) * HW + (kt & 3) * kstep;
        const u32x4 *pa = XBF16 ? reinterpret_cast<const u32x4 *>(static_cast<const uint16_t *>(x) + off)
                                : reinterpret_cast<const u32x4 *>(static_cast<const float *>(x) + off);
        // (measured and not kept: non-temporal loads here -- twice the L2 requests, 168 -> 359 us)
#pragma unroll
        for (int i = 0; i < NA; i++) st.a[i] = pa[i];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = tid + PT * i, row = p >> 3, cs = p & 7;
            st.b[i] = *reinterpret_cast<const u32x4 *>(wbase + size_t(row) * K + kt * BK + cs * 8);
        }
    };
    auto write_stage = [&](const Stage &st, int buf) __attribute__((always_inline)) {
        u32x4 lo, hi;
        if (XBF16) {
            lo = st.a[0]; hi = st.a[1];
        } else {
            const float *f = reinterpret_cast<const float *>(st.a);
            lo = u32x4{pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
            hi = u32x4{pack2bf(f[8], f[9]), pack2bf(f[10], f[11]), pack2bf(f[12], f[13]), pack2bf(f[14], f[15])};
        }
        *reinterpret_cast<u32x4 *>(abuf + buf * A_BYTES + aw0) = lo;
        *reinterpret_cast<u32x4 *>(abuf + buf * A_BYTES + aw1) = hi;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = tid + PT * i, row = p >> 3, cs = p & 7;
            *reinterpret_cast<u32x4 *>(bbuf + buf * B_BYTES + kc_off(row, cs)) = st.b[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int cur) __attribute__((always_inline)) {
        const char *ab = abuf + cur * A_BYTES + wr * 64 * 128, *bb = bbuf + cur * B_BYTES + wc * 64 * 128;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) fa[i] = kc_frag(ab, 16 * i, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; j++) fb[j] = kc_frag(bb, 16 * j, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    };

    Stage s0, s1;                                     // s0: even k-tiles, s1: odd k-tiles (KT = 4 C is even)
    load_stage(s0, 0);
    load_stage(s1, 1);
    write_stage(s0, 0);
    __syncthreads();
    for (int kt = 0; kt < KT; kt += 2) {
        // k-tile kt (buffer 0): s0 is free (written before the last barrier), s1 holds k-tile kt + 1
        if (kt + 2 < KT) load_stage(s0, kt + 2);
        compute(0);
        write_stage(s1, 1);
        __syncthreads();
        // k-tile kt + 1 (buffer 1): s1 is free, s0 holds k-tile kt + 2
        if (kt + 3 < KT) load_stage(s1, kt + 3);
        compute(1);
        if (kt + 2 < KT) write_stage(s0, 0);
        __syncthreads();
    }
    mfma_fence();
    // acc[i][j][r] = out[token row 64 wr + 16 i + (lane & 15)][column ct*TN + 64 wc + 16 j + 4 (lane >> 4) + r]
    const int g = lane >> 4, li = lane & 15, col0 = ct * TN + 64 * wc + 4 * g;
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) bv[j][r] = bias ? bf2f(bias[col0 + 16 * j + r]) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int orow = rows[64 * wr + 16 * i + li].out_row;
        if (orow < 0) continue;
        uint16_t *dst = y + size_t(orow) * D + col0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32x2 o = {pack2bf(acc[i][j][0] + bv[j][0], acc[i][j][1] + bv[j][1]),
                             pack2bf(acc[i][j][2] + bv[j][2], acc[i][j][3] + bv[j][3])};
            *reinterpret_cast<u32x2 *>(dst + 16 * j) = o;
        }
    }
}


// ----------------------------------------------------------------------------------------------------------------
// Backward: dW'[cls][d][c*256 + j] = sum over the class's token rows of dY[row][d] * tile_pixel[row][c][j], then
// dW[d][k*C + c] = sum over classes of dW'[cls][d][c*256 + inverse_perm_cls(k)].  One workgroup = one 256 (d) x 256
// (the 256 pixels of ONE channel) output tile over one range of the class's rows; per step of 64 rows every thread loads
// two 16-pixel row segments of its row's tile (whole segments, 16-byte vectors) and 64 bytes of its dY row, both go
// into k-major LDS images (device_common.h "st" layout) and are read as MFMA fragments by ds_read_b64_tr_b16 -- the
// loop of the weight-gradient GEMM (gemm8p.hip) with the gather as its B loader.  fp32 partial tiles go to slabs
// (deterministic fixed-order reduction, no atomics).
// ----------------------------------------------------------------------------------------------------------------
constexpr int BW_ROWS = 64, BW_IMG = BW_ROWS * 256;          // rows per step; one [64][128] st image

template <bool XBF16>
__global__ __launch_bounds__(PT) void pe2_bwd_kernel(const void *__restrict__ x, const int32_t *__restrict__ desc,
                                                     const uint16_t *__restrict__ dy, float *__restrict__ slabs,
                                                     int B, int C, int HW, int D, int KR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 stages x [dY half 0 | dY half 1 | X half 0 | X half 1]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 2, wc = wid & 3;
    const int ncls = desc[1], sstep = desc[3], N = desc[4], K = C * 256;
    const int32_t *toks = desc + DESC_HDR, *origin = desc + DESC_HDR + N;
    const int NDT = D / 256, tiles = NDT * C;
    // unit -> (class, row range z, d tile, channel).  The tiles of one row range (same image rows, same dY rows) run on
    // ONE XCD (blocks b, b + 8, ... share one): they re-read each other's operands from its L2 instead of from memory.
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    int u = ((idx / tiles) * 8 + xcd) * tiles + idx % tiles, cls = -1, cnt = 0, tok0 = 0, z = 0, ubase = 0;
    for (int c = 0; c < ncls; c++) {
        const int t0 = desc[6 + c], n_c = desc[6 + c + 1] - t0, nz = (n_c * B + KR - 1) / KR;
        if (u < nz * tiles) { cls = c; cnt = n_c; tok0 = t0; z = u / tiles; u -= z * tiles; break; }
        u -= nz * tiles;
        ubase += nz;
    }
    if (cls < 0) return;
    const int dt = u / C, ch = u % C;
    const int row_beg = z * KR, row_end = min(row_beg + KR, cnt * B), nsteps = (row_end - row_beg + BW_ROWS - 1) / BW_ROWS;

    const int srow = tid >> 3, sub = tid & 7;                        // staging: one row, 2 segments + 32 d per thread
    constexpr int NX = XBF16 ? 4 : 8;
    u32x4 xr[NX], yr[4];
    auto load_stage = [&](int st) __attribute__((always_inline)) {
        const int ml = row_beg + st * BW_ROWS + srow;
        const bool valid = ml < row_end;
        const int mc = valid ? ml : row_end - 1;
        const int n = toks[tok0 + mc / B], b = mc % B;
        const long long xoff = ((long long)(b) * C + ch) * HW + origin[n] + (2 * sub) * sstep;
        if (XBF16) {
            const uint16_t *px = static_cast<const uint16_t *>(x) + xoff;
#pragma unroll
            for (int sgi = 0; sgi < 2; sgi++)
#pragma unroll
                for (int i = 0; i < 2; i++) xr[2 * sgi + i] = *reinterpret_cast<const u32x4 *>(px + sgi * sstep + 8 * i);
        } else {
            const float *px = static_cast<const float *>(x) + xoff;
#pragma unroll
            for (int sgi = 0; sgi < 2; sgi++)
#pragma unroll
                for (int i = 0; i < 4; i++) xr[4 * sgi + i] = *reinterpret_cast<const u32x4 *>(px + sgi * sstep + 4 * i);
        }
        const uint16_t *py = dy + (size_t(b) * N + n) * D + dt * 256 + 32 * sub;
#pragma unroll
        for (int i = 0; i < 4; i++) yr[i] = valid ? *reinterpret_cast<const u32x4 *>(py + 8 * i) : u32x4{0u, 0u, 0u, 0u};
    };
    auto write_stage = [&](int buf) __attribute__((always_inline)) {
        char *base = smem + buf * 4 * BW_IMG;
#pragma unroll
        for (int i = 0; i < 4; i++) {                                // dY: d columns 32 sub + 8 i of this row
            const int col = 32 * sub + 8 * i;
            *reinterpret_cast<u32x4 *>(base + (col >> 7) * BW_IMG + st_off(srow, col & 127)) = yr[i];
        }
#pragma unroll
        for (int sgi = 0; sgi < 2; sgi++) {                          // X: segment 2 sub + sgi = pixels 16 seg .. +15 of the channel
            const int seg = 2 * sub + sgi, col = (seg & 7) * 16;
            char *img = base + (2 + (seg >> 3)) * BW_IMG;
            u32x4 lo, hi;
            if (XBF16) {
                lo = xr[2 * sgi]; hi = xr[2 * sgi + 1];
            } else {
                const float *f = reinterpret_cast<const float *>(xr) + 16 * sgi;
                lo = u32x4{pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
                hi = u32x4{pack2bf(f[8], f[9]), pack2bf(f[10], f[11]), pack2bf(f[12], f[13]), pack2bf(f[14], f[15])};
            }
            *reinterpret_cast<u32x4 *>(img + st_off(srow, col)) = lo;
            *reinterpret_cast<u32x4 *>(img + st_off(srow, col + 8)) = hi;
        }
    };

    f32x4 acc[8][4];                                                 // wave tile: 128 d (wr) x 64 pixels (wc)
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_stage(0);
    write_stage(0);
    __syncthreads();
    for (int st = 0; st < nsteps; st++) {
        const int cur = st & 1;
        if (st + 1 < nsteps) load_stage(st + 1);
        const char *base = smem + cur * 4 * BW_IMG;
        const char *yimg = base + wr * BW_IMG, *ximg = base + (2 + (wc >> 1)) * BW_IMG;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            bf16x8 fb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) fb[j] = st_frag(ximg, (wc & 1) * 64 + 16 * j, kk, lane);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const bf16x8 fa = st_frag(yimg, 16 * i, kk, lane);
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[j], acc[i][j], 0, 0, 0);
            }
        }
        if (st + 1 < nsteps) write_stage(cur ^ 1);
        __syncthreads();
    }
    mfma_fence();
    // acc[i][j][r] = dW'[d = dt*256 + 128 wr + 16 i + 4 (lane >> 4) + r][pixel 64 wc + 16 j + (lane & 15)] of channel ch
    float *slab = slabs + (size_t(ubase + z) * D + dt * 256 + 128 * wr + 4 * (lane >> 4)) * K + ch * 256 + 64 * wc + (lane & 15);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) slab[size_t(16 * i + r) * K + 16 * j] = acc[i][j][r];
}

// dW[d][k*C + c] = sum over classes and their row ranges of slab[u][d][c*256 + inverse_perm_cls(k)]
__global__ __launch_bounds__(256) void pe2_bwd_reduce(const float *__restrict__ slabs, const int32_t *__restrict__ desc, float *__restrict__ dw,
                                                      int B, int C, int D, int KR) {
    const int K = C * 256, ncls = desc[1], N = desc[4];
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= int64_t(D) * K) return;
    const int d = int(i / K), f = int(i % K), k = f / C, c = f % C;
    const int32_t *inv = desc + DESC_HDR + 2 * N + ncls * 256;
    float s = 0.f;
    int ubase = 0;
    for (int cl = 0; cl < ncls; cl++) {
        const int n_c = desc[6 + cl + 1] - desc[6 + cl], nz = (n_c * B + KR - 1) / KR;
        const int j = inv[cl * 256 + k];
        for (int z = 0; z < nz; z++) s += slabs[(size_t(ubase + z) * D + d) * K + c * 256 + j];
        ubase += nz;
    }
    dw[i] = s;
}

constexpr int PE2_BWD_LDS = 2 * 4 * BW_IMG;

// Rows per unit: the smallest multiple of 64 for which the row ranges of all classes number at most G = 8 x (32 / tiles):
// every XCD then holds whole groups (the `tiles` output tiles of one row range) on its 32 CUs in ONE round.
int pe2_bwd_row_range(int B, int ncls, const int32_t *cnt, int tiles, int *nz_out) {
    const int G = 8 * std::max(1, 32 / tiles);
    int64_t M = 0;
    for (int c = 0; c < ncls; c++) M += int64_t(cnt[c]) * B;
    int kr = int((M / G + BW_ROWS - 1) / BW_ROWS * BW_ROWS);
    if (kr < BW_ROWS) kr = BW_ROWS;
    for (;; kr += BW_ROWS) {
        int nz = 0;
        for (int c = 0; c < ncls; c++) nz += int((int64_t(cnt[c]) * B + kr - 1) / kr);
        if (nz <= G) { *nz_out = nz; return kr; }
    }
}

constexpr int PE2_LDS = 2 * A_BYTES + 2 * B_BYTES + TM * int(sizeof(RowInfo));

}  // namespace

// Eligibility + launch of the tiled forward.  -1: not eligible (the caller uses the generic kernel).
int pe2_fwd(const sfcvit_patch_embed_args &a, hipStream_t s) {
    if (!a.desc || a.P != 256 || a.D % TN || a.desc_ncls <= 0 || a.desc_ncls > MAXCLS) return -1;
    const int K = a.C * 256;
    const int64_t need = int64_t(a.desc_ncls) * a.D * K * 2;
    if (!a.workspace || a.workspace_bytes < need) return -1;
    if ((reinterpret_cast<uintptr_t>(a.x) & 15) || (a.HW & 7)) return -1;
    uint16_t *wp = static_cast<uint16_t *>(a.workspace);
    hipLaunchKernelGGL(pe2_permute_w_kernel, dim3(1024), dim3(256), 0, s, static_cast<const uint16_t *>(a.w), a.desc + DESC_HDR + 2 * a.N,
                       wp, a.D, a.C, a.desc_ncls);
    if (int rc = check_launch("patch_embed_fwd (tiled) permute")) return rc;
    for (const void *k : {reinterpret_cast<const void *>(&pe2_fwd_kernel<false>), reinterpret_cast<const void *>(&pe2_fwd_kernel<true>)})
        if (int rc = raise_lds_limit(k, PE2_LDS, "patch_embed_fwd (tiled) attribute")) return rc;
    int n_row_tiles = 0;
    for (int c = 0; c < a.desc_ncls; c++) n_row_tiles += int((int64_t(a.desc_cnt[c]) * a.B + TM - 1) / TM);
    const int NCT = a.D / TN;
    const int groups = (n_row_tiles + 7) / 8;                       // row tiles are dealt to the 8 XCD slots in groups of 8
    dim3 grid(unsigned(groups) * 8u * unsigned(NCT)), block(PT);
    if (a.x_is_bf16)
        hipLaunchKernelGGL(pe2_fwd_kernel<true>, grid, block, PE2_LDS, s, a.x, a.desc, wp, static_cast<const uint16_t *>(a.bias),
                           static_cast<uint16_t *>(a.y), a.B, a.C, a.HW, a.D, n_row_tiles);
    else
        hipLaunchKernelGGL(pe2_fwd_kernel<false>, grid, block, PE2_LDS, s, a.x, a.desc, wp, static_cast<const uint16_t *>(a.bias),
                           static_cast<uint16_t *>(a.y), a.B, a.C, a.HW, a.D, n_row_tiles);
    return check_launch("patch_embed_fwd (tiled)");
}

// Tiled backward: -1 = not eligible.  Workspace: fp32 slabs, one [D][C*256] per (class, row range).
int64_t pe2_bwd_workspace(int B, int C, int N, int D) {
    const int tiles = (D / 256) * C;
    return int64_t(8 * std::max(1, 32 / tiles)) * D * C * 256 * int64_t(sizeof(float));      // at most G slabs
}

int pe2_bwd(const sfcvit_patch_embed_args &a, hipStream_t s) {
    if (!a.desc || a.P != 256 || a.D % 256 || a.desc_ncls <= 0 || a.desc_ncls > MAXCLS) return -1;
    if ((reinterpret_cast<uintptr_t>(a.x) & 15) || (a.HW & 7)) return -1;
    int nz = 0;
    const int KR = pe2_bwd_row_range(a.B, a.desc_ncls, a.desc_cnt, (a.D / 256) * a.C, &nz);
    const int64_t need = int64_t(nz) * a.D * a.C * 256 * int64_t(sizeof(float));
    if (!a.workspace || a.workspace_bytes < need) return -1;
    for (const void *k : {reinterpret_cast<const void *>(&pe2_bwd_kernel<false>), reinterpret_cast<const void *>(&pe2_bwd_kernel<true>)})
        if (int rc = raise_lds_limit(k, PE2_BWD_LDS, "patch_embed_bwd (tiled) attribute")) return rc;
    float *slabs = static_cast<float *>(a.workspace);
    dim3 grid(unsigned((nz + 7) / 8) * 8u * unsigned((a.D / 256) * a.C)), block(PT);      // row ranges dealt to the 8 XCD slots
    if (a.x_is_bf16)
        hipLaunchKernelGGL(pe2_bwd_kernel<true>, grid, block, PE2_BWD_LDS, s, a.x, a.desc, static_cast<const uint16_t *>(a.y), slabs, a.B, a.C, a.HW, a.D, KR);
    else
        hipLaunchKernelGGL(pe2_bwd_kernel<false>, grid, block, PE2_BWD_LDS, s, a.x, a.desc, static_cast<const uint16_t *>(a.y), slabs, a.B, a.C, a.HW, a.D, KR);
    if (int rc = check_launch("patch_embed_bwd (tiled)")) return rc;
    const int64_t nw = int64_t(a.D) * a.C * 256;
    hipLaunchKernelGGL(pe2_bwd_reduce, dim3(unsigned((nw + 255) / 256)), dim3(256), 0, s, slabs, a.desc, static_cast<float *>(a.dw), a.B, a.C, a.D, KR);
    return check_launch("patch_embed_bwd (tiled) reduce");
}

}  // namespace sfcvit
