// Fused hierarchical (multi-scale) curve tokenizer, forward, for gfx950 (sfcvit_hier_tokenizer_fwd in include/sfcvit.h).
//
// The reference (src/tokenizers/multiscale/multi_hilbert.py:31-40 and its five siblings) runs L SFCEmbedding1D
// levels -- each a curve gather + patchify + Linear(K_l -> D) -- resamples them to a common token count, concatenates
// on the feature axis and applies fusion = Linear(L*D -> L*D): 2L + 2 passes, with the gathered tokens, the level
// outputs and their concatenation all materialised.  In every configuration the reference ships (main.py:269-274:
// [16, 4, 1] at 32 x 32) the levels have the SAME token count, so the resampling is the identity and
//     y[m, :] = Wf . concat_l( W_l tokens_l[m, :] + b_l ) + bf
// is two chained GEMMs per token row.  One workgroup owns 64 token rows:
//   phase 1, per level: the level's tokens are gathered from the image through its pixel table straight into an LDS
//            operand tile (they never exist in HBM), multiplied by W_l (B fragments read from L2, the whole W_l is a
//            few KB), bias added, rounded to bf16 -- the rounding the unfused path applies when it stores the level
//            output -- and written into the workgroup's [64][L*D] LDS image of the concatenation (and once to HBM:
//            backward needs it for the fusion weight gradient).
//   phase 2: that LDS image is the A operand of the fusion GEMM; each of the 4 waves owns 64 of every 256 output
//            columns and streams its Wf fragments from L2 through a 4-deep register ring (one workgroup per CU, one
//            wave per SIMD: the ring is what hides the L2 latency).
// LDS: 64 x (2 L D + 16) bytes of concatenation + 64 x (2 Kp + 16) of tokens (Kp = sum of the K_l, each rounded up to
// 32); the 16-byte row pad makes both images conflict-free for the 16-lane ds_read_b128 groups (row stride = 4 dwords
// mod 64).
// Shapes outside the envelope (different token counts per level, K_l % 8, D % 64, L*D % 256, LDS) return
// SFCVIT_EINVAL (sfcvit_hier_tokenizer_supported says so beforehand); the caller then composes the level kernels, torch's interpolate and the fusion GEMM.
#include "common_host.h"
#include "device_common.h"
#include <hip/hip_runtime.h>

namespace sfcvit {
namespace {

constexpr int HT_ROWS = 64, HT_THREADS = 256, HT_MAXL = 4, HT_RING = 4;

struct HierGeo {
    const void *x;
    const int32_t *pix[HT_MAXL];
    const uint16_t *w[HT_MAXL];
    const uint16_t *b[HT_MAXL];
    int P[HT_MAXL];
    const uint16_t *wf, *bf;
    uint16_t *h, *y;
    int B, C, HW, N, L, D, E, M;
    int h_stride, a_stride;     // LDS row strides in bytes
};

// Result fragment (i, j) of a wave: acc = mfma(W fragment, token fragment): lane holds token row 16 i + (lane & 15),
// columns 16 j + 4 (lane >> 4) .. + 3.
__device__ __forceinline__ void add_bias_pack(const f32x4 &acc, const uint16_t *__restrict__ bias, int n, u32x2 &out) {
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
        const u32x2 b2 = *reinterpret_cast<const u32x2 *>(bias + n);
        bv[0] = bf2f(uint16_t(b2[0])); bv[1] = bf2f(uint16_t(b2[0] >> 16));
        bv[2] = bf2f(uint16_t(b2[1])); bv[3] = bf2f(uint16_t(b2[1] >> 16));
    }
    out = u32x2{pack2bf(acc[0] + bv[0], acc[1] + bv[1]), pack2bf(acc[2] + bv[2], acc[3] + bv[3])};
}

// FUSE = true: everything above.  FUSE = false: phase 1 only -- the level outputs go straight to their columns of h
// in HBM (the concatenation without a torch.cat pass) and the fusion Linear is left to the caller's GEMM; LDS is then
// the token tile alone (<= 26 KiB at the reference's shape: several workgroups per CU hide the gather latency).
template <bool XBF16, bool FUSE>
__global__ __launch_bounds__(HT_THREADS) void hier_fwd_kernel(const HierGeo g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *himg = smem;                                   // [64][h_stride]: concat_l(level outputs), bf16 (FUSE only)
    char *aimg = smem + (FUSE ? HT_ROWS * g.h_stride : 0);   // [64][a_stride]: the tokens of ALL levels side by side, bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * HT_ROWS;
    const int lr = lane & 15, lg = lane >> 4;

    // ---- phase 1a: gather the tokens of every level (one exposed image-load latency per tile, not one per level) --
    {
        int kp_off = 0;
        for (int l = 0; l < g.L; l++) {
            const int P = g.P[l], K = P * g.C, Kp = (K + 31) & ~31, nvec = Kp >> 3;
            const int32_t *__restrict__ pix = g.pix[l];
            for (int v = tid; v < HT_ROWS * nvec; v += HT_THREADS) {
                const int r = v / nvec, f0 = (v - r * nvec) << 3, m = m0 + r;
                float val[8];
#pragma unroll
                for (int e = 0; e < 8; e++) val[e] = 0.f;
                if (m < g.M && f0 < K) {                     // K % 8 == 0: a vector is all features or all padding
                    const int b = m / g.N, t = m - b * g.N;
                    const size_t img = size_t(b) * g.C * g.HW;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int f = f0 + e, kk = f / g.C, c = f - kk * g.C;      // reference feature order kk * C + c
                        const size_t off = img + size_t(c) * g.HW + pix[size_t(t) * P + kk];
                        val[e] = XBF16 ? bf2f(static_cast<const uint16_t *>(g.x)[off]) : static_cast<const float *>(g.x)[off];
                    }
                }
                *reinterpret_cast<u32x4 *>(aimg + r * g.a_stride + (kp_off + f0) * 2) =
                    u32x4{pack2bf(val[0], val[1]), pack2bf(val[2], val[3]), pack2bf(val[4], val[5]), pack2bf(val[6], val[7])};
            }
            kp_off += Kp;
        }
    }
    __syncthreads();

    // ---- phase 1b: level projections into the LDS concatenation ---------------------------------------------------
    {
        int kp_off = 0;
        for (int l = 0; l < g.L; l++) {
            const int K = g.P[l] * g.C, Kp = (K + 31) & ~31;
            const uint16_t *__restrict__ w = g.w[l];
            for (int cg = wave; cg * 64 < g.D; cg += 4) {
                f32x4 acc[4][4];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int ks = 0; ks < Kp; ks += 32) {
                    const int k = ks + lg * 8;
                    bf16x8 fa[4], fb[4];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        fa[i] = *reinterpret_cast<const bf16x8 *>(aimg + (i * 16 + lr) * g.a_stride + (kp_off + k) * 2);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int n = cg * 64 + j * 16 + lr;
                        fb[j] = k < K ? *reinterpret_cast<const bf16x8 *>(w + size_t(n) * K + k) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
                mfma_fence();
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int n = cg * 64 + j * 16 + 4 * lg;             // column within the level
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        u32x2 o;
                        add_bias_pack(acc[i][j], g.b[l], n, o);
                        if (FUSE) *reinterpret_cast<u32x2 *>(himg + (i * 16 + lr) * g.h_stride + (l * g.D + n) * 2) = o;
                        else if (m0 + i * 16 + lr < g.M) *reinterpret_cast<u32x2 *>(g.h + size_t(m0 + i * 16 + lr) * g.E + l * g.D + n) = o;
                    }
                }
            }
            kp_off += Kp;
        }
    }
    if (!FUSE) return;
    __syncthreads();
    // the concatenation goes to HBM once (the fusion weight gradient needs it), as whole 16-byte vectors of whole rows
    {
        const int vpr = g.E >> 3;
        for (int v = tid; v < HT_ROWS * vpr; v += HT_THREADS) {
            const int r = v / vpr, c = v - r * vpr;
            if (m0 + r < g.M)
                *reinterpret_cast<u32x4 *>(g.h + size_t(m0 + r) * g.E + c * 8) = *reinterpret_cast<const u32x4 *>(himg + r * g.h_stride + c * 16);
        }
    }

    // ---- phase 2: fusion GEMM from the LDS concatenation ---------------------------------------------------------
    // A k-block is 64 deep = two MFMA k-steps; lane group lg takes k = 16 lg .. 16 lg + 15 of it (the first 8 in the
    // first MFMA, the other 8 in the second; the same assignment on the A side), so that a lane's Wf read is 32
    // contiguous bytes and the 4 lane groups of a row cover one whole 128-byte line: every line of Wf is fetched once.
    const int nkb = g.E >> 6;                                        // E % 256 == 0 -> a multiple of 4
    for (int cb = 0; cb < g.E; cb += 256) {
        const int n0 = cb + wave * 64;
        const uint16_t *__restrict__ wrow[4];
#pragma unroll
        for (int j = 0; j < 4; j++) wrow[j] = g.wf + size_t(n0 + j * 16 + lr) * g.E + lg * 16;
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 ring[HT_RING][4][2];
#pragma unroll
        for (int s = 0; s < HT_RING - 1; s++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                ring[s][j][0] = *reinterpret_cast<const bf16x8 *>(wrow[j] + s * 64);
                ring[s][j][1] = *reinterpret_cast<const bf16x8 *>(wrow[j] + s * 64 + 8);
            }
            __builtin_amdgcn_sched_barrier(0);   // oldest stage first: the loop's counted waits assume issue order = use order
        }
        for (int kb0 = 0; kb0 < nkb; kb0 += HT_RING) {
#pragma unroll
            for (int s = 0; s < HT_RING; s++) {
                // no branch around the prefetch (the last blocks re-read the final one): with a conditional load the
                // compiler loses count of the loads in flight and waits for ALL of them at the loop head -- 132 us
                const int kb = kb0 + s, nxt = min(kb + HT_RING - 1, nkb - 1);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    ring[(s + HT_RING - 1) % HT_RING][j][0] = *reinterpret_cast<const bf16x8 *>(wrow[j] + nxt * 64);
                    ring[(s + HT_RING - 1) % HT_RING][j][1] = *reinterpret_cast<const bf16x8 *>(wrow[j] + nxt * 64 + 8);
                }
                __builtin_amdgcn_sched_barrier(0);       // keep the prefetch here: the scheduler sinks it to its use otherwise
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    bf16x8 fa[4];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        fa[i] = *reinterpret_cast<const bf16x8 *>(himg + (i * 16 + lr) * g.h_stride + (kb * 64 + lg * 16 + hf * 8) * 2);
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[s][j][hf], fa[i], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mfma_fence();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + j * 16 + 4 * lg;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = m0 + i * 16 + lr;
                u32x2 o;
                add_bias_pack(acc[i][j], g.bf, n, o);
                if (m < g.M) *reinterpret_cast<u32x2 *>(g.y + size_t(m) * g.E + n) = o;
            }
        }
    }
}

int hier_lds_bytes(int E, int kp_sum) { return HT_ROWS * (2 * E + 16) + HT_ROWS * (2 * kp_sum + 16); }

}  // namespace
}  // namespace sfcvit

using namespace sfcvit;

extern "C" int sfcvit_hier_tokenizer_supported(int L, int D, int C, const int32_t *P) {
    if (L < 1 || L > HT_MAXL || D <= 0 || D % 64 || (L * D) % 256 || C <= 0 || !P) return 0;
    int kp_sum = 0;
    for (int l = 0; l < L; l++) {
        const int K = P[l] * C;
        if (P[l] <= 0 || K % 8) return 0;
        kp_sum += (K + 31) & ~31;
    }
    return hier_lds_bytes(L * D, kp_sum) <= 160 * 1024;
}

extern "C" int sfcvit_hier_tokenizer_fwd(const sfcvit_hier_args *a, void *stream) {
    if (!a || !a->x || !a->h || (a->wf && !a->y)) return fail(SFCVIT_EINVAL, "hier_tokenizer_fwd: null pointer");
    if (a->B <= 0 || a->C <= 0 || a->HW <= 0 || a->N <= 0)
        return fail(SFCVIT_EINVAL, "hier_tokenizer_fwd: B=%d C=%d HW=%d N=%d", a->B, a->C, a->HW, a->N);
    if (!sfcvit_hier_tokenizer_supported(a->L, a->D, a->C, a->P))
        return fail(SFCVIT_EINVAL, "hier_tokenizer_fwd: L=%d D=%d C=%d outside the fused kernel's envelope "
                    "(1..4 levels, D %% 64 == 0, L*D %% 256 == 0, P*C %% 8 == 0, 64 rows of L*D + sum K in 160 KiB LDS)",
                    a->L, a->D, a->C);
    if (int64_t(a->B) * a->N >= (int64_t(1) << 31) - HT_ROWS) return fail(SFCVIT_EINVAL, "hier_tokenizer_fwd: too many token rows");
    HierGeo g{};
    g.x = a->x;
    int kp_sum = 0;
    for (int l = 0; l < a->L; l++) {
        if (!a->pix[l] || !a->w[l]) return fail(SFCVIT_EINVAL, "hier_tokenizer_fwd: level %d: null pointer", l);
        if (int64_t(a->P[l]) * a->N != a->HW)
            return fail(SFCVIT_EINVAL, "hier_tokenizer_fwd: level %d: N * P = %d * %d != H*W = %d (levels must share the token count)",
                        l, a->N, a->P[l], a->HW);
        if (!aligned16(a->w[l]) || (a->b[l] && (reinterpret_cast<uintptr_t>(a->b[l]) & 7)))
            return fail(SFCVIT_EINVAL, "hier_tokenizer_fwd: level %d: weight / bias alignment", l);
        g.pix[l] = a->pix[l];
        g.w[l] = static_cast<const uint16_t *>(a->w[l]);
        g.b[l] = static_cast<const uint16_t *>(a->b[l]);
        g.P[l] = a->P[l];
        kp_sum += (a->P[l] * a->C + 31) & ~31;
    }
    const bool fuse = a->wf != nullptr;
    if (!aligned16(a->wf) || !aligned16(a->h) || !aligned16(a->y) || (a->bf && (reinterpret_cast<uintptr_t>(a->bf) & 7)))
        return fail(SFCVIT_EINVAL, "hier_tokenizer_fwd: fusion weight / output alignment");
    g.wf = static_cast<const uint16_t *>(a->wf);
    g.bf = static_cast<const uint16_t *>(a->bf);
    g.h = static_cast<uint16_t *>(a->h);
    g.y = static_cast<uint16_t *>(a->y);
    g.B = a->B; g.C = a->C; g.HW = a->HW; g.N = a->N; g.L = a->L; g.D = a->D;
    g.E = a->L * a->D;
    g.M = a->B * a->N;
    g.h_stride = 2 * g.E + 16;
    g.a_stride = 2 * kp_sum + 16;
    const int lds = fuse ? hier_lds_bytes(g.E, kp_sum) : HT_ROWS * g.a_stride;
    const int grid = (g.M + HT_ROWS - 1) / HT_ROWS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const void *k = a->x_is_bf16 ? (fuse ? reinterpret_cast<const void *>(&hier_fwd_kernel<true, true>) : reinterpret_cast<const void *>(&hier_fwd_kernel<true, false>))
                                 : (fuse ? reinterpret_cast<const void *>(&hier_fwd_kernel<false, true>) : reinterpret_cast<const void *>(&hier_fwd_kernel<false, false>));
    if (int rc = raise_lds_limit(k, 160 * 1024, "hier_tokenizer_fwd attribute")) return rc;
    if (a->x_is_bf16 && fuse) hipLaunchKernelGGL((hier_fwd_kernel<true, true>), dim3(grid), dim3(HT_THREADS), lds, s, g);
    else if (a->x_is_bf16) hipLaunchKernelGGL((hier_fwd_kernel<true, false>), dim3(grid), dim3(HT_THREADS), lds, s, g);
    else if (fuse) hipLaunchKernelGGL((hier_fwd_kernel<false, true>), dim3(grid), dim3(HT_THREADS), lds, s, g);
    else hipLaunchKernelGGL((hier_fwd_kernel<false, false>), dim3(grid), dim3(HT_THREADS), lds, s, g);
    return check_launch("hier_tokenizer_fwd");
}


// ---------------------------------------------------------------------------------------------------------------------
// Levels with DIFFERENT token counts (multi_hilbert.py:33-38): every coarser level is resampled to the first level's
// length by F.interpolate(mode="linear", align_corners=False) and the levels are concatenated on the feature axis.  Both
// in one pass: out[b, i, l D + d] = w0 y_l[b, i0, d] + w1 y_l[b, i1, d] with torch's source index
// src = max(0, (i + 0.5) N_l / N0 - 0.5), i0 = floor(src), i1 = min(i0 + 1, N_l - 1), w1 = src - i0 (fp32, as torch's
// upsample_linear1d computes it for bf16 input), a level of N0 tokens copied.  One thread per 8 columns of an output row.
// Backward is the transposed gather, in a fixed order: input token j of level l collects the output rows whose taps hit
// it (a window of ~2 N0 / N_l rows found by inverting the index map, each row's taps recomputed exactly as forward).
// ---------------------------------------------------------------------------------------------------------------------
namespace sfcvit {
namespace {
constexpr int RS_MAXL = 8;
struct ResampleArgs {
    const uint16_t *lev[RS_MAXL];       // forward: y_l [B, N_l, D];  backward: unused
    uint16_t *dlev[RS_MAXL];            // backward: d y_l
    int n[RS_MAXL];
    int L, B, N0, D;
};

__device__ __forceinline__ void rs_taps(int i, int nl, int n0, int &i0, int &i1, float &w1) {
    const float scale = float(nl) / float(n0);
    float src = scale * (float(i) + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = int(src);
    i0 = i0 > nl - 1 ? nl - 1 : i0;
    i1 = i0 + (i0 < nl - 1 ? 1 : 0);
    w1 = src - float(i0);
}

__global__ __launch_bounds__(256) void hier_resample_concat_kernel(const ResampleArgs a, uint16_t *__restrict__ out) {
    const int vpr = a.D / 8, per_row = a.L * vpr;
    const int64_t total = int64_t(a.B) * a.N0 * per_row;
    for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < total; t += int64_t(gridDim.x) * 256) {
        const int64_t row = t / per_row;
        const int r = int(t - row * per_row), l = r / vpr, v = r - l * vpr;
        const int b = int(row / a.N0), i = int(row - int64_t(b) * a.N0);
        const int nl = a.n[l];
        const uint16_t *y = a.lev[l] + size_t(b) * nl * a.D + v * 8;
        u32x4 o;
        if (nl == a.N0) {
            o = *reinterpret_cast<const u32x4 *>(y + size_t(i) * a.D);
        } else {
            int i0, i1;
            float w1;
            rs_taps(i, nl, a.N0, i0, i1, w1);
            const float w0 = 1.f - w1;
            const u32x4 p = *reinterpret_cast<const u32x4 *>(y + size_t(i0) * a.D);
            const u32x4 q = *reinterpret_cast<const u32x4 *>(y + size_t(i1) * a.D);
#pragma unroll
            for (int k = 0; k < 4; k++)
                o[k] = pack2bf(w0 * bf2f(uint16_t(p[k])) + w1 * bf2f(uint16_t(q[k])),
                               w0 * bf2f(uint16_t(p[k] >> 16)) + w1 * bf2f(uint16_t(q[k] >> 16)));
        }
        *reinterpret_cast<u32x4 *>(out + size_t(row) * a.L * a.D + size_t(l) * a.D + v * 8) = o;
    }
}

__global__ __launch_bounds__(256) void hier_resample_concat_bwd_kernel(const ResampleArgs a, const uint16_t *__restrict__ dout, int l) {
    const int vpr = a.D / 8, nl = a.n[l];
    const int64_t total = int64_t(a.B) * nl * vpr;
    const float inv = float(a.N0) / float(nl);
    for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < total; t += int64_t(gridDim.x) * 256) {
        const int64_t row = t / vpr;
        const int v = int(t - row * vpr), b = int(row / nl), j = int(row - int64_t(b) * nl);
        const uint16_t *g = dout + (size_t(b) * a.N0) * a.L * a.D + size_t(l) * a.D + v * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto add = [&](int i, float w) {
            const u32x4 d = *reinterpret_cast<const u32x4 *>(g + size_t(i) * a.L * a.D);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                acc[2 * k] += w * bf2f(uint16_t(d[k]));
                acc[2 * k + 1] += w * bf2f(uint16_t(d[k] >> 16));
            }
        };
        if (nl == a.N0) {
            add(j, 1.f);
        } else {
            // rows whose source index lies in (j - 1, j + 1): src = (i + 0.5) / inv - 0.5  (two rows of slack either side)
            int lo = int(floorf((float(j) - 0.5f) * inv - 0.5f)) - 2, hi = int(ceilf((float(j) + 1.5f) * inv - 0.5f)) + 2;
            lo = lo < 0 ? 0 : lo;
            hi = hi > a.N0 - 1 ? a.N0 - 1 : hi;
            for (int i = lo; i <= hi; i++) {
                int i0, i1;
                float w1;
                rs_taps(i, nl, a.N0, i0, i1, w1);
                if (i0 == j) add(i, 1.f - w1);
                if (i1 == j && w1 != 0.f) add(i, w1);
            }
        }
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = pack2bf(acc[2 * k], acc[2 * k + 1]);
        *reinterpret_cast<u32x4 *>(a.dlev[l] + size_t(row) * a.D + v * 8) = o;
    }
}

int rs_check(const void *const *lev, const int32_t *n_tokens, int L, int B, int N0, int D, const void *cat, const char *what) {
    if (!lev || !n_tokens || !cat) return fail(SFCVIT_EINVAL, "%s: null pointer", what);
    if (L < 1 || L > RS_MAXL || B <= 0 || N0 <= 0 || D <= 0 || D % 8) return fail(SFCVIT_EINVAL, "%s: L=%d B=%d N0=%d D=%d (1 <= L <= %d, D %% 8 == 0)", what, L, B, N0, D, RS_MAXL);
    if (n_tokens[0] != N0) return fail(SFCVIT_EINVAL, "%s: the first level has %d tokens, not N0 = %d", what, n_tokens[0], N0);
    if (!aligned16(cat)) return fail(SFCVIT_EINVAL, "%s: alignment", what);
    for (int l = 0; l < L; l++)
        if (!lev[l] || n_tokens[l] <= 0 || !aligned16(lev[l])) return fail(SFCVIT_EINVAL, "%s: level %d (pointer / token count / alignment)", what, l);
    return SFCVIT_OK;
}
}  // namespace
}  // namespace sfcvit

extern "C" int sfcvit_hier_resample_concat(const void *const *levels, const int32_t *n_tokens, int L, int B, int N0, int D, void *out,
                                           void *stream) {
    if (int rc = rs_check(levels, n_tokens, L, B, N0, D, out, "hier_resample_concat")) return rc;
    ResampleArgs a{};
    for (int l = 0; l < L; l++) { a.lev[l] = static_cast<const uint16_t *>(levels[l]); a.n[l] = n_tokens[l]; }
    a.L = L; a.B = B; a.N0 = N0; a.D = D;
    const int64_t total = int64_t(B) * N0 * L * (D / 8);
    const unsigned blocks = unsigned(std::min<int64_t>((total + 255) / 256, 1 << 16));
    hipLaunchKernelGGL(hier_resample_concat_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, static_cast<uint16_t *>(out));
    return check_launch("hier_resample_concat");
}

extern "C" int sfcvit_hier_resample_concat_bwd(const void *dout, const int32_t *n_tokens, int L, int B, int N0, int D, void *const *dlevels,
                                               void *stream) {
    if (int rc = rs_check(const_cast<const void *const *>(dlevels), n_tokens, L, B, N0, D, dout, "hier_resample_concat_bwd")) return rc;
    ResampleArgs a{};
    for (int l = 0; l < L; l++) { a.dlev[l] = static_cast<uint16_t *>(dlevels[l]); a.n[l] = n_tokens[l]; }
    a.L = L; a.B = B; a.N0 = N0; a.D = D;
    for (int l = 0; l < L; l++) {
        const int64_t total = int64_t(B) * n_tokens[l] * (D / 8);
        const unsigned blocks = unsigned(std::min<int64_t>((total + 255) / 256, 1 << 16));
        hipLaunchKernelGGL(hier_resample_concat_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                           static_cast<const uint16_t *>(dout), l);
        if (int rc = check_launch("hier_resample_concat_bwd")) return rc;
    }
    return SFCVIT_OK;
}
