// Whole-sequence attention kernels for gfx950 (head dim 64, N <= 256: every ViT at 16x16-pixel
// tokens up to 256x256 images).  One workgroup per (batch, head): K and V (forward, dQ) or Q and
// dO (dK/dV) are staged into LDS ONCE for the whole sequence (2 x 26 KiB at N = 196: the forward and dQ kernels
// pad the sequence to 16, not 32, keys so that THREE workgroups fit a CU's 160 KiB), then every
// wave walks its 16-row fragments with no barrier in the loop and no online-softmax rescale
// (a whole score row, <= 16 fragments, lives in registers).  The tiled kernels of attention.hip
// re-staged K/V for every 64 queries and synchronised twice per 64 keys; they remain the path
// for longer sequences.  MFMA orientations are those documented in attention.hip.
#include "attention_common.h"
#include "common_host.h"

namespace sfcvit {
namespace {

using namespace attn;

// Tried and not kept: the same kernels for N <= 576 (ViT-L at 384 x 384: K + V = 144 KiB of LDS, a 36-fragment score
// row in registers, one workgroup of four waves per CU).  Correct, but 14.0 vs 8.0 ms per ViT-L step forward and 26.7 vs
// 22.8 ms backward against the tiled kernels of attention.hip: at one wave per SIMD nothing hides the exp / reduction
// latencies between the MFMA bursts.
constexpr int MAXF = 16;   // 16-row fragments per sequence (N <= 256)
constexpr int MAXC = 8;    // 32-row chunks

// Stage `npad` rows x 64 cols of a [N, ld] matrix into an LDS image by LDS-DMA
// (global_load_lds_dwordx4: asynchronous, no staging registers).  The DMA writes lane-linear
// (slot p = tid + 256*i at LDS byte 16*p), so the image's bank swizzle is applied to the SOURCE
// column chunk.  Rows >= N are filled with a copy of row N-1 (finite values); every consumer
// masks them: keys >= N get probability / dS = 0, queries >= N get lse = +inf.
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
template <bool VT>
__device__ __forceinline__ void dma_seq(char *img, const uint16_t *__restrict__ src, int ld, int N, int npad, int tid) {
    for (int p = tid; p < npad * 8; p += THREADS) {     // npad % 32 == 16: the last trip is half a workgroup
        const int row = p >> 3, cs = p & 7;
        const int c = cs ^ kc_swz(row);        // "kc" and "vt" images share one swizzle now (device_common.h)
        const uint16_t *g = src + size_t(min(row, N - 1)) * ld + c * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(img + p * 16), 16, 0, 0);
    }
}

// 16 rows x 16 cols transposed fragment for the 16-deep MFMA (the odd last 16 keys of a sequence padded to 16)
__device__ __forceinline__ bf16x4 tr_frag16_at(const char *img, int r_lo, int lane_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFCVIT_LDS bf16x4 *)(img + r_lo * 128 + lane_off));
}
__device__ __forceinline__ bf16x4 pack_frag4(const f32x4 &a) {
    const u32x2 w = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
    return __builtin_bit_cast(bf16x4, w);
}

constexpr int WAVES = THREADS / 64;
constexpr int MAXOWN = MAXF / WAVES;   // fragments a wave owns at most (4)

// NFC = the number of 16-row fragments as a compile-time constant (0 = take it from N at run time).  With it the
// per-fragment `kf < nf` tests of the fully unrolled score row and the chunk loops fold away -- for the generic version
// hipcc emits ~250 branches around the forward kernel's MFMAs, each one a scheduling barrier.  ViT-B/16 @ 224
// (N = 196) runs the 13-fragment instances.
// DROP = dropout on the probabilities, at compile time.  Measured and not kept (round 3, ViT-B / 256, same process):
// seven waves per workgroup, two fragments per wave, two workgroups per CU -- 13 fragments then fall 2/2/2/2/2/2/1
// instead of 4/3/3/3 -- 109.7 vs 90.9 us; a start-up stagger of the three workgroups of a CU: 91-102 vs 90.5 us; handing the
// keep flags to the backward pass as a bit matrix (the compare's lane masks collected by v_writelane, one 8-byte store per
// lane and fragment; the one-pass backward then reads a word and extracts bits instead of hashing): forward 87.0 -> 98.5 us,
// backward 252.0 -> 242.0 us -- a wash, removed again.
template <int NFC, bool DROP>
__global__ __launch_bounds__(THREADS, 3) void attn_seq_fwd_kernel(const sfcvit_attn_args a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *kimg = smem, *vimg = smem + npad * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x, N = a.N, D = a.H * HD, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    uint16_t *out = static_cast<uint16_t *>(a.out) + size_t(b) * N * D + h * HD;
    const int nf = NFC ? NFC : (N + 15) >> 4, nc = NFC ? (NFC >> 1) : npad >> 5;   // npad = 16 nf; nc full 32-key chunks (+ a 16-key tail if nf is odd)
    // This wave's query fragments, fetched before the K/V staging so that their HBM latency is
    // hidden behind it (fragment index = wave + 4*o).
    bf16x8 qfr[MAXOWN][2];
#pragma unroll
    for (int o = 0; o < MAXOWN; o++) {
        qfr[o][0] = global_frag(qp, ld, 16 * (wave + WAVES * o), N, 0, lane);
        qfr[o][1] = global_frag(qp, ld, 16 * (wave + WAVES * o), N, 1, lane);
    }
    dma_seq<false>(kimg, kp, ld, N, npad, tid);
    dma_seq<true>(vimg, vp, ld, N, npad, tid);
    __syncthreads();                                     // LDS-DMA pending: hipcc drains vmcnt(0) here
    const float c2 = a.scale * 1.4426950408889634f;     // exp(x * scale) = exp2(x * c2)
    const LaneOff lo = lane_offsets(lane);
    constexpr bool drop = DROP;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);

#pragma unroll
    for (int o = 0; o < MAXOWN; o++) {
        const int qf = wave + WAVES * o;
        if (qf >= nf) break;                             // wave-uniform
        const int q = 16 * qf + (lane & 15);
        f32x4 s[MAXF];
#pragma unroll
        for (int kf = 0; kf < MAXF; kf++) {
            s[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kf < nf) {
#pragma unroll
                for (int kk = 0; kk < 2; kk++)
                    s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(kimg, 16 * kf, lo.k[kk]), qfr[o][kk], s[kf], 0, 0, 0);
            }
        }
        mfma_fence();                                    // branches between the MFMAs and the reads below
        float mx = -INFINITY;
#pragma unroll
        for (int kf = 0; kf < MAXF; kf++)
            if (kf < nf) {
                // only the boundary fragment needs the key mask; with the fragment count known at compile time that is the
                // last one (16 (NFC - 1) < N <= 16 NFC), and the other twelve carry no run-time test
                if (NFC ? kf == NFC - 1 : 16 * kf + 16 > N) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (16 * kf + 4 * (lane >> 4) + r >= N) s[kf][r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 4; r++) mx = fmaxf(mx, s[kf][r]);
            }
        mx = group_max(mx);                              // max of the RAW scores (scale > 0)
        const float mc = mx * c2;
        float l = 0.f;
        const uint32_t drk = drop_row_key(eff_seed(a.dropout_seed, a.seed_off), (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(q));
#pragma unroll
        for (int kf = 0; kf < MAXF; kf++)
            if (kf < nf) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    s[kf][r] = fast_exp2(s[kf][r] * c2 - mc);
                    l += s[kf][r];
                }
                if (drop) {
                    float keep[4];
                    drop_keep4(drk, 16 * kf + 4 * (lane >> 4), dth, dsc, keep);
#pragma unroll
                    for (int r = 0; r < 4; r++) s[kf][r] *= keep[r];
                }
            }
        l = group_sum(l);
        f32x4 acc[4];
#pragma unroll
        for (int hf = 0; hf < 4; hf++) acc[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < MAXC; c++)
            if (c < nc) {
                const bf16x8 pf = pack_frag(s[2 * c], s[2 * c + 1]);
#pragma unroll
                for (int hf = 0; hf < 4; hf++)
                    acc[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(vimg, 32 * c, lo.tv[hf]), pf, acc[hf], 0, 0, 0);
            }
        if (nf & 1) {                                    // the last 16 keys (wave-uniform; s[] only by constant index)
#pragma unroll
            for (int kf = 0; kf < MAXF; kf += 2)
                if (kf == nf - 1) {
                    const bf16x4 pf = pack_frag4(s[kf]);
#pragma unroll
                    for (int hf = 0; hf < 4; hf++)
                        acc[hf] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(tr_frag16_at(vimg, 16 * kf, lo.tv[hf]), pf, acc[hf], 0, 0, 0);
                }
        }
        mfma_fence();
        store_rows(out, D, q, q < N, acc, 1.f / l, lane);
        if (q < N && lane < 16) a.lse[(size_t(b) * a.H + h) * N + q] = mx * a.scale + __logf(l);
    }
}

// dK, dV: waves own 16-key fragments; Q and dO of the whole sequence are in LDS.
template <int NFC>
__global__ __launch_bounds__(THREADS, 2) void attn_seq_bwd_kv_kernel(const sfcvit_attn_args a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *qimg = smem, *doimg = smem + npad * 128;
    float *lse_s = reinterpret_cast<float *>(smem + 2 * npad * 128), *del_s = lse_s + npad;
    uint32_t *rkey_s = reinterpret_cast<uint32_t *>(del_s + npad);   // dropout row key of every query
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x, N = a.N, D = a.H * HD, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    const uint16_t *dop = static_cast<const uint16_t *>(a.dout) + size_t(b) * N * D + h * HD;
    const float *lse = a.lse + (size_t(b) * a.H + h) * N, *del = a.delta + (size_t(b) * a.H + h) * N;
    bf16x8 kfr[MAXOWN][2], vfr[MAXOWN][2];          // this wave's key fragments, fetched before the staging
#pragma unroll
    for (int o = 0; o < MAXOWN; o++)
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            kfr[o][kk] = global_frag(kp, ld, 16 * (wave + WAVES * o), N, kk, lane);
            vfr[o][kk] = global_frag(vp, ld, 16 * (wave + WAVES * o), N, kk, lane);
        }
    dma_seq<false>(qimg, qp, ld, N, npad, tid);
    dma_seq<false>(doimg, dop, D, N, npad, tid);
    for (int i = tid; i < npad; i += THREADS) {
        lse_s[i] = i < N ? lse[i] * 1.4426950408889634f : INFINITY;   // padded queries: p = exp2(-inf) = 0
        del_s[i] = i < N ? del[i] : 0.f;
        rkey_s[i] = drop_row_key(eff_seed(a.dropout_seed, a.seed_off), (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(i));
    }
    __syncthreads();
    const int nf = NFC ? NFC : (N + 15) >> 4, nc = NFC ? ((NFC + 1) >> 1) : npad >> 5;   // npad = 32 nc here
    const float scale = a.scale, c2 = a.scale * 1.4426950408889634f;
    const LaneOff lo = lane_offsets(lane);
    const bool drop = a.dropout_p > 0.f;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);
    uint16_t *dbase = static_cast<uint16_t *>(a.dqkv) + size_t(b) * N * ld + h * HD;

#pragma unroll
    for (int o = 0; o < MAXOWN; o++) {
        const int kfi = wave + WAVES * o;
        if (kfi >= nf) break;                            // wave-uniform
        const int key = 16 * kfi + (lane & 15);
        const bf16x8 (&kf)[2] = kfr[o];
        const bf16x8 (&vf)[2] = vfr[o];
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int hf = 0; hf < 4; hf++) dk[hf] = dv[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nc; c++) {
            f32x4 p[2], ds[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int qf = 2 * c + t;
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(qimg, 16 * qf, lo.k[kk]), kf[kk], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(doimg, 16 * qf, lo.k[kk]), vf[kk], dp, 0, 0, 0);
                }
                const int ql0 = 16 * qf + 4 * (lane >> 4);                // this lane's 4 queries: one 16-B LDS read each
                const f32x4 lse4 = *reinterpret_cast<const f32x4 *>(lse_s + ql0);    // lse * log2(e)
                const f32x4 del4 = *reinterpret_cast<const f32x4 *>(del_s + ql0);
                const u32x4 rk4 = *reinterpret_cast<const u32x4 *>(rkey_s + ql0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float pv = fast_exp2(s[r] * c2 - lse4[r]);
                    float keep = 1.f;
                    if (drop) {
                        bool k0b, k1b;
                        drop_keep2(rk4[r], uint32_t(key >> 1), dth, k0b, k1b);
                        keep = ((key & 1) ? k1b : k0b) ? dsc : 0.f;
                    }
                    p[t][r] = pv * keep;
                    ds[t][r] = pv * (dp[r] * keep - del4[r]) * scale;
                }
            }
            const bf16x8 pf = pack_frag(p[0], p[1]), dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int hf = 0; hf < 4; hf++) {
                dv[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(doimg, 32 * c, lo.t[hf]), pf, dv[hf], 0, 0, 0);
                dk[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(qimg, 32 * c, lo.t[hf]), dsf, dk[hf], 0, 0, 0);
            }
        }
        mfma_fence();
        store_rows(dbase + D, ld, key, key < N, dk, 1.f, lane);
        store_rows(dbase + 2 * D, ld, key, key < N, dv, 1.f, lane);
    }
}

// dQ: waves own 16-query fragments; K and V of the whole sequence are in LDS.
template <int NFC>
__global__ __launch_bounds__(THREADS, 3) void attn_seq_bwd_q_kernel(const sfcvit_attn_args a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *kimg = smem, *vimg = smem + npad * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x, N = a.N, D = a.H * HD, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    const uint16_t *dop = static_cast<const uint16_t *>(a.dout) + size_t(b) * N * D + h * HD;
    bf16x8 qfa[MAXOWN][2], dofa[MAXOWN][2];         // this wave's query fragments, fetched before the staging
    float lse_a[MAXOWN], del_a[MAXOWN];
#pragma unroll
    for (int o = 0; o < MAXOWN; o++) {
        const int qq = 16 * (wave + WAVES * o) + (lane & 15);
        lse_a[o] = qq < N ? a.lse[(size_t(b) * a.H + h) * N + qq] * 1.4426950408889634f : 0.f;
        del_a[o] = qq < N ? a.delta[(size_t(b) * a.H + h) * N + qq] : 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            qfa[o][kk] = global_frag(qp, ld, 16 * (wave + WAVES * o), N, kk, lane);
            dofa[o][kk] = global_frag(dop, D, 16 * (wave + WAVES * o), N, kk, lane);
        }
    }
    dma_seq<false>(kimg, kp, ld, N, npad, tid);
    dma_seq<false>(vimg, vp, ld, N, npad, tid);
    __syncthreads();
    const int nf = NFC ? NFC : (N + 15) >> 4, nc = NFC ? (NFC >> 1) : npad >> 5;
    const float scale = a.scale, c2 = a.scale * 1.4426950408889634f;
    const LaneOff lo = lane_offsets(lane);
    const bool drop = a.dropout_p > 0.f;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);
    uint16_t *dbase = static_cast<uint16_t *>(a.dqkv) + size_t(b) * N * ld + h * HD;

#pragma unroll
    for (int o = 0; o < MAXOWN; o++) {
        const int qf = wave + WAVES * o;
        if (qf >= nf) break;                             // wave-uniform
        const int q = 16 * qf + (lane & 15);
        const float lse_q = lse_a[o], del_q = del_a[o];
        const uint32_t drk = drop_row_key(eff_seed(a.dropout_seed, a.seed_off), (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(q));
        const bf16x8 (&qfr)[2] = qfa[o];
        const bf16x8 (&dof)[2] = dofa[o];
        f32x4 dq[4];
#pragma unroll
        for (int hf = 0; hf < 4; hf++) dq[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        // dS of one 16-key fragment: exp2(s c2 - lse) (dP keep - delta) scale, keys >= N zeroed
        auto ds_frag = [&](int kfi) __attribute__((always_inline)) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f}, ds;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(kimg, 16 * kfi, lo.k[kk]), qfr[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(vimg, 16 * kfi, lo.k[kk]), dof[kk], dp, 0, 0, 0);
            }
            float keep[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop) drop_keep4(drk, 16 * kfi + 4 * (lane >> 4), dth, dsc, keep);
#pragma unroll
            for (int r = 0; r < 4; r++) ds[r] = fast_exp2(s[r] * c2 - lse_q) * (dp[r] * keep[r] - del_q) * scale;
            if (16 * kfi + 16 > N) {                     // boundary fragment: keys >= N carry no gradient
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (16 * kfi + 4 * (lane >> 4) + r >= N) ds[r] = 0.f;
            }
            return ds;
        };
        for (int c = 0; c < nc; c++) {
            const f32x4 d0 = ds_frag(2 * c), d1 = ds_frag(2 * c + 1);
            const bf16x8 dsf = pack_frag(d0, d1);
#pragma unroll
            for (int hf = 0; hf < 4; hf++)
                dq[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(kimg, 32 * c, lo.t[hf]), dsf, dq[hf], 0, 0, 0);
        }
        if (nf & 1) {                                    // the last 16 keys
            const bf16x4 dsf = pack_frag4(ds_frag(nf - 1));
#pragma unroll
            for (int hf = 0; hf < 4; hf++)
                dq[hf] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(tr_frag16_at(kimg, 16 * (nf - 1), lo.t[hf]), dsf, dq[hf], 0, 0, 0);
        }
        mfma_fence();
        store_rows(dbase, ld, q, q < N, dq, 1.f, lane);
    }
}

constexpr int SEQ_MAX_N = 256;
constexpr int SEQ_MAX_LDS = 2 * SEQ_MAX_N * 128 + 3 * SEQ_MAX_N * 4;

int set_lds_limit() {
    for (const void *k : {reinterpret_cast<const void *>(&attn_seq_fwd_kernel<0, false>), reinterpret_cast<const void *>(&attn_seq_fwd_kernel<0, true>),
                          reinterpret_cast<const void *>(&attn_seq_fwd_kernel<13, false>), reinterpret_cast<const void *>(&attn_seq_fwd_kernel<13, true>),
                          reinterpret_cast<const void *>(&attn_seq_bwd_kv_kernel<0>), reinterpret_cast<const void *>(&attn_seq_bwd_q_kernel<0>),
                          reinterpret_cast<const void *>(&attn_seq_bwd_kv_kernel<13>), reinterpret_cast<const void *>(&attn_seq_bwd_q_kernel<13>)})
        if (int rc = raise_lds_limit(k, SEQ_MAX_LDS, "attention_seq attribute")) return rc;
    return SFCVIT_OK;
}

}  // namespace

int attn_seq_fwd(const sfcvit_attn_args &a, hipStream_t s) {
    if (a.N > SEQ_MAX_N) return -1;
    if (int rc = set_lds_limit()) return rc;
    const int npad = (a.N + 15) / 16 * 16;
    const int nf = (a.N + 15) / 16;
    const bool drop = a.dropout_p > 0.f;
    note_attn_kernel("attn_seq_fwd_kernel<%d, %s>", nf == 13 ? 13 : 0, drop ? "true" : "false");
    const dim3 grid(a.H, a.B), block(THREADS);
    const size_t lds = size_t(2 * npad * 128);
    if (nf == 13 && drop) hipLaunchKernelGGL((attn_seq_fwd_kernel<13, true>), grid, block, lds, s, a, npad);
    else if (nf == 13) hipLaunchKernelGGL((attn_seq_fwd_kernel<13, false>), grid, block, lds, s, a, npad);
    else if (drop) hipLaunchKernelGGL((attn_seq_fwd_kernel<0, true>), grid, block, lds, s, a, npad);
    else hipLaunchKernelGGL((attn_seq_fwd_kernel<0, false>), grid, block, lds, s, a, npad);
    return check_launch("attention_seq_fwd");
}

int attn_seq_bwd(const sfcvit_attn_args &a, hipStream_t s) {
    if (a.N > SEQ_MAX_N) return -1;
    if (int rc = set_lds_limit()) return rc;
    const int npad = (a.N + 31) / 32 * 32;
    const bool nf13 = (a.N + 15) / 16 == 13;
    note_attn_kernel("attn_seq_bwd_kv_kernel<%d>", nf13 ? 13 : 0);
    if (nf13) hipLaunchKernelGGL(attn_seq_bwd_kv_kernel<13>, dim3(a.H, a.B), dim3(THREADS), size_t(2 * npad * 128 + 3 * npad * 4), s, a, npad);
    else hipLaunchKernelGGL(attn_seq_bwd_kv_kernel<0>, dim3(a.H, a.B), dim3(THREADS), size_t(2 * npad * 128 + 3 * npad * 4), s, a, npad);
    if (int rc = check_launch("attention_seq_bwd kv")) return rc;
    const int npad16 = (a.N + 15) / 16 * 16;
    if (nf13) hipLaunchKernelGGL(attn_seq_bwd_q_kernel<13>, dim3(a.H, a.B), dim3(THREADS), size_t(2 * npad16 * 128), s, a, npad16);
    else hipLaunchKernelGGL(attn_seq_bwd_q_kernel<0>, dim3(a.H, a.B), dim3(THREADS), size_t(2 * npad16 * 128), s, a, npad16);
    return check_launch("attention_seq_bwd q");
}

}  // namespace sfcvit
