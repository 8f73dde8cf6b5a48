// Long-sequence attention for gfx950, forward and the two backward kernels: head dim 64, 256 < N <= 608 (ViT-L/16 at
// 384 x 384: N = 576).
//
// K and V of one (batch, head) -- 2 x 72 KiB at N = 576 -- are staged into LDS ONCE by LDS-DMA and stay there; the
// tiled kernel of attention.hip re-staged them for every 64 queries (9 x per head at N = 576) with two barriers per 64
// keys.  Round 1 tried this residency with the 4-wave kernels of attention_seq.hip and found it slower: one wave per
// SIMD hides none of the exp / reduction latency between the MFMA bursts.  Here the workgroup is 12 waves (3 per SIMD;
// 36 query fragments at N = 576 = exactly 3 per wave) and a wave's score row is cut into chunks of 12 key fragments
// (192 keys: 48 accumulator registers) with the online-softmax rescale between chunks -- 3 rescales per row at
// N = 576, against 9 in the tiled kernel.  No barrier after the staging.  MFMA orientations, LDS images and the
// dropout mask are those of attention_seq.hip / attention.hip (S^T = K Q^T with K rows from LDS, O^T += V^T P^T with
// V^T by ds_read_b64_tr_b16 and the exp'd accumulators as the B operand), so backward (attention.hip) regenerates the
// same mask from (seed, row, key).
#include "attention_common.h"
#include "common_host.h"

namespace sfcvit {
namespace {

using namespace attn;

constexpr int LT = 768, LW = LT / 64;   // 12 waves
constexpr int CKF = 12;                 // key fragments per softmax chunk
constexpr int LONG_MAX_N = 608;         // 2 images x 608 rows x 128 B = 152 KiB

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <bool VT>
__device__ __forceinline__ void dma_long(char *img, const uint16_t *__restrict__ src, int ld, int N, int npad, int tid) {
    for (int p = tid; p < npad * 8; p += LT) {           // rows >= N: copies of row N - 1 (finite; their keys are masked)
        const int row = p >> 3, cs = p & 7;
        const int c = cs ^ kc_swz(row);        // "kc" and "vt" images share one swizzle now (device_common.h)
        const uint16_t *g = src + size_t(min(row, N - 1)) * ld + c * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(img + p * 16), 16, 0, 0);
    }
}

// NFP = padded key fragments (npad / 16, even) as a compile-time constant, 0 = run time.
template <int NFP>
__global__ __launch_bounds__(LT) void attn_long_fwd_kernel(const sfcvit_attn_args a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *kimg = smem, *vimg = smem + npad * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x, N = a.N, D = a.H * HD, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    uint16_t *out = static_cast<uint16_t *>(a.out) + size_t(b) * N * D + h * HD;
    const int nfp = NFP ? NFP : npad >> 4, nqf = (N + 15) >> 4;
    bf16x8 qfr[2];
    qfr[0] = global_frag(qp, ld, 16 * wave, N, 0, lane);           // first query fragment: its latency hides behind the staging
    qfr[1] = global_frag(qp, ld, 16 * wave, N, 1, lane);
    dma_long<false>(kimg, kp, ld, N, npad, tid);
    dma_long<true>(vimg, vp, ld, N, npad, tid);
    __syncthreads();                                               // LDS-DMA pending: hipcc drains vmcnt(0) here
    const float c2 = a.scale * 1.4426950408889634f;               // exp(x * scale) = exp2(x * c2)
    const LaneOff lo = lane_offsets(lane);
    const bool drop = a.dropout_p > 0.f;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);
    const uint32_t seed = eff_seed(a.dropout_seed, a.seed_off);

    for (int qf = wave; qf < nqf; qf += LW) {
        const int q = 16 * qf + (lane & 15);
        bf16x8 qn[2];                                              // next fragment of this wave, in flight during this one
        qn[0] = global_frag(qp, ld, 16 * (qf + LW), N, 0, lane);
        qn[1] = global_frag(qp, ld, 16 * (qf + LW), N, 1, lane);
        const uint32_t drk = drop_row_key(seed, (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(q));
        f32x4 acc[4];
#pragma unroll
        for (int hf = 0; hf < 4; hf++) acc[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;                      // running max of the RAW scores; per-lane partial row sum
        for (int c0 = 0; c0 < nfp; c0 += CKF) {
            f32x4 s[CKF];
#pragma unroll
            for (int j = 0; j < CKF; j++) {
                s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c0 + j < nfp) {
#pragma unroll
                    for (int kk = 0; kk < 2; kk++)
                        s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(kimg, 16 * (c0 + j), lo.k[kk]), qfr[kk], s[j], 0, 0, 0);
                }
            }
            mfma_fence();
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < CKF; j++)
                if (c0 + j < nfp) {
                    if (16 * (c0 + j) + 16 > N) {                  // boundary and padding fragments: mask keys >= N
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (16 * (c0 + j) + 4 * (lane >> 4) + r >= N) s[j][r] = -INFINITY;
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) mx = fmaxf(mx, s[j][r]);
                }
            mx = group_max(mx);
            const float m_new = fmaxf(m_run, mx);                  // finite from the first chunk on (key 0 is valid)
            const float alpha = fast_exp2((m_run - m_new) * c2);   // first chunk: exp2(-inf) = 0
            const float mc = m_new * c2;
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int hf = 0; hf < 4; hf++) acc[hf] *= alpha;
#pragma unroll
            for (int j = 0; j < CKF; j++)
                if (c0 + j < nfp) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        s[j][r] = fast_exp2(s[j][r] * c2 - mc);
                        l_run += s[j][r];                          // the normaliser uses the un-dropped probabilities
                    }
                    if (drop) {
                        float keep[4];
                        drop_keep4(drk, 16 * (c0 + j) + 4 * (lane >> 4), dth, dsc, keep);
#pragma unroll
                        for (int r = 0; r < 4; r++) s[j][r] *= keep[r];
                    }
                }
#pragma unroll
            for (int c = 0; c < CKF / 2; c++)
                if (c0 + 2 * c < nfp) {                            // nfp is even: fragments come in pairs
                    const bf16x8 pf = pack_frag(s[2 * c], s[2 * c + 1]);
#pragma unroll
                    for (int hf = 0; hf < 4; hf++)
                        acc[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(vimg, 16 * (c0 + 2 * c), lo.tv[hf]), pf, acc[hf], 0, 0, 0);
                }
        }
        mfma_fence();
        const float l = group_sum(l_run);
        store_rows(out, D, q, q < N, acc, 1.f / l, lane);
        if (q < N && lane < 16) a.lse[(size_t(b) * a.H + h) * N + q] = m_run * a.scale + __logf(l);
        qfr[0] = qn[0];
        qfr[1] = qn[1];
    }
}

// Column sums of the workgroup's slice of dqkv (the in_proj bias gradient, sfcvit_attn_args.colsum_part).  A stored
// fragment x[hf][r] (row = lane & 15, column 16 hf + 4 (lane >> 4) + r) is summed over its 16 rows with DPP (every lane
// of a DPP row then holds all 16 sums of its 4-column group) and lane i keeps sum number i: ONE running register per
// lane instead of 16 (the dK/dV kernel has no 32 registers to spare).  `valid` masks rows that are not stored.
__device__ __forceinline__ void colsum_add(float &run, const f32x4 (&x)[4], bool valid, int lane) {
#pragma unroll
    for (int hf = 0; hf < 4; hf++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float v = row16_sum(valid ? x[hf][r] : 0.f);
            if ((lane & 15) == 4 * hf + r) run += v;
        }
}
// Waves are summed through LDS (the images are dead by now); 64 threads write the (batch, head) partial that
// launch_colsum_reduce sums over the batch.  Fixed order throughout: bit-reproducible.
__device__ __forceinline__ void wg_colsum(float run, char *smem, float *__restrict__ dst, int tid) {
    const int lane = tid & 63, wave = tid >> 6, i = lane & 15;
    float *red = reinterpret_cast<float *>(smem);
    __syncthreads();                                                  // every wave is done with the LDS images
    red[wave * 64 + 16 * (i >> 2) + 4 * (lane >> 4) + (i & 3)] = run;
    __syncthreads();
    if (tid < 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < LW; w++) t += red[w * 64 + tid];
        dst[tid] = t;
    }
}

// ---------------------------------------------------------------------------
// backward, dK / dV: Q and dO of the whole (batch, head) resident (kc images, 2 x 72 KiB at N = 576) plus lse, delta and
// the dropout row keys of every query (12 B per row); the 12 waves own 16-key fragments (3 each at N = 576) and walk all
// queries in 32-row chunks with no barrier.  Same arithmetic as attn_seq_bwd_kv_kernel (attention_seq.hip).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(LT) void attn_long_bwd_kv_kernel(const sfcvit_attn_args a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *qimg = smem, *doimg = smem + npad * 128;
    float *lse_s = reinterpret_cast<float *>(smem + 2 * npad * 128), *del_s = lse_s + npad;
    uint32_t *rkey_s = reinterpret_cast<uint32_t *>(del_s + npad);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x, N = a.N, D = a.H * HD, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    const uint16_t *dop = static_cast<const uint16_t *>(a.dout) + size_t(b) * N * D + h * HD;
    const float *lse = a.lse + (size_t(b) * a.H + h) * N, *del = a.delta + (size_t(b) * a.H + h) * N;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
        kf[kk] = global_frag(kp, ld, 16 * wave, N, kk, lane);
        vf[kk] = global_frag(vp, ld, 16 * wave, N, kk, lane);
    }
    dma_long<false>(qimg, qp, ld, N, npad, tid);
    dma_long<false>(doimg, dop, D, N, npad, tid);
    const uint32_t seed = eff_seed(a.dropout_seed, a.seed_off);
    for (int i = tid; i < npad; i += LT) {
        lse_s[i] = i < N ? lse[i] * 1.4426950408889634f : INFINITY;   // padded queries: p = exp2(-inf) = 0
        del_s[i] = i < N ? del[i] : 0.f;
        rkey_s[i] = drop_row_key(seed, (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(i));
    }
    __syncthreads();
    const int nf = (N + 15) >> 4, nc = npad >> 5;
    const float scale = a.scale, c2 = a.scale * 1.4426950408889634f;
    const LaneOff lo = lane_offsets(lane);
    const bool drop = a.dropout_p > 0.f;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);
    uint16_t *dbase = static_cast<uint16_t *>(a.dqkv) + size_t(b) * N * ld + h * HD;

    float csk = 0.f, csv = 0.f;                                       // running column sums of this wave's dK / dV rows (colsum_add)
    for (int kfi = wave; kfi < nf; kfi += LW) {
        const int key = 16 * kfi + (lane & 15);
        bf16x8 kn[2], vn[2];                                          // the wave's next key fragment, in flight during this one
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            kn[kk] = global_frag(kp, ld, 16 * (kfi + LW), N, kk, lane);
            vn[kk] = global_frag(vp, ld, 16 * (kfi + LW), N, kk, lane);
        }
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int hf = 0; hf < 4; hf++) dk[hf] = dv[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nc; c++) {
            f32x4 p[2], ds[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int qf = 2 * c + t;
                f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(qimg, 16 * qf, lo.k[kk]), kf[kk], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(doimg, 16 * qf, lo.k[kk]), vf[kk], dp, 0, 0, 0);
                }
                const int ql0 = 16 * qf + 4 * (lane >> 4);            // this lane's 4 queries: one 16-B LDS read each
                const f32x4 lse4 = *reinterpret_cast<const f32x4 *>(lse_s + ql0);
                const f32x4 del4 = *reinterpret_cast<const f32x4 *>(del_s + ql0);
                const u32x4 rk4 = *reinterpret_cast<const u32x4 *>(rkey_s + ql0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float pv = fast_exp2(sc[r] * c2 - lse4[r]);
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
        if (a.colsum_part) {                                          // padding keys hold non-zero dV (p = exp2(-lse)): not stored, not summed
            colsum_add(csk, dk, key < N, lane);
            colsum_add(csv, dv, key < N, lane);
        }
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            kf[kk] = kn[kk];
            vf[kk] = vn[kk];
        }
    }
    if (a.colsum_part) {
        float *part = a.colsum_part + size_t(b) * 3 * D + h * HD;
        wg_colsum(csk, smem, part + D, tid);
        wg_colsum(csv, smem, part + 2 * D, tid);
    }
}

// ---------------------------------------------------------------------------
// backward, dQ: K and V resident (both kc images); the 12 waves own 16-query fragments and walk all keys in 32-key chunks.
// Same arithmetic as attn_seq_bwd_q_kernel; the sequence is padded to 32 keys, padding keys carry dS = 0.
// Runs FIRST and also produces delta[b, h, q] = sum_d dO O for the dK/dV kernel: the dO fragment is in registers anyway,
// the matching O fragment is two more 16-byte loads per lane -- no separate delta pass over O and dO (28 us per layer).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(LT) void attn_long_bwd_q_kernel(const sfcvit_attn_args a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *kimg = smem, *vimg = smem + npad * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x, N = a.N, D = a.H * HD, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    const uint16_t *dop = static_cast<const uint16_t *>(a.dout) + size_t(b) * N * D + h * HD;
    const uint16_t *op = static_cast<const uint16_t *>(a.out) + size_t(b) * N * D + h * HD;
    const float *lse = a.lse + (size_t(b) * a.H + h) * N;
    float *del = a.delta + (size_t(b) * a.H + h) * N;
    // sum over the 64 head columns of dO * O for row (lane & 15): 16 products per lane, then over the 4 lane groups
    auto row_delta = [&](const bf16x8 (&x)[2], const bf16x8 (&y)[2]) __attribute__((always_inline)) {
        float t = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int e = 0; e < 8; e++) t += bf2f(uint16_t(x[kk][e])) * bf2f(uint16_t(y[kk][e]));
        return group_sum(t);
    };
    bf16x8 qfr[2], dof[2], ofr[2];
    float lse_q;
    {
        const int qq = 16 * wave + (lane & 15);
        lse_q = qq < N ? lse[qq] * 1.4426950408889634f : 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            qfr[kk] = global_frag(qp, ld, 16 * wave, N, kk, lane);
            dof[kk] = global_frag(dop, D, 16 * wave, N, kk, lane);
            ofr[kk] = global_frag(op, D, 16 * wave, N, kk, lane);
        }
    }
    dma_long<false>(kimg, kp, ld, N, npad, tid);
    dma_long<false>(vimg, vp, ld, N, npad, tid);
    __syncthreads();
    const int nqf = (N + 15) >> 4, nc = npad >> 5;
    const float scale = a.scale, c2 = a.scale * 1.4426950408889634f;
    const LaneOff lo = lane_offsets(lane);
    const bool drop = a.dropout_p > 0.f;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);
    const uint32_t seed = eff_seed(a.dropout_seed, a.seed_off);
    uint16_t *dbase = static_cast<uint16_t *>(a.dqkv) + size_t(b) * N * ld + h * HD;

    float csq = 0.f;                                                  // running column sums of this wave's dQ rows (colsum_add)
    for (int qf = wave; qf < nqf; qf += LW) {
        const int q = 16 * qf + (lane & 15), qn = q + 16 * LW;
        bf16x8 qnx[2], donx[2], onx[2];                               // next fragment of this wave
        const float lse_n = qn < N ? lse[qn] * 1.4426950408889634f : 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            qnx[kk] = global_frag(qp, ld, 16 * (qf + LW), N, kk, lane);
            donx[kk] = global_frag(dop, D, 16 * (qf + LW), N, kk, lane);
            onx[kk] = global_frag(op, D, 16 * (qf + LW), N, kk, lane);
        }
        const float del_q = row_delta(dof, ofr);                      // rows >= N: zero fragments -> 0
        if (q < N && lane < 16) del[q] = del_q;
        const uint32_t drk = drop_row_key(seed, (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(q));
        f32x4 dq[4];
#pragma unroll
        for (int hf = 0; hf < 4; hf++) dq[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto ds_frag = [&](int kfi) __attribute__((always_inline)) {
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f}, ds;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(kimg, 16 * kfi, lo.k[kk]), qfr[kk], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(vimg, 16 * kfi, lo.k[kk]), dof[kk], dp, 0, 0, 0);
            }
            float keep[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop) drop_keep4(drk, 16 * kfi + 4 * (lane >> 4), dth, dsc, keep);
#pragma unroll
            for (int r = 0; r < 4; r++) ds[r] = fast_exp2(sc[r] * c2 - lse_q) * (dp[r] * keep[r] - del_q) * scale;
            if (16 * kfi + 16 > N) {                                  // boundary / padding fragment: keys >= N carry no gradient
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
        mfma_fence();
        store_rows(dbase, ld, q, q < N, dq, 1.f, lane);
        if (a.colsum_part) colsum_add(csq, dq, q < N, lane);
        lse_q = lse_n;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            qfr[kk] = qnx[kk];
            dof[kk] = donx[kk];
            ofr[kk] = onx[kk];
        }
    }
    if (a.colsum_part) wg_colsum(csq, smem, a.colsum_part + size_t(b) * 3 * D + h * HD, tid);
}

constexpr int LONG_MAX_LDS = 2 * ((LONG_MAX_N + 31) / 32 * 32) * 128 + 3 * ((LONG_MAX_N + 31) / 32 * 32) * 4;

}  // namespace

// -1: not eligible; else a status.  a.delta is written here (dQ kernel) and read by the dK/dV kernel.
int attn_long_bwd(const sfcvit_attn_args &a, hipStream_t s) {
    if (a.hd != HD || a.N <= 256 || a.N > LONG_MAX_N) return -1;
    const int npad = (a.N + 31) / 32 * 32;
    for (const void *k : {reinterpret_cast<const void *>(&attn_long_bwd_kv_kernel), reinterpret_cast<const void *>(&attn_long_bwd_q_kernel)})
        if (int rc = raise_lds_limit(k, LONG_MAX_LDS, "attention_long attribute")) return rc;
    note_attn_kernel("attn_long_bwd_kv_kernel");
    hipLaunchKernelGGL(attn_long_bwd_q_kernel, dim3(a.H, a.B), dim3(LT), size_t(2 * npad * 128), s, a, npad);
    if (int rc = check_launch("attention_long_bwd q")) return rc;
    hipLaunchKernelGGL(attn_long_bwd_kv_kernel, dim3(a.H, a.B), dim3(LT), size_t(2 * npad * 128 + 3 * npad * 4), s, a, npad);
    return check_launch("attention_long_bwd kv");
}

// -1: not eligible (the caller takes the tiled kernel); else a status.
int attn_long_fwd(const sfcvit_attn_args &a, hipStream_t s) {
    if (a.hd != HD || a.N <= 256 || a.N > LONG_MAX_N) return -1;
    const int npad = (a.N + 31) / 32 * 32;
    const int lds = 2 * npad * 128;
    for (const void *k : {reinterpret_cast<const void *>(&attn_long_fwd_kernel<0>), reinterpret_cast<const void *>(&attn_long_fwd_kernel<36>)})
        if (int rc = raise_lds_limit(k, 2 * ((LONG_MAX_N + 31) / 32 * 32) * 128, "attention_long attribute")) return rc;
    note_attn_kernel("attn_long_fwd_kernel<%d>", npad == 576 ? 36 : 0);
    if (npad == 576) hipLaunchKernelGGL(attn_long_fwd_kernel<36>, dim3(a.H, a.B), dim3(LT), lds, s, a, npad);
    else hipLaunchKernelGGL(attn_long_fwd_kernel<0>, dim3(a.H, a.B), dim3(LT), lds, s, a, npad);
    return check_launch("attention_long_fwd");
}

}  // namespace sfcvit
