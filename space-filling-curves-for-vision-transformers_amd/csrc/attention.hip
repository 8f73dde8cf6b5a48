// Multi-head self-attention core for gfx950, head dim 64, no mask (see
// sfcvit_attention_fwd / _bwd in include/sfcvit.h).  Flash style: the N x N score
// matrix never exists in HBM; backward recomputes P from Q, K and the saved
// log-sum-exp.
//
// All products are v_mfma_f32_16x16x32_bf16.  Orientations are chosen so that the
// accumulator of one product is directly the B operand of the next one
// ("accumulator tile as the next MFMA's operand", cdna_hip_programming.md §3):
// a lane owns one query (or key) column and its registers enumerate the
// contraction index of the following product, so P / dS never cross lanes.
//
//   forward   (wave = 16 queries, loop over 64-key blocks staged in LDS)
//       S^T = K Q^T            A: K rows from LDS (b128)      B: Q from registers
//       O^T += V^T P^T         A: V^T via ds_read_b64_tr_b16  B: exp(S^T) accumulators
//   backward, dK/dV kernel (wave = 16 keys, loop over 64-query blocks in LDS)
//       S = Q K^T, dP = dO V^T A: Q / dO rows from LDS        B: K / V from registers
//       dV^T += dO^T P, dK^T += Q^T dS   A: tr-reads of dO / Q   B: P / dS accumulators
//   backward, dQ kernel    (wave = 16 queries, loop over 64-key blocks in LDS)
//       S^T = K Q^T, dP^T = V dO^T ; dQ^T += K^T dS^T  (A: tr-read of K, B: dS^T)
#include <cstdlib>
#include "attention_common.h"
#include "common_host.h"

namespace sfcvit {
namespace {

using namespace attn;

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void attn_fwd_kernel(const sfcvit_attn_args a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * IMG_BYTES];
    char *kimg = smem, *vimg = smem + IMG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, N = a.N, D = a.H * HD, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    const int q0 = blockIdx.x * BLK + wave * 16;
    const float scale = a.scale;
    const bool drop = a.dropout_p > 0.f;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);
    // mask row of this lane's query
    const uint32_t drk = drop_row_key(eff_seed(a.dropout_seed, a.seed_off), (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(q0 + (lane & 15)));

    bf16x8 qf[2];
    qf[0] = global_frag(qp, ld, q0, N, 0, lane);
    qf[1] = global_frag(qp, ld, q0, N, 1, lane);

    f32x4 o[4];
#pragma unroll
    for (int hf = 0; hf < 4; hf++) o[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < N; k0 += BLK) {
        __syncthreads();
        stage64<false>(kimg, kp, ld, k0, N, tid);
        stage64<true>(vimg, vp, ld, k0, N, tid);
        __syncthreads();

        f32x4 s[4];
#pragma unroll
        for (int kf = 0; kf < 4; kf++) {
            s[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; kk++)
                s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag(kimg, 16 * kf, kk, lane), qf[kk], s[kf], 0, 0, 0);
        }
        // s[kf][r] = S^T[key = k0 + 16kf + 4g + r][q = lane & 15]
        float mb = -INFINITY;
#pragma unroll
        for (int kf = 0; kf < 4; kf++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = k0 + 16 * kf + 4 * (lane >> 4) + r;
                s[kf][r] = key < N ? s[kf][r] * scale : -INFINITY;
                mb = fmaxf(mb, s[kf][r]);
            }
        mb = group_max(mb);
        const float m_new = fmaxf(m_run, mb);
        const float alpha = __expf(m_run - m_new);
        float ls = 0.f;
#pragma unroll
        for (int kf = 0; kf < 4; kf++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                s[kf][r] = __expf(s[kf][r] - m_new);
                ls += s[kf][r];
            }
        l_run = l_run * alpha + ls;      // the normaliser uses the un-dropped probabilities
        m_run = m_new;
        if (drop) {
#pragma unroll
            for (int kf = 0; kf < 4; kf++) {
                float keep[4];
                drop_keep4(drk, k0 + 16 * kf + 4 * (lane >> 4), dth, dsc, keep);
#pragma unroll
                for (int r = 0; r < 4; r++) s[kf][r] *= keep[r];
            }
        }
#pragma unroll
        for (int hf = 0; hf < 4; hf++) o[hf] *= alpha;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const bf16x8 pf = pack_frag(s[2 * c], s[2 * c + 1]);
#pragma unroll
            for (int hf = 0; hf < 4; hf++)
                o[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<true>(vimg, 32 * c, 32 * c + 16, 16 * hf, lane), pf,
                                                                o[hf], 0, 0, 0);
        }
    }
    mfma_fence();
    const float l_tot = group_sum(l_run);
    const int q = q0 + (lane & 15);
    uint16_t *out = static_cast<uint16_t *>(a.out) + size_t(b) * N * D + h * HD;
    store_rows(out, D, q, q < N, o, 1.f / l_tot, lane);
    if (q < N && lane < 16) a.lse[(size_t(b) * a.H + h) * N + q] = m_run + __logf(l_tot);
}

// ---------------------------------------------------------------------------
// delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void attn_delta_kernel(const uint16_t *__restrict__ dout,
                                                             const uint16_t *__restrict__ out, float *__restrict__ delta,
                                                             int B, int N, int H, int hd) {
    // one 8-lane group per (b, q, h): 8 lanes x 8 elements per 64 columns of the head
    const int64_t grp = (int64_t(blockIdx.x) * THREADS + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    const int64_t total = int64_t(B) * N * H;
    float s = 0.f;
    if (grp < total) {
        for (int c0 = 0; c0 < hd; c0 += 64) {
            const size_t off = size_t(grp) * hd + c0 + sub * 8;   // [B, N, H, hd] is contiguous
            const u32x4 x = *reinterpret_cast<const u32x4 *>(dout + off), y = *reinterpret_cast<const u32x4 *>(out + off);
#pragma unroll
            for (int i = 0; i < 4; i++)
                s += bf2f(uint16_t(x[i])) * bf2f(uint16_t(y[i])) + bf2f(uint16_t(x[i] >> 16)) * bf2f(uint16_t(y[i] >> 16));
        }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (grp < total && sub == 0) {
        const int64_t bq = grp / H;
        const int hh = int(grp % H);
        const int64_t bb = bq / N, qq = bq % N;
        delta[(bb * H + hh) * N + qq] = s;
    }
}

// ---------------------------------------------------------------------------
// backward: dK, dV (one workgroup = 64 keys of one (b, h); wave = 16 keys)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void attn_bwd_kv_kernel(const sfcvit_attn_args a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * IMG_BYTES + 2 * BLK * 4];
    char *qimg = smem, *doimg = smem + IMG_BYTES;
    float *lse_s = reinterpret_cast<float *>(smem + 2 * IMG_BYTES), *del_s = lse_s + BLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, N = a.N, D = a.H * HD, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    const uint16_t *dop = static_cast<const uint16_t *>(a.dout) + size_t(b) * N * D + h * HD;
    const float *lse = a.lse + (size_t(b) * a.H + h) * N, *del = a.delta + (size_t(b) * a.H + h) * N;
    const int key0 = blockIdx.x * BLK + wave * 16;
    const float scale = a.scale;
    const bool drop = a.dropout_p > 0.f;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);
    const int dkey = key0 + (lane & 15);
    const uint64_t dbh = (uint64_t(b) * a.H + h) * uint64_t(N);

    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
        kf[kk] = global_frag(kp, ld, key0, N, kk, lane);
        vf[kk] = global_frag(vp, ld, key0, N, kk, lane);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int hf = 0; hf < 4; hf++) dk[hf] = dv[hf] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int q0 = 0; q0 < N; q0 += BLK) {
        __syncthreads();
        stage64<false>(qimg, qp, ld, q0, N, tid);
        stage64<false>(doimg, dop, D, q0, N, tid);
        if (tid < BLK) {
            lse_s[tid] = q0 + tid < N ? lse[q0 + tid] : 0.f;
            del_s[tid] = q0 + tid < N ? del[q0 + tid] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; c++) {
            f32x4 p[2], ds[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int qf = 2 * c + t;
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag(qimg, 16 * qf, kk, lane), kf[kk], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag(doimg, 16 * qf, kk, lane), vf[kk], dp, 0, 0, 0);
                }
                // s[r] = S[q = q0 + 16qf + 4g + r][key = key0 + (lane&15)]
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int ql = 16 * qf + 4 * (lane >> 4) + r;
                    const float pv = __expf(s[r] * scale - lse_s[ql]);
                    float keep = 1.f;
                    if (drop) {
                        bool k0b, k1b;
                        drop_keep2(drop_row_key(eff_seed(a.dropout_seed, a.seed_off), dbh + uint64_t(q0 + ql)), uint32_t(dkey >> 1), dth, k0b, k1b);
                        keep = ((dkey & 1) ? k1b : k0b) ? dsc : 0.f;
                    }
                    p[t][r] = pv * keep;                                   // dropped probabilities feed dV
                    ds[t][r] = pv * (dp[r] * keep - del_s[ql]) * scale;
                }
            }
            const bf16x8 pf = pack_frag(p[0], p[1]), dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int hf = 0; hf < 4; hf++) {
                dv[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<false>(doimg, 32 * c, 32 * c + 16, 16 * hf, lane), pf,
                                                                 dv[hf], 0, 0, 0);
                dk[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<false>(qimg, 32 * c, 32 * c + 16, 16 * hf, lane), dsf,
                                                                 dk[hf], 0, 0, 0);
            }
        }
    }
    const int key = key0 + (lane & 15);
    uint16_t *dbase = static_cast<uint16_t *>(a.dqkv) + size_t(b) * N * ld + h * HD;
    mfma_fence();
    store_rows(dbase + D, ld, key, key < N, dk, 1.f, lane);
    store_rows(dbase + 2 * D, ld, key, key < N, dv, 1.f, lane);
}

// ---------------------------------------------------------------------------
// backward: dQ (one workgroup = 64 queries of one (b, h); wave = 16 queries)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void attn_bwd_q_kernel(const sfcvit_attn_args a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * IMG_BYTES];
    char *kimg = smem, *vimg = smem + IMG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, N = a.N, D = a.H * HD, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * HD;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    const uint16_t *dop = static_cast<const uint16_t *>(a.dout) + size_t(b) * N * D + h * HD;
    const int q0 = blockIdx.x * BLK + wave * 16;
    const int q = q0 + (lane & 15);
    const float scale = a.scale;
    const float lse_q = q < N ? a.lse[(size_t(b) * a.H + h) * N + q] : 0.f;
    const float del_q = q < N ? a.delta[(size_t(b) * a.H + h) * N + q] : 0.f;
    const bool drop = a.dropout_p > 0.f;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);
    const uint32_t drk = drop_row_key(eff_seed(a.dropout_seed, a.seed_off), (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(q));

    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
        qf[kk] = global_frag(qp, ld, q0, N, kk, lane);
        dof[kk] = global_frag(dop, D, q0, N, kk, lane);
    }
    f32x4 dq[4];
#pragma unroll
    for (int hf = 0; hf < 4; hf++) dq[hf] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < N; k0 += BLK) {
        __syncthreads();
        stage64<false>(kimg, kp, ld, k0, N, tid);
        stage64<false>(vimg, vp, ld, k0, N, tid);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; c++) {
            f32x4 ds[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int kfi = 2 * c + t;
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag(kimg, 16 * kfi, kk, lane), qf[kk], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag(vimg, 16 * kfi, kk, lane), dof[kk], dp, 0, 0, 0);
                }
                // s[r] = S^T[key = k0 + 16kfi + 4g + r][q]; keys >= N have K = V = 0 and add nothing
                float keep[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop) drop_keep4(drk, k0 + 16 * kfi + 4 * (lane >> 4), dth, dsc, keep);
#pragma unroll
                for (int r = 0; r < 4; r++) ds[t][r] = __expf(s[r] * scale - lse_q) * (dp[r] * keep[r] - del_q) * scale;
            }
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int hf = 0; hf < 4; hf++)
                dq[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<false>(kimg, 32 * c, 32 * c + 16, 16 * hf, lane), dsf,
                                                                 dq[hf], 0, 0, 0);
        }
    }
    uint16_t *dbase = static_cast<uint16_t *>(a.dqkv) + size_t(b) * N * ld + h * HD;
    mfma_fence();
    store_rows(dbase, ld, q, q < N, dq, 1.f, lane);
}

int check_args(const sfcvit_attn_args *a, const char *what, bool bwd) {
    if (!a || !a->qkv || !a->out || !a->lse) return fail(SFCVIT_EINVAL, "%s: null pointer", what);
    if (bwd && (!a->dout || !a->dqkv || !a->delta)) return fail(SFCVIT_EINVAL, "%s: null pointer", what);
    if (a->hd != 64 && a->hd != 128 && a->hd != 192 && a->hd != 256)
        return fail(SFCVIT_EINVAL, "%s: head dim %d not supported (64, 128, 192, 256)", what, a->hd);
    if (!(a->dropout_p >= 0.f && a->dropout_p < 1.f)) return fail(SFCVIT_EINVAL, "%s: dropout_p=%g out of [0, 1)", what, a->dropout_p);
    if (a->B <= 0 || a->N <= 0 || a->H <= 0 || a->B > 65535 || a->H > 65535)
        return fail(SFCVIT_EINVAL, "%s: B=%d N=%d H=%d", what, a->B, a->N, a->H);
    if (!aligned16(a->qkv) || !aligned16(a->out) || (bwd && (!aligned16(a->dout) || !aligned16(a->dqkv))))
        return fail(SFCVIT_EINVAL, "%s: tensors must be 16-byte aligned", what);
    return SFCVIT_OK;
}

}  // namespace

// attention_seq.hip: whole-sequence kernels; return -1 when N is too long for them.
int attn_seq_fwd(const sfcvit_attn_args &a, hipStream_t s);
int attn_seq_bwd(const sfcvit_attn_args &a, hipStream_t s);
// attention_bwd_fused.hip: dK, dV and dQ in one pass (head dim 64, N <= 224); -1 when not eligible.
int attn_seq_bwd_fused(const sfcvit_attn_args &a, int dq_sums, hipStream_t s);
// attention_wide.hip: head dims 128 / 192 / 256; return -1 for head dim 64.
int attn_wide_fwd(const sfcvit_attn_args &a, hipStream_t s);
int attn_wide_bwd(const sfcvit_attn_args &a, hipStream_t s);
// attention_long.hip: forward with K / V of the whole sequence resident, head dim 64, 256 < N <= 608; -1 otherwise.
int attn_long_fwd(const sfcvit_attn_args &a, hipStream_t s);
int attn_long_bwd(const sfcvit_attn_args &a, hipStream_t s);

}  // namespace sfcvit

using namespace sfcvit;

extern "C" int sfcvit_attention_fwd(const sfcvit_attn_args *a, void *stream) {
    if (int rc = check_args(a, "attention_fwd", false)) return rc;
    if (int rc = attn_wide_fwd(*a, static_cast<hipStream_t>(stream)); rc >= 0) return rc;
    if (int rc = attn_seq_fwd(*a, static_cast<hipStream_t>(stream)); rc >= 0) return rc;
    const char *el = getenv("SFCVIT_ATTN_LONG");             // "0": tiled kernel for every N > 256 (A/B, tests); read per call
    if (!(el && el[0] == '0'))
        if (int rc = attn_long_fwd(*a, static_cast<hipStream_t>(stream)); rc >= 0) return rc;
    dim3 grid((a->N + BLK - 1) / BLK, a->H, a->B);
    note_attn_kernel("attn_fwd_kernel");
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(THREADS), 0, static_cast<hipStream_t>(stream), *a);
    return check_launch("attention_fwd");
}

extern "C" int64_t sfcvit_attention_colsum_workspace(int B, int N, int H, int hd) {
    if (B <= 0 || N <= 0 || H <= 0 || hd <= 0) return 0;
    // one-pass backward: its [B][3D] partial rows, and behind them the partials of the pass over the Q third (two regions:
    // with deferred reductions both are read at the flush)
    const int64_t fused = int64_t(B) * 3 * H * hd * int64_t(sizeof(float)) + sfcvit_colsum_workspace(B * N, H * hd);
    const int64_t generic = sfcvit_colsum_workspace(B * N, 3 * H * hd);
    return fused > generic ? fused : generic;
}

extern "C" int sfcvit_attention_bwd(const sfcvit_attn_args *a, void *stream) {
    if (int rc = check_args(a, "attention_bwd", true)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int D3 = 3 * a->H * a->hd;
    if (a->colsum_out && (!a->colsum_part || a->colsum_part_bytes < sfcvit_attention_colsum_workspace(a->B, a->N, a->H, a->hd)))
        return fail(SFCVIT_EINVAL, "attention_bwd: colsum_out needs colsum_part of sfcvit_attention_colsum_workspace bytes");
    {   // one pass: dK, dV, dQ, delta and the column sums from a single evaluation of P and dS (hd = 64, N <= 224)
        const char *env = getenv("SFCVIT_ATTN_BWD_FUSED");       // "0": the two-kernel form (A/B measurements, tests)
        if (!(env && env[0] == '0')) {
            sfcvit_attn_args f = *a;
            if (!a->colsum_out) f.colsum_part = nullptr;
            // The column sums of dK and dV leave the kernel as 128 floats per item (its key waves hold whole columns).  Those of
            // dQ come from the key waves as well since round 4: sum_q dQ[q, :] = scale sum_k (sum_q dS[q, k]) K[k, :], one add
            // per score in the loop and a 16 x 64 product per wave after it.  (Round 3 took them from the two dQ waves -- per-chunk
            // lane reductions + LDS read-modify-writes on the waves a step waits for, +37 us per launch -- and therefore
            // defaulted to a separate 16-us pass over the Q third of dqkv, which SFCVIT_ATTN_DQSUM=pass still selects: A/B.)
            const char *dq = getenv("SFCVIT_ATTN_DQSUM");
            const int dq_in_kernel = !(dq && dq[0] == 'p');
            if (int rc = attn_seq_bwd_fused(f, dq_in_kernel, s); rc >= 0) {
                if (rc || !a->colsum_out) return rc;
                if (dq_in_kernel) return launch_colsum_reduce(a->colsum_part, a->B, D3, a->colsum_out, a->colsum_bf16, stream);
                // K | V thirds from the kernel's partial rows, the Q third (columns 0 .. D-1 of dqkv) from its own pass; disjoint
                // outputs and disjoint partial regions, so the two reductions may run in one deferred batch
                const int Dq = a->H * a->hd;
                char *outp = static_cast<char *>(a->colsum_out);
                if (int rc2 = reduce_cols(a->colsum_part + Dq, a->B, D3, D3 - Dq, outp + size_t(Dq) * (a->colsum_bf16 ? 2 : 4), a->colsum_bf16, stream))
                    return rc2;
                const int64_t head = int64_t(a->B) * D3 * int64_t(sizeof(float));
                return sfcvit_colsum(a->dqkv, a->B * a->N, Dq, D3, a->colsum_out, a->colsum_bf16, a->colsum_part + size_t(a->B) * D3,
                                     a->colsum_part_bytes - head, stream);
            }
        }
    }
    {   // sequence-resident kernels (hd = 64, 256 < N <= 608): delta comes out of their dQ kernel, the column sums too
        const char *el = getenv("SFCVIT_ATTN_LONG");         // "0": tiled kernels for every N > 256 (A/B, tests)
        if (!(el && el[0] == '0')) {
            sfcvit_attn_args f = *a;
            if (!a->colsum_out) f.colsum_part = nullptr;
            if (int rc = attn_long_bwd(f, s); rc >= 0) {
                if (rc || !a->colsum_out) return rc;
                return launch_colsum_reduce(a->colsum_part, a->B, D3, a->colsum_out, a->colsum_bf16, stream);
            }
        }
    }
    const int rc_rest = [&]() -> int {
    const int64_t groups = int64_t(a->B) * a->N * a->H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(unsigned((groups * 8 + THREADS - 1) / THREADS)), dim3(THREADS), 0, s,
                       static_cast<const uint16_t *>(a->dout), static_cast<const uint16_t *>(a->out), a->delta, a->B, a->N, a->H, a->hd);
    if (int rc = check_launch("attention_bwd delta")) return rc;
    if (int rc = attn_wide_bwd(*a, s); rc >= 0) return rc;
    if (int rc = attn_seq_bwd(*a, s); rc >= 0) return rc;
    dim3 grid((a->N + BLK - 1) / BLK, a->H, a->B);
    note_attn_kernel("attn_bwd_kv_kernel");
    hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, dim3(THREADS), 0, s, *a);
    if (int rc = check_launch("attention_bwd kv")) return rc;
    hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(THREADS), 0, s, *a);
    return check_launch("attention_bwd q");
    }();
    if (rc_rest || !a->colsum_out) return rc_rest;
    return sfcvit_colsum(a->dqkv, a->B * a->N, D3, D3, a->colsum_out, a->colsum_bf16, a->colsum_part, a->colsum_part_bytes, stream);
}
