// Whole-sequence attention for head dims 128 / 192 / 256 (= 64 S, S = 2..4), N <= 256.
//
// The reference's own script builds VisionTransformer1D(embed_dim 768, n_heads 4) -- head dim 192 (main.py:276-282);
// T / B / L all use 64, which attention_seq.hip and attention.hip are tuned for.  These kernels are the same
// algorithm (one workgroup per (batch, head), K / V or Q / dO of the whole sequence staged once by LDS-DMA, a whole
// score row in registers, accumulator-as-operand MFMA orientation) with the head dimension cut into S slices of
// 64 columns: every slice has its own 128-byte-row LDS image (so the fragment addressing and bank swizzles of the
// 64-wide kernels apply unchanged), scores sum over the slices, outputs are produced per slice.  Query / key
// fragments are fetched per 16-row fragment instead of up front (S x as many registers otherwise).  One workgroup
// per CU at S = 3, N = 196 (156 KiB of LDS): correctness and coverage first, not tuned like the 64-wide path.
#include "attention_common.h"
#include "common_host.h"

namespace sfcvit {
namespace {

using namespace attn;

constexpr int MAXF = 16, MAXC = 8, WAVES = THREADS / 64;

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// npad rows x 64 columns (one slice) of a [N, ld] matrix -> LDS image, by LDS-DMA (rows >= N copy row N-1).
template <bool VT>
__device__ __forceinline__ void dma_slice(char *img, const uint16_t *__restrict__ src, int ld, int N, int npad, int tid) {
    for (int p = tid; p < npad * 8; p += THREADS) {
        const int row = p >> 3, cs = p & 7;
        const int c = cs ^ kc_swz(row);        // "kc" and "vt" images share one swizzle now (device_common.h)
        const uint16_t *g = src + size_t(min(row, N - 1)) * ld + c * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(img + p * 16), 16, 0, 0);
    }
}

template <int S>
__global__ __launch_bounds__(THREADS) void attn_wide_fwd_kernel(const sfcvit_attn_args a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int img = npad * 128;
    char *kimg = smem, *vimg = smem + S * img;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x, N = a.N, hd = 64 * S, D = a.H * hd, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * hd;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    uint16_t *out = static_cast<uint16_t *>(a.out) + size_t(b) * N * D + h * hd;
    const int nf = npad >> 4, nc = npad >> 5, nqf = (N + 15) >> 4;   // key fragments (padded to 32 keys), query fragments
#pragma unroll
    for (int sl = 0; sl < S; sl++) {
        dma_slice<false>(kimg + sl * img, kp + 64 * sl, ld, N, npad, tid);
        dma_slice<true>(vimg + sl * img, vp + 64 * sl, ld, N, npad, tid);
    }
    __syncthreads();
    const float c2 = a.scale * 1.4426950408889634f;
    const LaneOff lo = lane_offsets(lane);
    const bool drop = a.dropout_p > 0.f;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);
    for (int qf = wave; qf < nqf; qf += WAVES) {
        const int q = 16 * qf + (lane & 15);
        bf16x8 qfr[S][2];
#pragma unroll
        for (int sl = 0; sl < S; sl++)
#pragma unroll
            for (int kk = 0; kk < 2; kk++) qfr[sl][kk] = global_frag(qp + 64 * sl, ld, 16 * qf, N, kk, lane);
        f32x4 s[MAXF];
#pragma unroll
        for (int kf = 0; kf < MAXF; kf++) {
            s[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kf < nf) {
#pragma unroll
                for (int sl = 0; sl < S; sl++)
#pragma unroll
                    for (int kk = 0; kk < 2; kk++)
                        s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(kimg + sl * img, 16 * kf, lo.k[kk]), qfr[sl][kk], s[kf], 0, 0, 0);
            }
        }
        mfma_fence();
        float mx = -INFINITY;
#pragma unroll
        for (int kf = 0; kf < MAXF; kf++)
            if (kf < nf) {
                if (16 * kf + 16 > N) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (16 * kf + 4 * (lane >> 4) + r >= N) s[kf][r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 4; r++) mx = fmaxf(mx, s[kf][r]);
            }
        mx = group_max(mx);
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
        f32x4 acc[S][4];
#pragma unroll
        for (int sl = 0; sl < S; sl++)
#pragma unroll
            for (int hf = 0; hf < 4; hf++) acc[sl][hf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < MAXC; c++)
            if (c < nc) {
                const bf16x8 pf = pack_frag(s[2 * c], s[2 * c + 1]);
#pragma unroll
                for (int sl = 0; sl < S; sl++)
#pragma unroll
                    for (int hf = 0; hf < 4; hf++)
                        acc[sl][hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(vimg + sl * img, 32 * c, lo.tv[hf]), pf, acc[sl][hf], 0, 0, 0);
            }
        mfma_fence();
#pragma unroll
        for (int sl = 0; sl < S; sl++) store_rows(out + 64 * sl, D, q, q < N, acc[sl], 1.f / l, lane);
        if (q < N && lane < 16) a.lse[(size_t(b) * a.H + h) * N + q] = mx * a.scale + __logf(l);
    }
}

// dK, dV: waves own 16-key fragments; Q and dO (all slices) of the whole sequence are in LDS.
template <int S>
__global__ __launch_bounds__(THREADS) void attn_wide_bwd_kv_kernel(const sfcvit_attn_args a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int img = npad * 128;
    char *qimg = smem, *doimg = smem + S * img;
    float *lse_s = reinterpret_cast<float *>(smem + 2 * S * img), *del_s = lse_s + npad;
    uint32_t *rkey_s = reinterpret_cast<uint32_t *>(del_s + npad);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x, N = a.N, hd = 64 * S, D = a.H * hd, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * hd;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    const uint16_t *dop = static_cast<const uint16_t *>(a.dout) + size_t(b) * N * D + h * hd;
    const float *lse = a.lse + (size_t(b) * a.H + h) * N, *del = a.delta + (size_t(b) * a.H + h) * N;
#pragma unroll
    for (int sl = 0; sl < S; sl++) {
        dma_slice<false>(qimg + sl * img, qp + 64 * sl, ld, N, npad, tid);
        dma_slice<false>(doimg + sl * img, dop + 64 * sl, D, N, npad, tid);
    }
    for (int i = tid; i < npad; i += THREADS) {
        lse_s[i] = i < N ? lse[i] * 1.4426950408889634f : INFINITY;
        del_s[i] = i < N ? del[i] : 0.f;
        rkey_s[i] = drop_row_key(eff_seed(a.dropout_seed, a.seed_off), (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(i));
    }
    __syncthreads();
    const int nf = (N + 15) >> 4, nc = npad >> 5;          // npad % 32 == 0 here
    const float scale = a.scale, c2 = a.scale * 1.4426950408889634f;
    const LaneOff lo = lane_offsets(lane);
    const bool drop = a.dropout_p > 0.f;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);
    uint16_t *dbase = static_cast<uint16_t *>(a.dqkv) + size_t(b) * N * ld + h * hd;
    for (int kfi = wave; kfi < nf; kfi += WAVES) {
        const int key = 16 * kfi + (lane & 15);
        bf16x8 kf[S][2], vf[S][2];
#pragma unroll
        for (int sl = 0; sl < S; sl++)
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                kf[sl][kk] = global_frag(kp + 64 * sl, ld, 16 * kfi, N, kk, lane);
                vf[sl][kk] = global_frag(vp + 64 * sl, ld, 16 * kfi, N, kk, lane);
            }
        f32x4 dk[S][4], dv[S][4];
#pragma unroll
        for (int sl = 0; sl < S; sl++)
#pragma unroll
            for (int hf = 0; hf < 4; hf++) dk[sl][hf] = dv[sl][hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nc; c++) {
            f32x4 p[2], ds[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int qf = 2 * c + t;
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < S; sl++)
#pragma unroll
                    for (int kk = 0; kk < 2; kk++) {
                        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(qimg + sl * img, 16 * qf, lo.k[kk]), kf[sl][kk], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(doimg + sl * img, 16 * qf, lo.k[kk]), vf[sl][kk], dp, 0, 0, 0);
                    }
                const int ql0 = 16 * qf + 4 * (lane >> 4);
                const f32x4 lse4 = *reinterpret_cast<const f32x4 *>(lse_s + ql0);
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
            for (int sl = 0; sl < S; sl++)
#pragma unroll
                for (int hf = 0; hf < 4; hf++) {
                    dv[sl][hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(doimg + sl * img, 32 * c, lo.t[hf]), pf, dv[sl][hf], 0, 0, 0);
                    dk[sl][hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(qimg + sl * img, 32 * c, lo.t[hf]), dsf, dk[sl][hf], 0, 0, 0);
                }
        }
        mfma_fence();
#pragma unroll
        for (int sl = 0; sl < S; sl++) {
            store_rows(dbase + D + 64 * sl, ld, key, key < N, dk[sl], 1.f, lane);
            store_rows(dbase + 2 * D + 64 * sl, ld, key, key < N, dv[sl], 1.f, lane);
        }
    }
}

// dQ: waves own 16-query fragments; K and V (all slices) of the whole sequence are in LDS.
template <int S>
__global__ __launch_bounds__(THREADS) void attn_wide_bwd_q_kernel(const sfcvit_attn_args a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int img = npad * 128;
    char *kimg = smem, *vimg = smem + S * img;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x, N = a.N, hd = 64 * S, D = a.H * hd, ld = 3 * D;
    const uint16_t *base = static_cast<const uint16_t *>(a.qkv) + size_t(b) * N * ld + h * hd;
    const uint16_t *qp = base, *kp = base + D, *vp = base + 2 * D;
    const uint16_t *dop = static_cast<const uint16_t *>(a.dout) + size_t(b) * N * D + h * hd;
#pragma unroll
    for (int sl = 0; sl < S; sl++) {
        dma_slice<false>(kimg + sl * img, kp + 64 * sl, ld, N, npad, tid);
        dma_slice<false>(vimg + sl * img, vp + 64 * sl, ld, N, npad, tid);
    }
    __syncthreads();
    const int nf = (N + 15) >> 4, nc = npad >> 5;          // npad % 32 == 0 here
    const float scale = a.scale, c2 = a.scale * 1.4426950408889634f;
    const LaneOff lo = lane_offsets(lane);
    const bool drop = a.dropout_p > 0.f;
    const uint32_t dth = drop_thresh(a.dropout_p);
    const float dsc = 1.f / (1.f - a.dropout_p);
    uint16_t *dbase = static_cast<uint16_t *>(a.dqkv) + size_t(b) * N * ld + h * hd;
    for (int qf = wave; qf < nf; qf += WAVES) {
        const int q = 16 * qf + (lane & 15);
        const float lse_q = q < N ? a.lse[(size_t(b) * a.H + h) * N + q] * 1.4426950408889634f : 0.f;
        const float del_q = q < N ? a.delta[(size_t(b) * a.H + h) * N + q] : 0.f;
        const uint32_t drk = drop_row_key(eff_seed(a.dropout_seed, a.seed_off), (uint64_t(b) * a.H + h) * uint64_t(N) + uint64_t(q));
        bf16x8 qfr[S][2], dof[S][2];
#pragma unroll
        for (int sl = 0; sl < S; sl++)
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                qfr[sl][kk] = global_frag(qp + 64 * sl, ld, 16 * qf, N, kk, lane);
                dof[sl][kk] = global_frag(dop + 64 * sl, D, 16 * qf, N, kk, lane);
            }
        f32x4 dq[S][4];
#pragma unroll
        for (int sl = 0; sl < S; sl++)
#pragma unroll
            for (int hf = 0; hf < 4; hf++) dq[sl][hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nc; c++) {
            f32x4 ds[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int kfi = 2 * c + t;
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < S; sl++)
#pragma unroll
                    for (int kk = 0; kk < 2; kk++) {
                        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(kimg + sl * img, 16 * kfi, lo.k[kk]), qfr[sl][kk], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc_frag_at(vimg + sl * img, 16 * kfi, lo.k[kk]), dof[sl][kk], dp, 0, 0, 0);
                    }
                float keep[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop) drop_keep4(drk, 16 * kfi + 4 * (lane >> 4), dth, dsc, keep);
#pragma unroll
                for (int r = 0; r < 4; r++) ds[t][r] = fast_exp2(s[r] * c2 - lse_q) * (dp[r] * keep[r] - del_q) * scale;
                if (16 * kfi + 16 > N) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (16 * kfi + 4 * (lane >> 4) + r >= N) ds[t][r] = 0.f;
                }
            }
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int sl = 0; sl < S; sl++)
#pragma unroll
                for (int hf = 0; hf < 4; hf++)
                    dq[sl][hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_at(kimg + sl * img, 32 * c, lo.t[hf]), dsf, dq[sl][hf], 0, 0, 0);
        }
        mfma_fence();
#pragma unroll
        for (int sl = 0; sl < S; sl++) store_rows(dbase + 64 * sl, ld, q, q < N, dq[sl], 1.f, lane);
    }
}

constexpr int LDS_LIMIT = 160 * 1024;

template <int S>
int launch_fwd(const sfcvit_attn_args &a, int npad, size_t lds, hipStream_t s) {
    if (int rc = raise_lds_limit(reinterpret_cast<const void *>(&attn_wide_fwd_kernel<S>), LDS_LIMIT, "attention_wide attribute")) return rc;
    note_attn_kernel("attn_wide_fwd_kernel<%d>", S);
    hipLaunchKernelGGL(attn_wide_fwd_kernel<S>, dim3(a.H, a.B), dim3(THREADS), lds, s, a, npad);
    return check_launch("attention_wide_fwd");
}

template <int S>
int launch_bwd(const sfcvit_attn_args &a, int npad, size_t lds_kv, size_t lds_q, hipStream_t s) {
    for (const void *k : {reinterpret_cast<const void *>(&attn_wide_bwd_kv_kernel<S>), reinterpret_cast<const void *>(&attn_wide_bwd_q_kernel<S>)})
        if (int rc = raise_lds_limit(k, LDS_LIMIT, "attention_wide attribute")) return rc;
    note_attn_kernel("attn_wide_bwd_kv_kernel<%d>", S);
    hipLaunchKernelGGL(attn_wide_bwd_kv_kernel<S>, dim3(a.H, a.B), dim3(THREADS), lds_kv, s, a, npad);
    if (int rc = check_launch("attention_wide_bwd kv")) return rc;
    hipLaunchKernelGGL(attn_wide_bwd_q_kernel<S>, dim3(a.H, a.B), dim3(THREADS), lds_q, s, a, npad);
    return check_launch("attention_wide_bwd q");
}

}  // namespace

// Head dims 128 / 192 / 256.  -1: not this path's business (hd == 64); otherwise a status (EINVAL with a message when
// the sequence does not fit the LDS).
int attn_wide_fwd(const sfcvit_attn_args &a, hipStream_t s) {
    if (a.hd == 64) return -1;
    const int S = a.hd / 64;
    if (a.hd % 64 || S < 2 || S > 4) return fail(SFCVIT_EINVAL, "attention: head dim %d not supported (64, 128, 192, 256)", a.hd);
    const int npad = (a.N + 31) / 32 * 32;
    const size_t lds = size_t(2) * S * npad * 128;
    if (a.N > 256 || lds > size_t(LDS_LIMIT))
        return fail(SFCVIT_EINVAL, "attention: head dim %d with N = %d needs %zu KiB of LDS (limit 160); only head dim 64 has a tiled kernel", a.hd, a.N, lds >> 10);
    if (S == 2) return launch_fwd<2>(a, npad, lds, s);
    if (S == 3) return launch_fwd<3>(a, npad, lds, s);
    return launch_fwd<4>(a, npad, lds, s);
}

int attn_wide_bwd(const sfcvit_attn_args &a, hipStream_t s) {
    if (a.hd == 64) return -1;
    const int S = a.hd / 64;
    if (a.hd % 64 || S < 2 || S > 4) return fail(SFCVIT_EINVAL, "attention: head dim %d not supported (64, 128, 192, 256)", a.hd);
    const int npad = (a.N + 31) / 32 * 32;
    const size_t lds_q = size_t(2) * S * npad * 128, lds_kv = lds_q + size_t(3) * npad * 4;
    if (a.N > 256 || lds_kv > size_t(LDS_LIMIT))
        return fail(SFCVIT_EINVAL, "attention: head dim %d with N = %d needs %zu KiB of LDS (limit 160); only head dim 64 has a tiled kernel", a.hd, a.N, lds_kv >> 10);
    if (S == 2) return launch_bwd<2>(a, npad, lds_kv, lds_q, s);
    if (S == 3) return launch_bwd<3>(a, npad, lds_kv, lds_q, s);
    return launch_bwd<4>(a, npad, lds_kv, lds_q, s);
}

}  // namespace sfcvit
