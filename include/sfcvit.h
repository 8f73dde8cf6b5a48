/*
 * sfcvit.h -- C ABI of the MI355X (gfx950) SFC-ViT hot path.
 *
 * The reference (RemcoHoger/Space-Filling-Curves-for-Vision-Transformers) is pure
 * Python on PyTorch and has no FFI of its own: these entry points
 * are what a binding for the path would bind.  Each one names the reference
 * interface it replaces (paths relative to the reference root; `torch:` = the
 * installed torch the reference calls into).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  Unless marked HOST every
 *     pointer is a device (HBM) pointer and must be 16-byte aligned.
 *   - bf16 tensors are passed as `const void*` to raw bf16 bits, row-major.
 *   - `stream` is a hipStream_t (NULL = default stream).  No entry point
 *     synchronises, allocates or copies to the host: all are graph-capturable.
 *   - return value: 0 = launched / done, nonzero = rejected (see sfcvit_last_error()).
 *     Nothing is launched when an argument check fails.
 *   - inputs are never written; outputs are fully overwritten.
 */
#ifndef SFCVIT_H
#define SFCVIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFCVIT_ABI_VERSION 1

enum sfcvit_status {
    SFCVIT_OK = 0,
    SFCVIT_EINVAL = 1,   /* bad argument (shape, alignment, enum) */
    SFCVIT_ELAUNCH = 2,  /* HIP reported an error at launch */
    SFCVIT_ENODEV = 3    /* no usable gfx950 device */
};

int sfcvit_abi_version(void);
/* HOST: message of the last failing call on this thread ("" if none). */
const char *sfcvit_last_error(void);
/* HOST: number of visible HIP devices, <0 on error (does not create a context). */
int sfcvit_device_count(void);

/* ------------------------------------------------------------------------
 * Curve index tables (host, integer, bit-exact with the reference)
 * ---------------------------------------------------------------------- */
enum sfcvit_curve {
    SFCVIT_CURVE_HILBERT = 0, /* src/curves/space_filling_curves.py:168-202 */
    SFCVIT_CURVE_Z = 1,       /* :134-165 */
    SFCVIT_CURVE_MOORE = 2,   /* :205-251 */
    SFCVIT_CURVE_PEANO = 3,   /* :74-131  */
    SFCVIT_CURVE_RASTER = 4,  /* identity order (RasterScan1DEmbedding, zigzag_embedding1D.py:30-39) */
    SFCVIT_CURVE_SPIRAL = 5,  /* inward spiral from the bottom-left cell: OnionEmbedding1D.onion_indices,
                                 src/tokenizers/_1D/onion_embedding1D.py:35-53 = multiscale/multi_onion.py:68-87 */
    SFCVIT_CURVE_HILBERT_T = 6 /* the private generator of _2D/HilbertEmbedding (src/tokenizers/_2D/
                                 hilbert_embedding.py:30-78): Hilbert without the (x,y) swap; n a power of two */
};

/* HOST. embed_and_prune_sfc(curve, n, n) (space_filling_curves.py:471-491) as the
 * flat table r*n+c that SFCEmbedding1D._sfc_indices builds (multi_hilbert.py:68-72).
 * out_flat: n*n int32. */
int sfcvit_curve_table(int curve, int n, int32_t *out_flat);
/* HOST. Same points as (row, col) int64 pairs: the `hilbert_indices` / `z_indices`
 * buffers of HilbertEmbedding1D / MortonEmbedding1D (hilbert_embedding1D.py:20-28).
 * out_rc: n*n*2 int64. */
int sfcvit_curve_table_rc(int curve, int n, int64_t *out_rc);

/* HOST. Per-token pixel offsets for the fused tokenizer kernel.
 * Token t of SFCEmbedding1D(img, p, g) (multi_hilbert.py:74-84) is made of
 * P = g*p*p pixels; pixel kk = gi*p*p + p1*p + p2 is at
 * (row, col) = ((flat[t*g+gi] / grid)*p + p1, (flat[t*g+gi] % grid)*p + p2), grid = img/p.
 * The 1-D tokenizers are the case p = 1, g = patch_size with `flat` = r*img+c.
 * flat: grid*grid int32 (curve table); out_pix: (grid*grid/g) * P int32 = img*img entries. */
int sfcvit_pixel_table(const int32_t *flat, int img, int p, int g, int32_t *out_pix);

/* ------------------------------------------------------------------------
 * Fused SFC gather + patchify + linear projection
 *   replaces HilbertEmbedding1D.forward / MortonEmbedding1D.forward
 *   (src/tokenizers/_1D/hilbert_embedding1D.py:30-44), RasterScan1DEmbedding.forward
 *   (zigzag_embedding1D.py:30-39) and SFCEmbedding1D.forward (multi_hilbert.py:74-84).
 * ---------------------------------------------------------------------- */
typedef struct sfcvit_patch_embed_args {
    const void *x;      /* [B, C, H*W] image, fp32 (x_is_bf16 = 0) or bf16 */
    const int32_t *pix; /* [N, P] pixel offsets (sfcvit_pixel_table), device copy */
    const void *w;      /* [D, P*C] bf16, feature index = kk*C + c (reference layout) */
    const void *bias;   /* [D] bf16 or NULL */
    void *y;            /* fwd: out [B*N, D] bf16 ; bwd: in, dY [B*N, D] bf16 */
    void *dw;           /* bwd: out [D, P*C] fp32 */
    void *dbias;        /* bwd: out [D] fp32 (NULL = skip) */
    void *workspace;    /* sfcvit_patch_embed_workspace(...) bytes, 16-byte aligned */
    int64_t workspace_bytes;
    int32_t B, C, HW, N, P, D; /* N*P = H*W; D a multiple of 8; P*C arbitrary (padded to 8 inside; the
                                  vectorised table reads need P % 8 == 0, other P take a scalar gather) */
    int32_t x_is_bf16;
    /* Optional tile descriptor (device copy of what sfcvit_tile_descriptors wrote; NULL = generic kernels).  With it,
     * P = 256 and D % 256 == 0 the forward runs the tiled kernel (csrc/patch_embed_tiled.hip: whole 16-pixel row
     * segments loaded with 16-byte vectors straight into the LDS tile, the intra-tile curve order folded into a
     * per-class permuted weight).  desc_ncls / desc_cnt = its classes and tokens per class (host copies);
     * the forward workspace must then hold desc_ncls * D * C * 256 bf16. */
    const int32_t *desc;
    int32_t desc_ncls;
    int32_t desc_cnt[8];
} sfcvit_patch_embed_args;

/* HOST: analyse a pixel table (host copy of sfcvit_pixel_table's output): > 0 = number of int32 written to `desc`
 * (every token is a 16 x 16 pixel tile or a strip of 256 consecutive pixels), 0 = not tileable, < 0 = error.
 * Capacity 16 + 2 N + 2 * 8 * 256 always suffices. */
int sfcvit_tile_descriptors(const int32_t *pix, int N, int P, int img_w, int32_t *desc, int capacity);

/* HOST: workspace bytes for fwd (bwd = 0) / bwd (bwd = 1). */
int64_t sfcvit_patch_embed_workspace(int B, int C, int N, int P, int D, int bwd);

int sfcvit_patch_embed_fwd(const sfcvit_patch_embed_args *a, void *stream);
/* The gather alone: tokens[b * N + n][kk * C + c] = bf16(x[b, c, pix[n][kk]]) -- the reference's
 * `x_flat[:, :, perm].reshape(B, N, P * C)` (src/tokenizers/_1D/hilbert_embedding1D.py:36-41) as one pass: the image is read
 * once (every 16 x 16 tile of a Hilbert / Z token by one workgroup), rows are written whole.  tokens is [B * N, ld] bf16 with
 * ld >= P * C, ld % 8 == 0 (columns P * C .. ld - 1 are zeroed).  order (device, [N], or NULL) = the tokens sorted by their
 * lowest pixel offset: workgroups then take horizontally adjacent 16 x 16 tiles in pairs, whose 64-byte rows share
 * 128-byte lines (a performance hint only; any permutation of 0 .. N - 1 gives the same tokens).  Two-stage patch embed = this + sfcvit_gemm (round 3: at
 * ViT-B the persistent GEMMs run the projection and its weight gradient 2-3x faster than the fused gather kernels). */
int sfcvit_tokens_gather(const void *x, int x_is_bf16, const int32_t *pix, const int32_t *order, int B, int C, int HW, int N, int P,
                         void *tokens, int ld, void *stream);
/* The same tokens for pixel tables whose tokens are 16 x 16 pixel tiles of an fp32 image (sfcvit_tile_descriptors mode 1:
 * every Hilbert / Z tokenizer at 256 pixels per token): whole 128-byte image lines by 16-byte loads, the curve order applied on
 * the way out of LDS (round 4; csrc/patch_embed.hip tokens_gather_tiles_kernel).  origin (device, [N]) = flat offset of each
 * token's top-left pixel (descriptor words [16 + N, 16 + 2 N)); order as above; ld must be 256 * C, 1 <= C <= 4, x fp32. */
int sfcvit_tokens_gather_tiles(const void *x, const int32_t *pix, const int32_t *order, const int32_t *origin, int B, int C, int H,
                               int W, int N, void *tokens, int ld, void *stream);
/* dW = sum_{b,t} dY[b,t,:]^T tokens[b,t,:] with the tokens re-gathered from x
 * (nothing but x is saved for backward); the image receives no gradient. */
int sfcvit_patch_embed_bwd(const sfcvit_patch_embed_args *a, void *stream);

/* ------------------------------------------------------------------------
 * Fused hierarchical tokenizer, forward
 *   replaces HierarchicalHilbertEmbedding.forward (src/tokenizers/multiscale/multi_hilbert.py:31-40) and its
 *   siblings multi_morton.py / multi_moore.py / multi_peano.py / multi_onion.py / multi_zigzag.py (same lines) when
 *   every level has the same token count N (true of every configuration the reference ships, main.py:269-274; the
 *   F.interpolate(mode='linear') to N tokens is then the identity):
 *       h[m, l*D:(l+1)*D] = bf16( W_l tokens_l[m, :] + b_l )      level l = SFCEmbedding1D (multi_hilbert.py:74-84)
 *       y[m, :]           = bf16( Wf h[m, :] + bf )               fusion  = nn.Linear(L*D, L*D)
 *   in ONE kernel: tokens gathered through the per-level pixel tables, level outputs kept in LDS as the A operand
 *   of the fusion GEMM.  h is also written out (the fusion weight gradient needs it); backward composes
 *   sfcvit_gemm (dh, dWf) and sfcvit_patch_embed_bwd per level.
 *   wf = NULL: the kernel stops after the first line -- gather + every level projection + the concatenation (each
 *   level writes its own columns of h; no torch.cat pass) -- and the caller runs the fusion Linear as sfcvit_gemm.
 *   That pair is the faster one at the reference's shape (the all-in-one kernel re-streams Wf once per
 *   64 token rows and is bound by L2 -> CU delivery); both are kept, bit-identical in h.
 * ---------------------------------------------------------------------- */
#define SFCVIT_HIER_MAX_LEVELS 4
typedef struct sfcvit_hier_args {
    const void *x;                               /* [B, C, H*W] image, fp32 (x_is_bf16 = 0) or bf16 */
    const int32_t *pix[SFCVIT_HIER_MAX_LEVELS];  /* level l: [N, P[l]] pixel offsets (sfcvit_pixel_table), device */
    const void *w[SFCVIT_HIER_MAX_LEVELS];       /* level l: [D, P[l]*C] bf16, feature index kk*C + c, 16-byte aligned */
    const void *b[SFCVIT_HIER_MAX_LEVELS];       /* level l: [D] bf16 or NULL */
    const void *wf;                              /* [L*D, L*D] bf16; NULL = levels + concatenation only (y unused) */
    const void *bf;                              /* [L*D] bf16 or NULL */
    void *h;                                     /* out [B*N, L*D] bf16: concatenated level outputs */
    void *y;                                     /* out [B*N, L*D] bf16 */
    int32_t P[SFCVIT_HIER_MAX_LEVELS];           /* pixels per token of level l; N * P[l] = H*W for every level */
    int32_t B, C, HW, N, L, D;
    int32_t x_is_bf16;
} sfcvit_hier_args;

/* HOST: 1 if (L, D, C, P[0..L)) is inside the fused kernel's envelope: 1..4 levels, D % 64 == 0, L*D % 256 == 0,
 * P[l]*C % 8 == 0, and 64 rows of (L*D + sum of P[l]*C) bf16 fit the 160 KiB LDS; else 0 (compose the level calls). */
int sfcvit_hier_tokenizer_supported(int L, int D, int C, const int32_t *P);
/* SFCVIT_EINVAL outside that envelope or when the levels do not share the token count. */
int sfcvit_hier_tokenizer_fwd(const sfcvit_hier_args *a, void *stream);
/* Levels with different token counts: the reference resamples every coarser level to the first level's length with
 * F.interpolate(mode="linear", align_corners=False) and concatenates on the feature axis
 * (src/tokenizers/multiscale/multi_hilbert.py:33-38).  Both in one pass over bf16 level outputs:
 *   levels[l] (device) = y_l [B, n_tokens[l], D];  out = [B, N0, L * D], N0 = n_tokens[0];  `levels` / `n_tokens` are HOST arrays.
 * _bwd: dlevels[l] [B, n_tokens[l], D] = the transposed resampling of dout's column block l (fixed summation order). */
int sfcvit_hier_resample_concat(const void *const *levels, const int32_t *n_tokens, int L, int B, int N0, int D, void *out, void *stream);
int sfcvit_hier_resample_concat_bwd(const void *dout, const int32_t *n_tokens, int L, int B, int N0, int D, void *const *dlevels,
                                    void *stream);

/* ------------------------------------------------------------------------
 * bf16 MFMA GEMM with fused epilogue
 *   replaces nn.Linear forward/backward at every site of the path:
 *   in_proj / out_proj (torch:nn/functional.py:5822-5833,6632-6637), linear1/linear2
 *   (torch:nn/modules/transformer.py:980-982), MixerBlock.channel_mix (src/models/vit.py:262-266),
 *   FactorisedLinear's two einsums (vit.py:289-292) and the classifier (vit.py:319).
 *
 *   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )
 *   a_kmajor = 0: A is [M, lda] with k contiguous;  1: A is [K, lda] with m contiguous
 *   b_kmajor = 0: B is [N, ldb] with k contiguous;  1: B is [K, ldb] with n contiguous
 *   epilogue, in this order (fp32):  v += bias[n];  aux_out[m,n] = bf16(v);
 *   v = act(v);  v = dropout(v);  v += residual[m,n];  v *= dact(aux_in[m,n]) * dact_scale;  C[m,n] = v
 * ---------------------------------------------------------------------- */
enum sfcvit_act { SFCVIT_ACT_NONE = 0, SFCVIT_ACT_RELU = 1, SFCVIT_ACT_GELU = 2 };
/* dact: 0 none; RELU: (aux_in > 0); GELU: gelu'(aux_in) (erf form, nn.GELU default) */

typedef struct sfcvit_gemm_args {
    const void *a, *b;
    void *c;
    const void *bias;      /* [N] bf16 or NULL */
    const void *residual;  /* [M, ldr] bf16 or NULL */
    const void *aux_in;    /* [M, ldaux] bf16, needed when dact != 0 */
    void *aux_out;         /* [M, ldaux] bf16 or NULL */
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldr, ldaux;
    int32_t a_kmajor, b_kmajor;
    int32_t act, dact;
    int32_t c_is_f32;      /* 0: C is bf16, 1: C is fp32 */
    int32_t splitk;        /* >1: split K over that many workgroups per tile (weight-gradient
                              shapes); needs `workspace`, allows no epilogue */
    void *workspace;       /* fp32 slabs, sfcvit_gemm_workspace(M, N, splitk) bytes */
    int64_t workspace_bytes;
    int32_t force_generic; /* kernel choice, for tests and benchmarks: 0 = automatic (persistent 8-phase kernel when
                              both operands are k-contiguous and the shape sits on its tile grid, else the LDS-DMA
                              ring kernel, else the generic 128 x 128 kernel); 1 = generic only; 4 / 6 / 7 = ring
                              kernel (heuristic / 256 x 128 / 256 x 256); 8 / 9 / 10 = persistent kernel with
                              256- / 224- / 192-row tiles (EINVAL when not eligible) */
    float dropout_p;       /* > 0: after act, before residual: v = keep(m, n) ? v / (1 - p) : 0 (nn.Dropout, training) */
    uint32_t dropout_seed;
    float dact_scale;      /* multiplies v together with dact (0 = 1): 1/(1-p) of a dropout that followed the ReLU */
    int32_t row_offset;    /* dropout mask row of C row m is m + row_offset (a GEMM computed as row slices
                              keeps one mask) */
    void *colsum_out;      /* NULL, or [N]: column sums over m of the epilogue's result (the bias gradient of the Linear
                              that produced the A operand's gradient chain, e.g. db1 = colsum(dh)): fp32, or bf16 when
                              colsum_bf16 != 0.  Fused into the persistent kernel's epilogue when that kernel runs
                              (partials of the fp32 values in `workspace`, then a fixed-order reduce); otherwise a
                              separate pass over the stored bf16 C.  Needs workspace_bytes >=
                              sfcvit_gemm_colsum_workspace(M, N); not with split-K or fp32 C. */
    int32_t colsum_bf16;
    const uint32_t *seed_off; /* NULL, or a device word added to dropout_seed when the kernel RUNS (sfcvit_step_advance):
                                 lets a captured graph draw a new mask per replay */
    void *actmask;         /* NULL, or a bit matrix [M][ld_actmask bytes], bit (n & 7) of byte n >> 3 of row m <-> C[m, n] > 0:
                              with act = RELU it is WRITTEN (fused into the persistent kernel's epilogue, a separate pass
                              over C otherwise); with dact = RELU it MAY be read in place of aux_in (the persistent kernel
                              does: 2 bytes instead of 32 per lane and row; aux_in must be valid all the same).
                              linear2's dX through ReLU + dropout needs only the sign pattern of the stored activation
                              (torch:nn/modules/transformer.py:980-982): 19 MB instead of 308 MB per ViT-B layer.
                              N % 16 == 0, ld_actmask even and >= N / 8, 2-byte aligned. */
    int32_t ld_actmask;
} sfcvit_gemm_args;

int sfcvit_gemm(const sfcvit_gemm_args *a, void *stream);
/* HOST: workspace bytes for a call with colsum_out. */
int64_t sfcvit_gemm_colsum_workspace(int M, int N);
/* HOST: name of the kernel the calling thread's last sfcvit_gemm launched, as rocprofv3 prints it (without the
 * namespace), e.g. "gemm8p_kernel<7, 6, true>" -- lets a benchmark key its live timings by kernel symbol. */
int sfcvit_last_gemm_kernel(char *buf, int n);
/* HOST: bytes of workspace sfcvit_gemm needs for this split (0 when splitk <= 1). */
int64_t sfcvit_gemm_workspace(int M, int N, int splitk);

/* dst[c, r] = src[r, c] (bf16; R, C, lds, ldd multiples of 8).  One transposed copy of a weight per
 * step lets dX = dY W run with both operands k-contiguous (sfcvit_gemm's fastest layout). */
int sfcvit_transpose(const void *src, int R, int C, int lds, void *dst, int ldd, void *stream);

/* The same for many contiguous matrices in ONE launch (every weight of a model at the start of its backward pass: 51 launches
 * of 5 us become one).  `tiles` is a DEVICE array with one entry per 64 x 64 tile; matrix i occupies src_base + src_off
 * elements as [R, C] and its transpose dst_base + dst_off as [C, R]; R and C multiples of 8, offsets multiples of 8. */
typedef struct sfcvit_transpose_tile {
    int64_t src_off, dst_off;   /* in bf16 elements */
    int32_t R, C;               /* the matrix this tile belongs to */
    int32_t r0, c0;             /* first row / column of the tile */
} sfcvit_transpose_tile;
int sfcvit_transpose_batched(const void *src_base, void *dst_base, const sfcvit_transpose_tile *tiles, int n_tiles, void *stream);

/* Deferred partial-sum reductions.  Every column-sum-like result of this library (sfcvit_colsum, the bias sums of
 * sfcvit_gemm / sfcvit_attention_bwd, dgamma / dbeta / column sums of sfcvit_layernorm_bwd) is a main kernel that writes
 * fp32 partial rows into the caller's workspace and a small fixed-order reduction over them.  While deferral is on
 * (process-wide switch), calls queue that reduction instead of launching it; sfcvit_reduce_flush launches everything
 * queued as ONE kernel on `stream` (which must be the stream the calls used).  The caller keeps the workspaces and the
 * outputs alive and unread until the flush.  sfcvit_reduce_defer returns the previous setting; _pending the queue length;
 * _discard empties the queue without launching (after an aborted pass). */
int sfcvit_reduce_defer(int on);
int sfcvit_reduce_pending(void);
int sfcvit_reduce_flush(void *stream);
int sfcvit_reduce_discard(void);

/* Column sums: out[n] = sum_m x[m, n] (bias gradients). x bf16 [M, ld]; out fp32 [N] (overwritten).
 * Two passes through `workspace` (sfcvit_colsum_workspace bytes, HOST query) instead of float atomics, so the
 * result is bit-reproducible from run to run. */
int64_t sfcvit_colsum_workspace(int M, int N);
int sfcvit_colsum(const void *x, int M, int N, int ld, void *out, int out_bf16, void *workspace, int64_t workspace_bytes,
                  void *stream);   /* out: fp32 [N], or bf16 [N] when out_bf16 != 0 */

/* ------------------------------------------------------------------------
 * LayerNorm (biased variance, affine) -- nn.LayerNorm at norm1/norm2
 * (torch:nn/modules/transformer.py:951-958), channel_mix_ln (vit.py:254,272), mlp_head.0 (vit.py:303)
 * ---------------------------------------------------------------------- */
/* y = (x - mean) * rstd * gamma + beta ; mean, rstd fp32 [M] saved for backward. */
int sfcvit_layernorm_fwd(const void *x, const void *gamma, const void *beta, void *y,
                         float *mean, float *rstd, int M, int D, float eps, void *stream);
/* dx (bf16) ; dgamma, dbeta fp32 [D].  If dx_add != NULL, dx = dx_add + LN-backward
 * (gradient arriving over the residual branch).  ws: fp32 workspace of
 * sfcvit_layernorm_bwd_ws(M, D) bytes. */
int sfcvit_layernorm_bwd(const void *dy, const void *x, const float *mean, const float *rstd,
                         const void *gamma, const void *dx_add, void *dx, float *dgamma,
                         float *dbeta, int M, int D, void *ws, void *stream);
/* Same, plus (dx_drop != NULL) dx_drop[m, d] = keep(m, d) ? dx / (1 - p) : 0 (bf16): the gradient
 * entering a sub-layer whose output went through nn.Dropout(p) before the residual add (dropout1 /
 * dropout2, torch:nn/modules/transformer.py:953-957), mask regenerated from `seed`; and
 * (dcol != NULL) dcol[d] = sum_m of that outgoing gradient (dx_drop if given, else dx), fp32 [D]: the
 * bias gradient of the sub-layer's last Linear, for free in the same pass. */
int sfcvit_layernorm_bwd_drop(const void *dy, const void *x, const float *mean, const float *rstd,
                              const void *gamma, const void *dx_add, void *dx, void *dx_drop, float p,
                              uint32_t seed, const uint32_t *seed_off, void *dgamma, void *dbeta, void *dcol, int grads_bf16, int M, int D,
                              void *ws, void *stream);
/* grads_bf16 != 0: dgamma / dbeta / dcol are bf16 [D] instead of fp32 -- the caller passes views of its flat
 * gradient buffer and no cast / accumulate pass follows. */
int64_t sfcvit_layernorm_bwd_ws(int M, int D);
/* HOST: name of the main kernel the calling thread's last sfcvit_layernorm_bwd / _bwd_drop launched, as rocprofv3 prints
 * it (e.g. "ln_bwd_cols_kernel<4, true>"): tests assert through it that a width ran on the column-sum kernel. */
int sfcvit_last_rowwise_kernel(char *buf, int n);

/* ------------------------------------------------------------------------
 * Multi-head self-attention core (no mask) on the packed projection
 *   replaces F.scaled_dot_product_attention as reached from nn.MultiheadAttention
 *   (torch:nn/functional.py:6623-6631); qkv is the in_proj output [B, N, 3*H*hd]
 *   with q | k | v in thirds, head h at columns h*hd .. h*hd+hd-1 of each third.
 * ---------------------------------------------------------------------- */
typedef struct sfcvit_attn_args {
    const void *qkv; /* [B, N, 3*H*hd] bf16 */
    void *out;       /* fwd: out [B, N, H*hd] bf16 ; bwd: in */
    float *lse;      /* [B, H, N] fp32 log-sum-exp of the scaled scores: fwd out, bwd in */
    const void *dout; /* bwd: [B, N, H*hd] bf16 */
    void *dqkv;       /* bwd: out [B, N, 3*H*hd] bf16 */
    float *delta;     /* bwd: workspace [B, H, N] fp32 */
    int32_t B, N, H, hd;
    float scale;      /* 1/sqrt(hd) */
    float dropout_p;  /* > 0: dropout on the attention probabilities (SDPA dropout_p, training mode);
                         row = (b*H + h)*N + q, col = key of the mask function */
    uint32_t dropout_seed;
    const uint32_t *seed_off; /* as in sfcvit_gemm_args */
    /* backward only, optional: column sums of dqkv over all B * N rows (= the in_proj bias gradient,
     * torch:nn/functional.py:5822-5833) written to colsum_out ([3 D], fp32 or bf16).  colsum_part = workspace of
     * sfcvit_attention_colsum_workspace(B, N, H, hd) bytes.  The one-pass kernel (hd = 64, N <= 224) emits the sums
     * itself; the other paths run sfcvit_colsum over the dqkv they wrote. */
    float *colsum_part;
    int64_t colsum_part_bytes;
    void *colsum_out;
    int32_t colsum_bf16;
} sfcvit_attn_args;
int64_t sfcvit_attention_colsum_workspace(int B, int N, int H, int hd);

int sfcvit_attention_fwd(const sfcvit_attn_args *a, void *stream);
int sfcvit_attention_bwd(const sfcvit_attn_args *a, void *stream);
/* HOST: name of the main kernel the calling thread's last sfcvit_attention_fwd / _bwd launched, as rocprofv3 prints it
 * (e.g. "attn_seq_bwd_fused_kernel<13, true>"): tests assert through it that a shape ran on the production kernel. */
int sfcvit_last_attn_kernel(char *buf, int n);

/* ------------------------------------------------------------------------
 * Elementwise / loss / optimizer
 * ---------------------------------------------------------------------- */
/* y = gelu_erf(x) (nn.GELU in MultiLayerPredictor, vit.py:308); bf16, n elements. */
int sfcvit_gelu_fwd(const void *x, void *y, int64_t n, void *stream);
/* dx = dy * gelu'(x) */
int sfcvit_gelu_bwd(const void *dy, const void *x, void *dx, int64_t n, void *stream);
/* GELU followed by nn.Dropout(p) (MultiLayerPredictor, vit.py:308-309) on a [rows, cols] tensor, cols % 8 == 0. */
int sfcvit_gelu_drop_fwd(const void *x, void *y, int rows, int cols, float p, uint32_t seed, const uint32_t *seed_off, void *stream);
int sfcvit_gelu_drop_bwd(const void *dy, const void *x, void *dx, int rows, int cols, float p, uint32_t seed, const uint32_t *seed_off,
                         void *stream);
/* The keep mask itself, as bf16 {0, 1/(1-p)} (tests and debugging): out [rows, cols]. */
int sfcvit_dropout_mask(void *out, int64_t rows, int cols, float p, uint32_t seed, void *stream);

/* SoftTargetCrossEntropy (main.py:45-51), forward and gradient in one pass.
 * logits bf16 [B, ld] (first C columns used), targets fp32 [B, C];
 * loss_rows fp32 [B] = -sum_c t*log_softmax ; dlogits bf16 [B, ld] = (softmax*sum_c t - t) * gscale
 * (columns C..ld-1 are written 0).  The mean over B is gscale = 1/B by the caller. */
int sfcvit_soft_ce(const void *logits, const float *targets, float *loss_rows, void *dlogits,
                   int B, int C, int ld, float gscale, void *stream);

/* Sum of squares of a bf16 (is_f32 = 0) or fp32 buffer, accumulated into *out (fp32, device).  Block partials
 * go through `workspace` (SFCVIT_SUMSQ_WORKSPACE_BYTES, device) and are added in a fixed order: reproducible. */
#define SFCVIT_SUMSQ_WORKSPACE_BYTES 4096
int sfcvit_sumsq_accum(const void *g, int64_t n, int is_f32, float *out, void *workspace, void *stream);

/* Fused clip_grad_norm_ + AdamW step (src/training/train.py:165-166, main.py:288-289) on a
 * flat buffer.  clip coefficient = min(1, max_norm / (sqrt(*sumsq) + 1e-6)) is computed on the
 * device from *sumsq (torch.nn.utils.clip_grad_norm_ semantics).  master: fp32 copy of the
 * parameters (updated), param: bf16 parameters (rewritten from master), grad bf16,
 * m, v fp32.  Decoupled weight decay as torch.optim.AdamW. */
typedef struct sfcvit_adamw_args {
    void *param;       /* bf16 [n] */
    float *master;     /* fp32 [n] */
    const void *grad;  /* bf16 [n] */
    float *m, *v;      /* fp32 [n] */
    const float *sumsq; /* device scalar: sum of squares of ALL grads (NULL = no clipping) */
    int64_t n;
    float lr, beta1, beta2, eps, weight_decay, max_norm;
    float grad_scale;  /* every gradient (and the norm) is multiplied by this first:
                          1/world_size after a SUM all-reduce, else 1 */
    int32_t step;      /* 1-based */
    const float *dev_state; /* NULL, or the device step state of sfcvit_step_advance: lr and the bias corrections are then
                               read from it when the kernel RUNS (lr / step above are ignored) */
} sfcvit_adamw_args;
int sfcvit_adamw_step(const sfcvit_adamw_args *a, void *stream);

/* Device-resident step state, 8 words (16-byte aligned): [0] dropout seed offset (uint32), [1] step (int32), [2] learning
 * rate (float, host-written), [3] 1 - beta1^step, [4] 1 / sqrt(1 - beta2^step).  sfcvit_step_advance increments the step
 * and refreshes [0], [3], [4]; run it once per training step before the forward (first node of a captured step).  The
 * reference keeps all of this on the host (torch.optim.AdamW's step counter, torch's Philox offset); on the device a
 * whole training step replays from one hipGraph (main.py:284's torch.compile(mode="reduce-overhead") intent). */
int sfcvit_step_advance(void *state, float beta1, float beta2, uint32_t seed_base, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SFCVIT_H */
