"""Block-level autograd functions of the hot path, built on sfcvit.ops (HIP kernels).

Granularity follows the reference's modules so that every fusion the backward
needs is local to one function:
  patch_embed    HilbertEmbedding1D / MortonEmbedding1D / RasterScan1DEmbedding / SFCEmbedding1D .forward
  mixer_block    MixerBlock.forward                     (src/models/vit.py:268-273)
  encoder_layer  nn.TransformerEncoderLayer, post-norm  (torch:nn/modules/transformer.py:951-982)
  predictor_head MultiLayerPredictor(n_layers=2)        (src/models/vit.py:295-319)
  linear, layer_norm, gelu                              generic pieces
  soft_target_cross_entropy                             main.py:45-51
All activations and parameters are bf16 inside; fp32 parameters are cast on entry
(differentiably), so gradients come back in the parameter's own dtype.
"""
import math

import torch
from torch.autograd import Function

from . import ops

_BF16 = torch.bfloat16


def _bf(t):
    return t if t is None or t.dtype == _BF16 else t.to(_BF16)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _like(g, ref):
    """Gradient `g` in the dtype of the tensor it belongs to."""
    return g if g.dtype == ref.dtype else g.to(ref.dtype)


_SLOTS_OFF = False     # sfcvit.library: inside a traced custom op gradients must be fresh tensors, never views of the flat buffer


def _slot(p):
    """A fresh view of parameter `p`'s place in its optimizer's flat gradient buffer (FlatGradBuffer.slots), or None.
    Kernels write the gradient there directly and return the view: autograd, finding p.grad unset, adopts it as
    p.grad without a copy -- no fp32 -> bf16 cast and no `p.grad += g` pass per parameter (146 + 106 tiny kernels per
    ViT-B step).  Only when p.grad is None: with an existing gradient autograd would add the view to itself."""
    if _SLOTS_OFF or p is None or not p.is_leaf or p.grad is not None:
        return None
    slot = getattr(p, "_sfcvit_slot", None)
    if slot is None:
        return None
    buf, off = slot
    if getattr(p, "_sfcvit_claimed", -1) == buf.epoch:
        # second use of a shared parameter in one backward: autograd must add.  It adds (first use's slot view + this
        # use's fresh tensor) as soon as this use returns, i.e. it READS the slot now: a reduction the first use queued
        # for the end of the pass (ops._Deferring) has to land before that.
        ops.flush_deferred(end=False)
        return None
    p._sfcvit_claimed = buf.epoch
    view = buf.flat_grad[off:off + p.numel()].view(p.shape)
    view._sfcvit_deferrable = True          # ops._Deferring: a reduction that ends here may wait for the end of the backward pass
    return view


def _wgrad(dy2, x2, w=None):
    """dW[out, in] = dY^T X  (contraction over rows; both operands k-major), into w's gradient slot if it has one."""
    return ops.gemm(dy2, x2, a_kmajor=True, b_kmajor=True, out=_slot(w))


class _SideStream:
    """Weight-gradient GEMMs do not feed the rest of backward, so they run on a second HIP stream
    next to the dX chain: the tail round of one kernel (1176 tiles on 512 workgroup slots leave the
    third round 30 % full) could be filled with workgroups of the other.  MEASURED SLOWER on MI355X
    (5 110 vs 5 210 img/s, A/B in one process): the two GEMMs evict each other's panels from L2.
    Kept as an option (SFCVIT_SIDE_STREAM=1), off by default."""
    _streams = {}

    def __init__(self, device):
        self.main = torch.cuda.current_stream(device)
        key = (device.index, self.main.cuda_stream)
        if key not in _SideStream._streams:
            _SideStream._streams[key] = torch.cuda.Stream(device)
        self.side = _SideStream._streams[key]

    def run(self, fn, *tensors):
        """fn() on the side stream after everything issued so far on the main stream."""
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            out = fn()
        for t in tensors:                       # inputs allocated on the main stream, read on the side stream
            t.record_stream(self.side)
        return out

    def join(self, *outs):
        self.main.wait_stream(self.side)
        for t in outs:                          # allocated on the side stream, consumed on the main stream
            if t is not None:
                t.record_stream(self.main)


import os as _os
SIDE_STREAM_WGRAD = _os.environ.get("SFCVIT_SIDE_STREAM", "0") == "1"
USE_ACTMASK = _os.environ.get("SFCVIT_ACTMASK", "1") == "1"        # "0": linear2's dX reads the stored activation (A/B)


def _ln_slots(w, b):
    """grad_out for ops.layernorm_bwd: (dgamma slot, dbeta slot, None) when both parameters have one, else None."""
    sw, sb = _slot(w), _slot(b)
    return (sw, sb, None) if sw is not None and sb is not None else None


def _ln_grads(dg, db, go):
    return (dg, db) if go is not None else (dg.to(_BF16), db.to(_BF16))


def _bgrad(dy2, b=None):
    out = _slot(b)
    return ops.colsum(dy2, out=out) if out is not None else ops.colsum(dy2).to(_BF16)


# ----------------------------------------------------------------------------
PE_TWO_STAGE = _os.environ.get("SFCVIT_PE_FUSED", "0") != "1"   # "1": the fused gather-GEMM kernels (rounds 1-2) instead of gather + GEMM


class _PatchEmbed2(Function):
    """Gather, then project (round 3).  The fused kernels of rounds 1-2 make the gather the A loader of their own GEMM and run
    it at 280 TFLOP/s (192 us forward, 185 + 30 us backward at ViT-B / 256 images, the image read 2-3 times); materialising the
    token matrix once (154 MB of fp32 image in, 77 MB of bf16 tokens out) lets the projection and its weight gradient run on
    the persistent GEMMs at 950-1 150 TFLOP/s -- what the reference does (hilbert_embedding1D.py:36-43), minus its reshape copy."""

    @staticmethod
    def forward(ctx, x, pix, w, b, desc, order):
        B, (N, P) = x.shape[0], pix.shape
        tokens = ops.gather_tokens(_c(x), pix, desc, order)
        ctx.save_for_backward(tokens, w)
        ctx.small = (b,)
        return ops.gemm(tokens, w, bias=b).view(B, N, w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        tokens, w = ctx.saved_tensors
        dy2 = _c(dy).view(-1, dy.shape[-1])
        b = ctx.small[0]
        return None, None, _wgrad(dy2, tokens, w), (_bgrad(dy2, b) if b is not None else None), None, None


class _PatchEmbed(Function):
    @staticmethod
    def forward(ctx, x, pix, w, b, desc):
        tiled = desc is not None and pix.shape[1] == 256 and w.shape[0] % 256 == 0
        if tiled:
            # tiles / strips: the tiled kernel reads whole 16-pixel row segments of the image AS IT IS (fp32 from the
            # loader: one pass over 154 MB at ViT-B / 256 images; a bf16 copy first would cost that pass plus its own
            # write and re-read) and rounds to bf16 on the way into LDS -- the same rounding, the same tokens
            x = _c(x)
        else:
            # The generic kernels round the pixels to bf16 when they load them; doing that once up front gives the same
            # tokens bit for bit, halves the bytes both kernels pull through the texture path (they are bound by it: every
            # 128-column tile of D re-reads its token tile) and halves what is kept for backward (77 instead of 154 MB).
            x = _c(x if x.dtype == _BF16 else x.to(_BF16))
        ctx.save_for_backward(x, pix)
        ctx.D = w.shape[0]
        ctx.has_bias = b is not None
        ctx.desc = desc if tiled else None
        return ops.patch_embed_fwd(x, pix, w, b, ctx.desc)

    @staticmethod
    def backward(ctx, dy):
        x, pix = ctx.saved_tensors
        dw, db = ops.patch_embed_bwd(x, pix, _c(dy), ctx.D, want_bias=ctx.has_bias, desc=ctx.desc)
        return None, None, dw.to(_BF16), (db.to(_BF16) if db is not None else None), None


def _traced():
    """Under torch.compile the blocks run as `sfcvit::` custom ops (sfcvit/library.py): one graph, no breaks."""
    return torch.compiler.is_compiling()


def pe_two_stage(x, pix, D):
    """Gather + GEMM (default) or the fused kernels?  The GEMM wants 16-byte rows (P * C a multiple of 8), and the split only
    pays where the projection is big enough to be GPU-bound: two launches forward and four backward instead of one and two
    cost a launch-bound model more than the faster GEMM returns (CIFAR-size tokenizers: 0.3 GFLOP per pass)."""
    K = pix.shape[1] * x.shape[1]
    return PE_TWO_STAGE and K % 8 == 0 and 2.0 * x.shape[0] * pix.shape[0] * K * D >= 4e9


def patch_embed(x, pix, weight, bias, desc=None, order=None):
    """Curve gather + patchify + projection: x [B,C,H,W] (fp32 or bf16) -> [B,N,D] bf16.
    desc = ops.TileDesc of the pixel table (tokens are 16 x 16 tiles / 256-pixel strips) or None: picks the tile gather
    kernel (and the fused kernels of SFCVIT_PE_FUSED=1); order = ops.gather_order of the table on the device, or None."""
    if _traced():
        from . import library
        return library.patch_embed(x, pix, _bf(weight), _bf(bias), desc)
    if pe_two_stage(x, pix, weight.shape[0]):
        return _PatchEmbed2.apply(x, pix, _bf(weight), _bf(bias), desc, order)
    return _PatchEmbed.apply(x, pix, _bf(weight), _bf(bias), desc)


class _ResampleConcat(Function):
    """Linear resampling of the coarser levels to the first level's token count + concatenation (multi_hilbert.py:33-38)."""

    @staticmethod
    def forward(ctx, *levels):
        levels = [_c(t) for t in levels]
        ctx.n_tokens, ctx.D = [t.shape[1] for t in levels], levels[0].shape[2]
        return ops.hier_resample_concat(levels)

    @staticmethod
    def backward(ctx, dout):
        return tuple(ops.hier_resample_concat_bwd(_c(dout), ctx.n_tokens, ctx.D))


def hier_resample_concat(levels):
    """levels: list of [B, N_l, D] -> [B, N_0, L * D] bf16 (sfcvit_hier_resample_concat)."""
    return _ResampleConcat.apply(*[_bf(t) for t in levels])


# SFCVIT_HIER_ONE_KERNEL=1: gather + levels + concatenation + fusion Linear in ONE kernel (csrc/hier_tokenizer.hip,
# FUSE = true).  Default: the same kernel without its last phase (gather + levels + concatenation) followed by the
# fusion Linear on the persistent 8-phase GEMM -- measured faster at the reference's shape.
HIER_ONE_KERNEL = _os.environ.get("SFCVIT_HIER_ONE_KERNEL", "0") == "1"


class _HierTokenizer(Function):
    """Fused hierarchical tokenizer (ops.hier_tokenizer_fwd).  Saved for backward: the image (bf16, what the kernel's
    gather rounds to anyway) and the concatenated level outputs h; backward is the chain rule on existing kernels:
    dh = dy Wf, dWf = dy^T h, then per level the tokenizer weight gradient from that level's columns of dh."""

    @staticmethod
    def forward(ctx, x, wf, bf, n_levels, one_kernel, *rest):
        pix = rest[:n_levels]
        w = rest[n_levels:2 * n_levels]
        b = rest[2 * n_levels:3 * n_levels]
        x = _c(x if x.dtype == _BF16 else x.to(_BF16))
        if one_kernel:
            y, h = ops.hier_tokenizer_fwd(x, pix, w, b, wf, bf)
        else:
            _, h = ops.hier_tokenizer_fwd(x, pix, w, b, None, None)
            y = ops.gemm(h.view(-1, h.shape[-1]), wf, bias=bf).view(h.shape)
        ctx.save_for_backward(x, h, wf, *pix)
        ctx.n_levels, ctx.params = n_levels, (w, b, bf)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, wf = ctx.saved_tensors[:3]
        pix = ctx.saved_tensors[3:]
        w, b, bf = ctx.params
        L = ctx.n_levels
        E = h.shape[-1]
        D = E // L
        dy2, h2 = _c(dy).view(-1, E), h.view(-1, E)
        dwf = _wgrad(dy2, h2, wf)
        dbf = _bgrad(dy2, bf) if bf is not None else None
        dh = ops.gemm_dx(dy2, wf).view(h.shape)
        dws, dbs = [], []
        for l in range(L):
            dwl, dbl = ops.patch_embed_bwd(x, pix[l], _c(dh[..., l * D:(l + 1) * D]), D, want_bias=b[l] is not None)
            dws.append(dwl.to(_BF16))
            dbs.append(dbl.to(_BF16) if dbl is not None else None)
        return (None, dwf, dbf, None, None) + (None,) * L + tuple(dws) + tuple(dbs)


def hier_tokenizer(x, pix_list, weights, biases, fusion_weight, fusion_bias, one_kernel=None):
    """y = fusion(concat_l(level_l(x))) for levels with one common token count (reference:
    multiscale/multi_hilbert.py:31-40).  x [B,C,H,W] -> [B, N, L*D] bf16.  one_kernel: True = everything in one kernel,
    False = fused gather + levels + concatenation, then the fusion Linear as a GEMM; None = HIER_ONE_KERNEL."""
    L = len(pix_list)
    one = HIER_ONE_KERNEL if one_kernel is None else bool(one_kernel)
    return _HierTokenizer.apply(x, _bf(fusion_weight), _bf(fusion_bias), L, one, *pix_list, *[_bf(w) for w in weights],
                                *[_bf(b) for b in biases])


# ----------------------------------------------------------------------------
class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b, act):
        x2 = _c(x).view(-1, x.shape[-1])
        if act == ops.ACT_GELU:
            y, pre = ops.gemm(x2, w, bias=b, act=act, want_aux=True)
            ctx.save_for_backward(x2, w, pre)
        else:
            y = ops.gemm(x2, w, bias=b, act=act)
            ctx.save_for_backward(x2, w, y if act == ops.ACT_RELU else None)
        ctx.act, ctx.has_bias, ctx.shape = act, b is not None, x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, aux = ctx.saved_tensors
        dy2 = _c(dy).view(-1, dy.shape[-1])
        if ctx.act == ops.ACT_RELU:
            dy2 = torch.where(aux > 0, dy2, torch.zeros_like(dy2))
        elif ctx.act == ops.ACT_GELU:
            dy2 = ops.gelu_bwd(dy2, aux)
        dx = ops.gemm_dx(dy2, w).view(ctx.shape)
        return dx, _wgrad(dy2, x2, w), (_bgrad(dy2) if ctx.has_bias else None), None


def linear(x, weight, bias=None, act=ops.ACT_NONE):
    """y = act(x W^T + b).  Output widths that are not a multiple of 8 (a 10-class head) are computed on zero-padded
    rows of W (16-byte output rows) and sliced back; the padding is differentiable torch plumbing."""
    x, weight, bias = _bf(x), _bf(weight), _bf(bias)
    out = weight.shape[0]
    pad = (-out) % 8
    if pad:
        weight = torch.nn.functional.pad(weight, (0, 0, 0, pad))
        bias = torch.nn.functional.pad(bias, (0, pad)) if bias is not None else None
        return _Linear.apply(x, weight, bias, act)[..., :out]
    return _Linear.apply(x, weight, bias, act)


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        x2 = _c(x).view(-1, x.shape[-1])
        y, mean, rstd = ops.layernorm_fwd(x2, w, b, eps)
        ctx.save_for_backward(x2, mean, rstd, w)
        ctx.small = (b,)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, w = ctx.saved_tensors
        go = _ln_slots(w, ctx.small[0])
        dx, dg, db = ops.layernorm_bwd(_c(dy).view(-1, dy.shape[-1]), x2, mean, rstd, w, grad_out=go)
        return (dx.view(dy.shape), *_ln_grads(dg, db, go), None)


def layer_norm(x, weight, bias, eps=1e-5):
    return _LayerNorm.apply(_bf(x), _bf(weight), _bf(bias), eps)


class _Gelu(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(_c(dy), x)


def gelu(x):
    return _Gelu.apply(_bf(x))


class _GeluDrop(Function):
    """dropout_p(gelu_erf(x)) in one pass, the mask regenerated in backward (nn.GELU followed by nn.Dropout:
    MultiLayerPredictor's hidden layers, src/models/vit.py:303-318)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = _c(x)
        ctx.save_for_backward(x)
        ctx.p, ctx.seed, ctx.shape = p, seed, x.shape
        return ops.gelu_drop_fwd(x.view(-1, x.shape[-1]), p, seed).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = ops.gelu_drop_bwd(_c(dy).view(-1, x.shape[-1]), x.view(-1, x.shape[-1]), ctx.p, ctx.seed)
        return dx.view(ctx.shape), None, None


def gelu_dropout(x, p):
    """Dropout(p)(GELU(x)), training mode (p > 0 draws a fresh mask per call; p = 0 is gelu)."""
    if p <= 0:
        return gelu(x)
    return _GeluDrop.apply(_bf(x), float(p), ops.next_seed())


class _Attention(Function):
    """softmax(q k^T / sqrt(hd)) v per head on a packed projection [B, N, 3 * H * hd] (q | k | v thirds, head h at
    columns h*hd.. of each third): the core of nn.MultiheadAttention and of altvit.Attention (altvit.py:131-142)."""

    @staticmethod
    def forward(ctx, qkv, n_heads, scale):
        qkv = _c(qkv)
        out, lse = ops.attention_fwd(qkv, n_heads, scale=scale)
        ctx.save_for_backward(qkv, out, lse)
        ctx.n_heads, ctx.scale = n_heads, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        return ops.attention_bwd(qkv, out, lse, _c(dout), ctx.n_heads, scale=ctx.scale), None, None


def _head_padding(D, n_heads):
    """(head dim, kernel head dim) of a model width / head count; equal when the kernels have that head dim."""
    if D % n_heads:
        raise ValueError(f"attention: width {D} is not a multiple of n_heads = {n_heads}")
    hd = D // n_heads
    hp = ops.padded_head_dim(hd)
    if hp is None:
        raise ValueError(f"attention: head dim {hd} (= {D} / {n_heads}) exceeds the largest supported one, "
                         f"{max(ops.SUPPORTED_HEAD_DIMS)}")
    return hd, hp


def attention(qkv, n_heads):
    """Head dims the kernels do not have (32, 48, 96, ...) run zero-padded to the next one that exists: the padded
    q / k / v columns add nothing to q k^T or p v, the softmax scale stays 1 / sqrt(head dim)."""
    qkv = _bf(qkv)
    D = qkv.shape[-1] // 3
    hd, hp = _head_padding(D, n_heads)
    if hp == hd:
        return _Attention.apply(qkv, n_heads, None)
    lead = qkv.shape[:-1]
    padded = torch.nn.functional.pad(qkv.reshape(*lead, 3, n_heads, hd), (0, hp - hd)).reshape(*lead, 3 * n_heads * hp)
    out = _Attention.apply(padded, n_heads, 1.0 / math.sqrt(hd))
    return out.reshape(*lead, n_heads, hp)[..., :hd].reshape(*lead, D)


def _fast_gemm_shape(M, N, K):
    """Shapes the persistent 8-phase GEMM takes (csrc/gemm8p.hip: gemm8p_dispatch)."""
    return M >= 192 and N % 256 == 0 and K % 128 == 0 and K >= 256


# ----------------------------------------------------------------------------
class _Mixer(Function):
    """x + W2 gelu(W1 LN(x) + b1) + b2"""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, eps):
        x2 = _c(x).view(-1, x.shape[-1])
        z, mean, rstd = ops.layernorm_fwd(x2, ln_w, ln_b, eps)
        if _fast_gemm_shape(z.shape[0], w1.shape[0], z.shape[1]):
            # the persistent 8-phase GEMM has no erf-GELU / second-output epilogue: pre-activation from it (~1000 TFLOP/s)
            # + one elementwise pass beats the older kernel's fused epilogue (500 TFLOP/s): 174 vs 240 us at ViT-B / 256
            u = ops.gemm(z, w1, bias=b1)
            h = ops.gelu_fwd(u)
        else:
            h, u = ops.gemm(z, w1, bias=b1, act=ops.ACT_GELU, want_aux=True)
        y = ops.gemm(h, w2, bias=b2, residual=x2)
        ctx.save_for_backward(x2, mean, rstd, z, u, h, ln_w, w1, w2)
        ctx.small = (ln_b, b1, b2)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, z, u, h, ln_w, w1, w2 = ctx.saved_tensors
        ln_b, b1, b2 = ctx.small
        dy2 = _c(dy).view(-1, dy.shape[-1])
        dw2, db2 = _wgrad(dy2, h, w2), _bgrad(dy2, b2)
        if _fast_gemm_shape(dy2.shape[0], w2.shape[1], dy2.shape[1]):
            du = ops.gelu_bwd(ops.gemm_dx(dy2, w2), u)                     # as in forward: plain 8-phase GEMM + elementwise pass
        else:
            du = ops.gemm_dx(dy2, w2, aux_in=u, dact=ops.ACT_GELU)
        dw1, db1 = _wgrad(du, z, w1), _bgrad(du, b1)
        dz = ops.gemm_dx(du, w1)
        go = _ln_slots(ln_w, ln_b)
        dx, dg, dbeta = ops.layernorm_bwd(dz, x2, mean, rstd, ln_w, dx_add=dy2, grad_out=go)
        return (dx.view(dy.shape), *_ln_grads(dg, dbeta, go), dw1, db1, dw2, db2, None)


def mixer_block(x, ln_w, ln_b, w1, b1, w2, b2, eps=1e-5):
    if _traced():
        from . import library
        return library.mixer_block(_bf(x), _bf(ln_w), _bf(ln_b), _bf(w1), _bf(b1), _bf(w2), _bf(b2), eps)
    return _Mixer.apply(_bf(x), _bf(ln_w), _bf(ln_b), _bf(w1), _bf(b1), _bf(w2), _bf(b2), eps)


# ----------------------------------------------------------------------------
class _EncoderLayer(Function):
    """Post-norm transformer encoder layer (torch:nn/modules/transformer.py:951-982):
         a  = out_proj(attention(in_proj(x)));  x1 = LN1(x + drop1(a))
         f  = W2 drop(relu(W1 x1 + b1)) + b2;   y  = LN2(x1 + drop2(f))
    with dropout p on the attention probabilities as well.  p = 0 (eval) makes every mask the
    identity.  Masks are functions of (seed, element index), regenerated in backward."""

    @staticmethod
    def forward(ctx, x, in_w, in_b, out_w, out_b, n1_w, n1_b, w1, b1, w2, b2, n2_w, n2_b, n_heads, eps, p, seeds, scale):
        B, N, D = x.shape
        Da = in_w.shape[0] // 3                             # attention width: D, or n_heads * padded head dim (encoder_layer)
        sa, s1_, sf, s2_ = seeds
        x2 = _c(x).view(B * N, D)
        qkv = ops.gemm(x2, in_w, bias=in_b)
        o, lse = ops.attention_fwd(qkv.view(B, N, 3 * Da), n_heads, p, sa, scale=scale)
        s1 = ops.gemm(o.view(B * N, Da), out_w, bias=out_b, residual=x2, dropout_p=p, dropout_seed=s1_)
        x1, mean1, rstd1 = ops.layernorm_fwd(s1, n1_w, n1_b, eps)
        # The dX GEMM through ReLU + dropout needs only the sign pattern of h: a bit matrix written by linear1's epilogue
        # (19 MB instead of 308 MB per ViT-B layer read back in backward; sfcvit_gemm_args.actmask)
        hbits = (torch.empty((B * N, w1.shape[0] // 8), device=x.device, dtype=torch.uint8)
                 if USE_ACTMASK and w1.shape[0] % 16 == 0 else None)
        h = ops.gemm(x1, w1, bias=b1, act=ops.ACT_RELU, dropout_p=p, dropout_seed=sf, actmask=hbits)
        s2 = ops.gemm(h, w2, bias=b2, residual=x1, dropout_p=p, dropout_seed=s2_)
        y, mean2, rstd2 = ops.layernorm_fwd(s2, n2_w, n2_b, eps)
        ctx.hbits = hbits
        ctx.save_for_backward(x2, qkv, o, lse, s1, mean1, rstd1, x1, h, s2, mean2, rstd2,
                              in_w, out_w, n1_w, w1, w2, n2_w)
        ctx.n_heads, ctx.shape, ctx.p, ctx.seeds, ctx.scale = n_heads, (B, N, D), p, seeds, scale
        ctx.small = (in_b, out_b, n1_b, b1, b2, n2_b)      # parameters only needed for their gradient slots
        return y.view(B, N, D)

    @staticmethod
    def backward(ctx, dy):
        (x2, qkv, o, lse, s1, mean1, rstd1, x1, h, s2, mean2, rstd2,
         in_w, out_w, n1_w, w1, w2, n2_w) = ctx.saved_tensors
        B, N, D = ctx.shape
        p = ctx.p
        sa, s1_, sf, s2_ = ctx.seeds
        dy2 = _c(dy).view(B * N, D)
        in_b, out_b, n1_b, b1, b2, n2_b = ctx.small
        ss = _SideStream(dy2.device) if SIDE_STREAM_WGRAD else None
        wg = (lambda a_, b_, w_: ss.run(lambda: _wgrad(a_, b_, w_), a_, b_)) if ss else _wgrad

        def ln_bwd(dyv, s, mean, rstd, g_w, g_b, prev_bias, seed):
            """LayerNorm backward; also returns the column sums of the gradient it hands to the sub-layer = the bias
            gradient of linear2 / out_proj, for free in the same pass.  Gradients go straight into their slots when
            all three parameters have one."""
            slots = (_slot(g_w), _slot(g_b), _slot(prev_bias))
            go = slots if all(t is not None for t in slots) else None
            if p > 0:
                dsv, dg, dbt, dsub, dcol = ops.layernorm_bwd(dyv, s, mean, rstd, g_w, drop_p=p, drop_seed=seed,
                                                             want_colsum=True, grad_out=go)
            else:
                dsv, dg, dbt, dcol = ops.layernorm_bwd(dyv, s, mean, rstd, g_w, want_colsum=True, grad_out=go)
                dsub = dsv
            if go is None:
                dg, dbt, dcol = dg.to(_BF16), dbt.to(_BF16), dcol.to(_BF16)
            return dsv, dg, dbt, dsub, dcol

        ds2, dg2, dbt2, df, db2 = ln_bwd(dy2, s2, mean2, rstd2, n2_w, n2_b, b2, s2_)
        dw2 = wg(df, h, w2)
        # h is stored after relu + dropout: (h > 0) is the joint mask, 1/(1-p) the dropout scale
        # ... and the bias gradient of linear1 is the column sum of dh: fused into that GEMM's epilogue
        b1_slot = _slot(b1)
        dh, db1 = ops.gemm_dx(df, w2, aux_in=h, dact=ops.ACT_RELU, dact_scale=1.0 / (1.0 - p),
                              colsum=b1_slot if b1_slot is not None else True, actmask=ctx.hbits)
        if b1_slot is None:
            db1 = db1.to(_BF16)
        dw1 = wg(dh, x1, w1)
        dx1 = ops.gemm_dx(dh, w1, residual=ds2)
        ds1, dg1, dbt1, da, dbo = ln_bwd(dx1, s1, mean1, rstd1, n1_w, n1_b, out_b, s1_)
        Da = in_w.shape[0] // 3
        o2 = o.view(B * N, Da)
        dwo = wg(da, o2, out_w)
        do = ops.gemm_dx(da, out_w)
        # the in_proj bias gradient (column sums of dqkv) comes out of the attention backward itself
        bi_slot = _slot(in_b)
        dqkv, dbi = ops.attention_bwd(qkv.view(B, N, 3 * Da), o, lse, do.view(B, N, Da), ctx.n_heads, p, sa,
                                      colsum=bi_slot if bi_slot is not None else True, scale=ctx.scale)
        dqkv = dqkv.view(B * N, 3 * Da)
        if bi_slot is None:
            dbi = dbi.to(_BF16)
        dwi = wg(dqkv, x2, in_w)
        dx = ops.gemm_dx(dqkv, in_w, residual=ds1)
        if ss:
            ss.join(dw2, dw1, dwo, dwi)
        return (dx.view(B, N, D), dwi, dbi, dwo, dbo, dg1, dbt1, dw1, db1, dw2, db2, dg2, dbt2, None, None, None, None, None)


def encoder_layer(x, in_w, in_b, out_w, out_b, n1_w, n1_b, w1, b1, w2, b2, n2_w, n2_b, n_heads, eps=1e-5,
                  dropout_p=0.0):
    """dropout_p > 0 = training-mode nn.TransformerEncoderLayer (fresh masks every call).
    A head dim the attention kernels do not have (embed_dim / n_heads = 32, 48, 96, ...) runs on the next one that
    exists: in_proj's rows and out_proj's columns are zero-padded per head (differentiable torch plumbing on the
    weights, 3 D^2 elements -- not on the activations), so q, k, v come out of the projection GEMM already padded, the
    padded columns add nothing to q k^T or p v, and out_proj ignores them; the softmax scale stays 1 / sqrt(head dim)."""
    args = [_bf(t) for t in (x, in_w, in_b, out_w, out_b, n1_w, n1_b, w1, b1, w2, b2, n2_w, n2_b)]
    D = args[0].shape[-1]
    hd, hp = _head_padding(D, n_heads)
    scale = None
    if hp != hd:
        pad = torch.nn.functional.pad
        args[1] = pad(args[1].reshape(3, n_heads, hd, D), (0, 0, 0, hp - hd)).reshape(3 * n_heads * hp, D)
        if args[2] is not None:
            args[2] = pad(args[2].reshape(3, n_heads, hd), (0, hp - hd)).reshape(3 * n_heads * hp)
        args[3] = pad(args[3].reshape(D, n_heads, hd), (0, hp - hd)).reshape(D, n_heads * hp)
        scale = 1.0 / math.sqrt(hd)
    if _traced():
        from . import library      # (dropout seeds are drawn inside the op: nothing data-dependent in the traced graph)
        return library.encoder_layer(args, n_heads, eps, float(dropout_p), scale)
    seeds = tuple(ops.next_seed() for _ in range(4)) if dropout_p > 0 else (0, 0, 0, 0)
    return _EncoderLayer.apply(*args, n_heads, eps, float(dropout_p), seeds, scale)


# ----------------------------------------------------------------------------
def _pad_rows(t, rows):
    if t.shape[0] == rows:
        return t
    out = torch.zeros((rows,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    out[: t.shape[0]] = t
    return out


class _Head(Function):
    """LN -> h = z W_emb^T -> y = <h, W_seq> -> GELU -> Dropout(p) -> classifier.
    The classifier is computed on a class count padded to a multiple of 8 (16-byte rows)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w_emb, w_seq, wc, bc, eps, p, seed):
        B, N, D = x.shape
        R, O, C = w_emb.shape[0], w_seq.shape[0], wc.shape[0]
        x2 = _c(x).view(B * N, D)
        z, mean, rstd = ops.layernorm_fwd(x2, ln_w, ln_b, eps)
        h = ops.gemm(z, w_emb)                                      # [B*N, R]
        y1 = ops.gemm(h.view(B, N * R), w_seq.view(O, N * R))      # [B, O]
        a = ops.gelu_drop_fwd(y1, p, seed) if p > 0 else ops.gelu_fwd(y1)
        cpad = (C + 7) // 8 * 8
        wc_p, bc_p = _pad_rows(wc, cpad), _pad_rows(bc, cpad)
        logits = ops.gemm(a, wc_p, bias=bc_p)                       # [B, cpad]
        ctx.save_for_backward(x2, mean, rstd, z, h, y1, a, ln_w, w_emb, w_seq, wc_p)
        ctx.small = (ln_b, bc)
        ctx.dims = (B, N, D, R, O, C, cpad)
        ctx.p, ctx.seed = p, seed
        return logits[:, :C] if cpad != C else logits

    @staticmethod
    def backward(ctx, dlogits):
        x2, mean, rstd, z, h, y1, a, ln_w, w_emb, w_seq, wc_p = ctx.saved_tensors
        B, N, D, R, O, C, cpad = ctx.dims
        ln_b, bc = ctx.small
        if cpad == C:                                   # (wc_p IS the classifier weight then: gradients straight into the slots)
            dl = _c(dlogits)
            dwc, dbc = _wgrad(dl, a, wc_p), _bgrad(dl, bc)
        else:
            dl = torch.zeros((B, cpad), device=dlogits.device, dtype=_BF16)
            dl[:, :C] = dlogits
            dwc = ops.gemm(dl, a, a_kmajor=True, b_kmajor=True)[:C]
            dbc = ops.colsum(dl)[:C].to(_BF16)
        da = ops.gemm_dx(dl, wc_p)                      # [B, O]
        dy1 = ops.gelu_drop_bwd(da, y1, ctx.p, ctx.seed) if ctx.p > 0 else ops.gelu_bwd(da, y1)
        h2 = h.view(B, N * R)
        sseq = _slot(w_seq)
        dwseq = ops.gemm(dy1, h2, a_kmajor=True, b_kmajor=True, out=None if sseq is None else sseq.view(O, N * R))
        dwseq = dwseq.view(O, N, R) if sseq is None else sseq
        dh = ops.gemm(dy1, w_seq.view(O, N * R), b_kmajor=True).view(B * N, R)
        dwemb = _wgrad(dh, z, w_emb)
        dz = ops.gemm_dx(dh, w_emb)
        go = _ln_slots(ln_w, ln_b)
        dx, dg, dbeta = ops.layernorm_bwd(dz, x2, mean, rstd, ln_w, grad_out=go)
        return (dx.view(B, N, D), *_ln_grads(dg, dbeta, go), dwemb, dwseq, dwc, dbc, None, None, None)


def predictor_head(x, ln_w, ln_b, w_emb, w_seq, wc, bc, eps=1e-5, dropout_p=0.0):
    if _traced():
        from . import library
        return library.predictor_head(_bf(x), _bf(ln_w), _bf(ln_b), _bf(w_emb), _c(_bf(w_seq)), _bf(wc), _bf(bc), eps, float(dropout_p))
    seed = ops.next_seed() if dropout_p > 0 else 0
    return _Head.apply(_bf(x), _bf(ln_w), _bf(ln_b), _bf(w_emb), _c(_bf(w_seq)), _bf(wc), _bf(bc), eps,
                       float(dropout_p), seed)


# ----------------------------------------------------------------------------
class _SoftCE(Function):
    @staticmethod
    def forward(ctx, logits, targets):
        B, C = logits.shape
        base = _c(logits)                   # [B, C], row stride C (2-byte loads: no alignment needed)
        rows, dl = ops.soft_ce(base, _c(targets.float()), C, 1.0 / B)
        ctx.save_for_backward(dl)
        ctx.C = C
        return rows.mean()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl[:, : ctx.C] * g.to(dl.dtype)), None


def soft_target_cross_entropy(logits, targets):
    """-(targets * log_softmax(logits)).sum(-1).mean() with the gradient produced in the same pass."""
    if _traced():
        from . import library
        return library.soft_ce(_bf(logits), targets)
    return _SoftCE.apply(_bf(logits), targets)


# ----------------------------------------------------------------------------
# torch.compile (main.py:284 wraps the model in torch.compile(mode="reduce-overhead")).  The blocks a VisionTransformer{,1D}
# is made of -- patch_embed, mixer_block, encoder_layer, predictor_head, soft_target_cross_entropy -- are registered as
# `sfcvit::` custom ops with fake kernels and autograd formulas (sfcvit/library.py) and take that path whenever Dynamo is
# tracing: the model compiles into ONE graph and "reduce-overhead" replays it from a hipGraph.  The remaining pieces
# (used by the hierarchical tokenizers, altvit and MultiLayerPredictor(n_layers > 2)) stay opaque: Dynamo breaks the graph
# around each and runs it as written.
# ----------------------------------------------------------------------------
for _name in ("hier_tokenizer", "linear", "layer_norm", "gelu", "gelu_dropout", "attention"):
    globals()[_name] = torch.compiler.disable(globals()[_name], recursive=True)
del _name
