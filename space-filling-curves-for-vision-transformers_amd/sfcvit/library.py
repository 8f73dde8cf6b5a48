"""`torch.library` surface of the hot path: the blocks of sfcvit.functional as traceable custom ops.

The reference wraps its model in `torch.compile(model, mode="reduce-overhead")` (main.py:284).  The blocks of
sfcvit.functional are autograd.Functions around ctypes launches of the C ABI: opaque to Dynamo.  Here each block is
registered as a pair of `torch.library.custom_op`s in the `sfcvit::` namespace -- `<block>` (forward: the result plus
what backward needs) and `<block>_bwd` -- with fake (meta) kernels and an autograd formula, so that Dynamo traces a model
built from them into ONE graph (no graph breaks), AOTAutograd can build the backward graph, and
`mode="reduce-overhead"` captures the launches into a hipGraph (every C-ABI entry is capturable: nothing allocates
outside torch's caching allocator or synchronises).  sfcvit.functional's public functions take this path whenever
`torch.compiler.is_compiling()`; eager calls keep the autograd.Function path, whose backward writes gradients straight
into the optimizer's flat buffer -- a traced op may not return views of a tensor it was not given, so under compile
every gradient is a fresh tensor and FusedAdamW adopts (copies) it.

The op bodies call the very same Function.forward / .backward code with a stand-in for autograd's ctx: one
implementation of the arithmetic, two ways of being called.

Training-mode dropout traces as well: the seeds are drawn inside the op bodies (from torch's CPU generator, as eagerly, so
a compiled and an eager forward under the same torch.manual_seed draw the same masks) and travel to backward as a small CPU
tensor -- which makes "reduce-overhead" skip its hipGraph for TRAINING graphs (inductor does not capture graphs that hold
CPU tensors); use sfcvit.training.GraphedTrainStep (device-resident step state) for graphed training.  Compiled
inference / evaluation holds device tensors only and replays bit-identically.
"""
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import functional as F
from . import ops

_BF16 = torch.bfloat16


class _Ctx:
    """What Function.forward / .backward touch of autograd's ctx."""

    def __init__(self):
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class _no_slots:
    """Gradients as fresh tensors (functional._slot returns None): a custom op may not return views of the flat buffer."""

    def __enter__(self):
        self.prev, F._SLOTS_OFF = F._SLOTS_OFF, True

    def __exit__(self, *exc):
        F._SLOTS_OFF = self.prev
        return False


def _e(like, dtype=None):
    """Empty placeholder for an absent optional OUTPUT (custom ops return tensors, not None)."""
    return torch.empty(0, device=like.device, dtype=dtype or like.dtype)


def _opt(t):
    return t if t is not None and t.numel() else None


# ----------------------------------------------------------------------------------------------------------------------
# patch_embed: x [B,C,H,W] (fp32 / bf16), pix [N,P] int32, w [D, P*C] bf16, b [D]?, tile descriptor (device words + host
# facts: mode, ncls, cnt...) -> y [B,N,D] bf16, xs (the bf16 image kept for backward; empty when x itself is kept)
# ----------------------------------------------------------------------------------------------------------------------
def _desc_from(desc_dev, meta):
    if desc_dev is None or not meta:
        return None
    d = ops.TileDesc.__new__(ops.TileDesc)
    d.mode, d.ncls, d.cnt, d.dev = int(meta[0]), int(meta[1]), [int(v) for v in meta[2:]], desc_dev
    return d


@torch.library.custom_op("sfcvit::patch_embed", mutates_args=())
def patch_embed_op(x: Tensor, pix: Tensor, w: Tensor, b: Optional[Tensor], desc_dev: Optional[Tensor],
                   desc_meta: List[int]) -> Tuple[Tensor, Tensor]:
    ctx = _Ctx()
    y = F._PatchEmbed.forward(ctx, x, pix, w, b, _desc_from(desc_dev, desc_meta))
    xs = ctx.saved_tensors[0]
    return y, (xs if xs.data_ptr() != x.data_ptr() else _e(x, _BF16))


@patch_embed_op.register_fake
def _(x, pix, w, b, desc_dev, desc_meta):
    tiled = desc_dev is not None and pix.shape[1] == 256 and w.shape[0] % 256 == 0
    keeps_x = x.is_contiguous() and (tiled or x.dtype == _BF16)
    y = x.new_empty((x.shape[0], pix.shape[0], w.shape[0]), dtype=_BF16)
    return y, (x.new_empty(0, dtype=_BF16) if keeps_x else x.new_empty(x.shape, dtype=x.dtype if tiled else _BF16))


@torch.library.custom_op("sfcvit::patch_embed_bwd", mutates_args=())
def patch_embed_bwd_op(dy: Tensor, xs: Tensor, pix: Tensor, D: int, has_bias: bool, desc_dev: Optional[Tensor],
                       desc_meta: List[int]) -> Tuple[Tensor, Tensor]:
    ctx = _Ctx()
    ctx.saved_tensors, ctx.D, ctx.has_bias = (xs, pix), D, has_bias
    tiled = desc_dev is not None and pix.shape[1] == 256 and D % 256 == 0
    ctx.desc = _desc_from(desc_dev, desc_meta) if tiled else None
    _, _, dw, db, _ = F._PatchEmbed.backward(ctx, dy)
    return dw, (db if db is not None else _e(dy))


@patch_embed_bwd_op.register_fake
def _(dy, xs, pix, D, has_bias, desc_dev, desc_meta):
    C = xs.shape[1]
    return dy.new_empty((D, pix.shape[1] * C)), dy.new_empty(D if has_bias else 0)


def _pe_setup(ctx, inputs, output):
    x, pix, w, b, desc_dev, desc_meta = inputs
    _, xs = output
    ctx.save_for_backward(xs if xs.numel() else x, pix, desc_dev)
    ctx.D, ctx.has_bias, ctx.meta = w.shape[0], b is not None, desc_meta
    ctx.set_materialize_grads(False)


def _pe_backward(ctx, dy, _dxs):
    xs, pix, desc_dev = ctx.saved_tensors
    dw, db = torch.ops.sfcvit.patch_embed_bwd(dy.contiguous(), xs, pix, ctx.D, ctx.has_bias, desc_dev, ctx.meta)
    return None, None, dw, (db if ctx.has_bias else None), None, None


patch_embed_op.register_autograd(_pe_backward, setup_context=_pe_setup)


# the two-stage form (functional._PatchEmbed2: gather, then the projection on the GEMM kernels): y, tokens
@torch.library.custom_op("sfcvit::patch_embed2", mutates_args=())
def patch_embed2_op(x: Tensor, pix: Tensor, w: Tensor, b: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    ctx = _Ctx()
    y = F._PatchEmbed2.forward(ctx, x, pix, w, b)
    return y.contiguous(), ctx.saved_tensors[0]


@patch_embed2_op.register_fake
def _(x, pix, w, b):
    B, (N, P) = x.shape[0], pix.shape
    return x.new_empty((B, N, w.shape[0]), dtype=_BF16), x.new_empty((B * N, (P * x.shape[1] + 7) // 8 * 8), dtype=_BF16)


@torch.library.custom_op("sfcvit::patch_embed2_bwd", mutates_args=())
def patch_embed2_bwd_op(dy: Tensor, tokens: Tensor, w: Tensor, has_bias: bool) -> Tuple[Tensor, Tensor]:
    ctx = _Ctx()
    ctx.saved_tensors, ctx.small = (tokens, w), (w.new_empty(0) if has_bias else None,)
    with _no_slots():
        _, _, dw, db = F._PatchEmbed2.backward(ctx, dy)
    return dw, (db if db is not None else _e(dy))


@patch_embed2_bwd_op.register_fake
def _(dy, tokens, w, has_bias):
    return dy.new_empty(w.shape), dy.new_empty(w.shape[0] if has_bias else 0)


def _pe2_setup(ctx, inputs, output):
    x, pix, w, b = inputs
    ctx.save_for_backward(output[1], w)
    ctx.has_bias = b is not None
    ctx.set_materialize_grads(False)


def _pe2_backward(ctx, dy, _dtokens):
    tokens, w = ctx.saved_tensors
    dw, db = torch.ops.sfcvit.patch_embed2_bwd(dy.contiguous(), tokens, w, ctx.has_bias)
    return None, None, dw, (db if ctx.has_bias else None)


patch_embed2_op.register_autograd(_pe2_backward, setup_context=_pe2_setup)


# ----------------------------------------------------------------------------------------------------------------------
# mixer_block
# ----------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("sfcvit::mixer_block", mutates_args=())
def mixer_block_op(x: Tensor, ln_w: Tensor, ln_b: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor,
                   eps: float) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    ctx = _Ctx()
    y = F._Mixer.forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, eps)
    _, mean, rstd, z, u, h = ctx.saved_tensors[:6]
    return y, mean, rstd, z, u, h


@mixer_block_op.register_fake
def _(x, ln_w, ln_b, w1, b1, w2, b2, eps):
    M = x.numel() // x.shape[-1]
    f32 = lambda *s: x.new_empty(s, dtype=torch.float32)      # noqa: E731
    return (x.new_empty(x.shape), f32(M), f32(M), x.new_empty((M, x.shape[-1])), x.new_empty((M, w1.shape[0])),
            x.new_empty((M, w1.shape[0])))


@torch.library.custom_op("sfcvit::mixer_block_bwd", mutates_args=())
def mixer_block_bwd_op(dy: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, z: Tensor, u: Tensor, h: Tensor, ln_w: Tensor,
                       w1: Tensor, w2: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    ctx = _Ctx()
    ctx.saved_tensors = (F._c(x).view(-1, x.shape[-1]), mean, rstd, z, u, h, ln_w, w1, w2)
    ctx.small = (None, None, None)                      # (no gradient slots inside a traced op: fresh tensors)
    with _no_slots():
        return tuple(F._Mixer.backward(ctx, dy)[:7])


@mixer_block_bwd_op.register_fake
def _(dy, x, mean, rstd, z, u, h, ln_w, w1, w2):
    ne = lambda t: t.new_empty(t.shape)                       # noqa: E731
    return ne(x), ne(ln_w), ne(ln_w), ne(w1), w1.new_empty(w1.shape[0]), ne(w2), w2.new_empty(w2.shape[0])


def _mixer_setup(ctx, inputs, output):
    x, ln_w, _, w1, _, w2, _, _ = inputs
    ctx.save_for_backward(x, *output[1:], ln_w, w1, w2)
    ctx.set_materialize_grads(False)


def _mixer_backward(ctx, dy, *_):
    x, mean, rstd, z, u, h, ln_w, w1, w2 = ctx.saved_tensors
    return (*torch.ops.sfcvit.mixer_block_bwd(dy.contiguous(), x, mean, rstd, z, u, h, ln_w, w1, w2), None)


mixer_block_op.register_autograd(_mixer_backward, setup_context=_mixer_setup)


# ----------------------------------------------------------------------------------------------------------------------
# encoder_layer (post-norm nn.TransformerEncoderLayer; seeds = [] -> drawn inside the op when p > 0)
# ----------------------------------------------------------------------------------------------------------------------
_ENC_SAVED = 12     # qkv, o, lse, s1, mean1, rstd1, x1, h, s2, mean2, rstd2 (+ hbits, seeds below)


@torch.library.custom_op("sfcvit::encoder_layer", mutates_args=())
def encoder_layer_op(x: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor, n1_w: Tensor, n1_b: Tensor,
                     w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, n2_w: Tensor, n2_b: Tensor, n_heads: int, eps: float,
                     p: float, seeds: List[int], scale: Optional[float]) -> List[Tensor]:
    ctx = _Ctx()
    if p > 0 and not seeds:
        seeds = [ops.next_seed() for _ in range(4)]
    y = F._EncoderLayer.forward(ctx, x, in_w, in_b, out_w, out_b, n1_w, n1_b, w1, b1, w2, b2, n2_w, n2_b, n_heads, eps, p,
                                tuple(seeds) if seeds else (0, 0, 0, 0), scale)
    saved = list(ctx.saved_tensors[1:_ENC_SAVED])
    hbits = ctx.hbits if ctx.hbits is not None else _e(x, torch.uint8)
    # what backward regenerates the masks from: a CPU tensor (read back without a device sync); without dropout an empty
    # DEVICE tensor, so that an inference / evaluation graph holds no CPU tensor and "reduce-overhead" can capture it
    seed_t = torch.tensor(list(ctx.seeds), dtype=torch.int64) if p > 0 else _e(x, torch.int64)
    return [y] + saved + [hbits, seed_t]


@encoder_layer_op.register_fake
def _(x, in_w, in_b, out_w, out_b, n1_w, n1_b, w1, b1, w2, b2, n2_w, n2_b, n_heads, eps, p, seeds, scale):
    B, N, D = x.shape
    M, Da, Fd = B * N, in_w.shape[0] // 3, w1.shape[0]
    bf = lambda *s: x.new_empty(s, dtype=_BF16)              # noqa: E731
    f32 = lambda *s: x.new_empty(s, dtype=torch.float32)      # noqa: E731
    use_bits = F.USE_ACTMASK and Fd % 16 == 0
    return [bf(B, N, D), bf(M, 3 * Da), bf(B, N, Da), f32(B, n_heads, N), bf(M, D), f32(M), f32(M), bf(M, D), bf(M, Fd), bf(M, D),
            f32(M), f32(M), x.new_empty((M, Fd // 8) if use_bits else (0,), dtype=torch.uint8),
            torch.empty(4, dtype=torch.int64) if p > 0 else x.new_empty(0, dtype=torch.int64)]


@torch.library.custom_op("sfcvit::encoder_layer_bwd", mutates_args=())
def encoder_layer_bwd_op(dy: Tensor, x: Tensor, saved: List[Tensor], weights: List[Tensor], n_heads: int, p: float,
                         scale: Optional[float]) -> List[Tensor]:
    qkv, o, lse, s1, mean1, rstd1, x1, h, s2, mean2, rstd2, hbits, seed_t = saved
    in_w, out_w, n1_w, w1, w2, n2_w = weights
    ctx = _Ctx()
    B, N, D = x.shape
    ctx.saved_tensors = (F._c(x).view(B * N, D), qkv, o, lse, s1, mean1, rstd1, x1, h, s2, mean2, rstd2, in_w, out_w, n1_w, w1, w2, n2_w)
    ctx.hbits = _opt(hbits)
    ctx.n_heads, ctx.shape, ctx.p, ctx.scale = n_heads, (B, N, D), p, scale
    ctx.seeds = tuple(int(v) for v in seed_t.tolist()) if p > 0 else (0, 0, 0, 0)
    ctx.small = (None,) * 6
    with _no_slots():
        return list(F._EncoderLayer.backward(ctx, dy)[:13])


@encoder_layer_bwd_op.register_fake
def _(dy, x, saved, weights, n_heads, p, scale):
    in_w, out_w, n1_w, w1, w2, n2_w = weights
    ne = lambda t: t.new_empty(t.shape)                       # noqa: E731
    v = lambda t, n: t.new_empty(n)                           # noqa: E731
    D = x.shape[-1]
    return [ne(x), ne(in_w), v(in_w, in_w.shape[0]), ne(out_w), v(out_w, D), v(n1_w, D), v(n1_w, D), ne(w1), v(w1, w1.shape[0]),
            ne(w2), v(w2, D), v(n2_w, D), v(n2_w, D)]


def _enc_setup(ctx, inputs, output):
    x, in_w, _, out_w, _, n1_w, _, w1, _, w2, _, n2_w, _, n_heads, _, p, seeds, scale = inputs
    ctx.save_for_backward(x, *output[1:], in_w, out_w, n1_w, w1, w2, n2_w)
    ctx.n_heads, ctx.p, ctx.scale, ctx.n_seeds = n_heads, p, scale, len(seeds)
    ctx.set_materialize_grads(False)


def _enc_backward(ctx, grads):
    t = ctx.saved_tensors
    x, saved, weights = t[0], list(t[1:14]), list(t[14:20])
    g = torch.ops.sfcvit.encoder_layer_bwd(grads[0].contiguous(), x, saved, weights, ctx.n_heads, ctx.p, ctx.scale)
    # (this op returns a Tensor[]: torch.library then checks the gradients against the pytree of the inputs, in which the
    # int[] argument is a list node -- an op without tensor lists gets one None per argument instead)
    return (*g, None, None, None, [None] * ctx.n_seeds, None)


encoder_layer_op.register_autograd(_enc_backward, setup_context=_enc_setup)


# ----------------------------------------------------------------------------------------------------------------------
# predictor_head (MultiLayerPredictor, n_layers = 2)
# ----------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("sfcvit::predictor_head", mutates_args=())
def predictor_head_op(x: Tensor, ln_w: Tensor, ln_b: Tensor, w_emb: Tensor, w_seq: Tensor, wc: Tensor, bc: Tensor, eps: float,
                      p: float, seed: int) -> List[Tensor]:
    ctx = _Ctx()
    if p > 0 and seed == 0:
        seed = ops.next_seed()
    logits = F._Head.forward(ctx, x, ln_w, ln_b, w_emb, w_seq, wc, bc, eps, p, seed)
    _, mean, rstd, z, h, y1, a, _, _, _, wc_p = ctx.saved_tensors
    return [logits.contiguous(), mean, rstd, z, h, y1, a, wc_p if wc_p.data_ptr() != wc.data_ptr() else wc_p.clone(),
            torch.tensor([seed], dtype=torch.int64) if p > 0 else _e(x, torch.int64)]


@predictor_head_op.register_fake
def _(x, ln_w, ln_b, w_emb, w_seq, wc, bc, eps, p, seed):
    B, N, D = x.shape
    R, O, C = w_emb.shape[0], w_seq.shape[0], wc.shape[0]
    cpad = (C + 7) // 8 * 8
    bf = lambda *s: x.new_empty(s, dtype=_BF16)              # noqa: E731
    f32 = lambda *s: x.new_empty(s, dtype=torch.float32)      # noqa: E731
    return [bf(B, C), f32(B * N), f32(B * N), bf(B * N, D), bf(B * N, R), bf(B, O), bf(B, O), bf(cpad, O),
            torch.empty(1, dtype=torch.int64) if p > 0 else x.new_empty(0, dtype=torch.int64)]


@torch.library.custom_op("sfcvit::predictor_head_bwd", mutates_args=())
def predictor_head_bwd_op(dlogits: Tensor, x: Tensor, saved: List[Tensor], weights: List[Tensor], n_classes: int,
                          p: float) -> List[Tensor]:
    mean, rstd, z, h, y1, a, wc_p, seed_t = saved
    ln_w, w_emb, w_seq = weights
    B, N, D = x.shape
    ctx = _Ctx()
    ctx.saved_tensors = (F._c(x).view(B * N, D), mean, rstd, z, h, y1, a, ln_w, w_emb, w_seq, wc_p)
    ctx.dims = (B, N, D, w_emb.shape[0], w_seq.shape[0], n_classes, wc_p.shape[0])
    ctx.small = (None, None)
    ctx.p, ctx.seed = p, (int(seed_t.item()) if p > 0 else 0)
    with _no_slots():
        g = F._Head.backward(ctx, dlogits)[:7]
    return [t.contiguous() for t in g]


@predictor_head_bwd_op.register_fake
def _(dlogits, x, saved, weights, n_classes, p):
    ln_w, w_emb, w_seq = weights
    wc_p = saved[6]
    ne = lambda t: t.new_empty(t.shape)                       # noqa: E731
    return [ne(x), ne(ln_w), ne(ln_w), ne(w_emb), ne(w_seq), wc_p.new_empty((n_classes, wc_p.shape[1])), wc_p.new_empty(n_classes)]


def _head_setup(ctx, inputs, output):
    x, ln_w, _, w_emb, w_seq, wc, _, _, p, _ = inputs
    ctx.save_for_backward(x, *output[1:], ln_w, w_emb, w_seq)
    ctx.C, ctx.p = wc.shape[0], p
    ctx.set_materialize_grads(False)


def _head_backward(ctx, grads):
    t = ctx.saved_tensors
    g = torch.ops.sfcvit.predictor_head_bwd(grads[0].contiguous(), t[0], list(t[1:9]), list(t[9:12]), ctx.C, ctx.p)
    return (*g, None, None, None)


predictor_head_op.register_autograd(_head_backward, setup_context=_head_setup)


# ----------------------------------------------------------------------------------------------------------------------
# soft-target cross entropy: loss and the logit gradient in one pass
# ----------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("sfcvit::soft_ce", mutates_args=())
def soft_ce_op(logits: Tensor, targets: Tensor) -> Tuple[Tensor, Tensor]:
    ctx = _Ctx()
    loss = F._SoftCE.forward(ctx, logits, targets)
    return loss, ctx.saved_tensors[0]


@soft_ce_op.register_fake
def _(logits, targets):
    return logits.new_empty((), dtype=torch.float32), logits.new_empty(logits.shape)


def _ce_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.set_materialize_grads(False)


def _ce_backward(ctx, g, _):
    (dl,) = ctx.saved_tensors
    return dl * g.to(dl.dtype), None


soft_ce_op.register_autograd(_ce_backward, setup_context=_ce_setup)


# ----------------------------------------------------------------------------------------------------------------------
# entry points used by sfcvit.functional when torch.compiler.is_compiling()
# ----------------------------------------------------------------------------------------------------------------------
def patch_embed(x, pix, weight, bias, desc):
    if F.pe_two_stage(x, pix, weight.shape[0]):
        return torch.ops.sfcvit.patch_embed2(x.contiguous(), pix, weight, bias)[0]
    meta = [desc.mode, desc.ncls, *desc.cnt] if desc is not None else []
    return torch.ops.sfcvit.patch_embed(x, pix, weight, bias, desc.dev if desc is not None else None, meta)[0]


def mixer_block(x, ln_w, ln_b, w1, b1, w2, b2, eps):
    return torch.ops.sfcvit.mixer_block(x, ln_w, ln_b, w1, b1, w2, b2, eps)[0]


def encoder_layer(args, n_heads, eps, p, scale):
    return torch.ops.sfcvit.encoder_layer(*args, n_heads, eps, p, [], scale)[0]


def predictor_head(x, ln_w, ln_b, w_emb, w_seq, wc, bc, eps, p):
    return torch.ops.sfcvit.predictor_head(x, ln_w, ln_b, w_emb, w_seq, wc, bc, eps, p, 0)[0]


def soft_ce(logits, targets):
    return torch.ops.sfcvit.soft_ce(logits, targets)[0]
