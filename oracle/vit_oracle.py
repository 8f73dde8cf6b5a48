"""fp32 PyTorch-CPU restatement of the reference hot path (TEST INFRASTRUCTURE).

Written as explicit math on a plain {reference state_dict key: tensor} mapping,
so that it shares no code with the product modules and none with torch's
nn.TransformerEncoderLayer.  Pinned by tests/golden/model_*.json, which
tools/make_golden.py produced from the imported reference classes.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file (see oracle/__init__.py).

Reference citations (relative to /root/reference unless prefixed torch:):
  tokenizers   src/tokenizers/_1D/hilbert_embedding1D.py:30-44 (= morton),
               src/tokenizers/_1D/zigzag_embedding1D.py:30-39,
               src/tokenizers/multiscale/multi_hilbert.py:74-84
  mixer        src/models/vit.py:268-273
  encoder      src/models/vit.py:197-206,241 -> torch:nn/modules/transformer.py:951-982
               (post-norm, relu, eps 1e-5), torch:nn/functional.py:5822-5833,6623-6637
  head         src/models/vit.py:289-292,303-319
  forward      src/models/vit.py:366-385 (2-D), :445-458 (1-D)
  loss         main.py:45-51
  step         src/training/train.py:153-167, main.py:288-289
"""
import math
from dataclasses import dataclass

import torch

from . import curves as _curves


# ----------------------------------------------------------------------------
# rounding-point mode
# ----------------------------------------------------------------------------
# The HIP path keeps parameters and every activation it writes to HBM in bf16 and accumulates in fp32.  With
# `rounding_points()` active the functions below round to bf16 at exactly those stores (and nowhere else), so that the
# remaining HIP-vs-oracle difference is fp32 summation order plus the bf16 rounding of the *gradients* the backward
# kernels store -- not the forward's rounding noise, which a 3e-2 tolerance against the fp32 oracle has to swallow.
# Store points followed (product file: sfcvit/functional.py): gathered pixels and projected tokens (tokenizer), LN outputs,
# the mixer's pre-GELU and GELU outputs, Q|K|V, the softmax numerators fed to P.V (csrc/attention_*.hip pack P to bf16
# and divide by the fp32 row sum of the unrounded numerators), the attention output, each residual sum before its
# LayerNorm, the ReLU hidden, the head's three GEMM outputs and its GELU, the logits.
# Backward: rounding is treated as the identity (straight-through), i.e. the gradients are exact fp32 gradients of the
# rounded forward, taken with respect to the bf16 parameter values.
_ROUNDING = False


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _r(t):
    return _RoundBF16.apply(t) if _ROUNDING else t


class rounding_points:
    """Context manager: `with vit_oracle.rounding_points(): logits = vit_oracle.forward(x, sd, cfg)`."""

    def __enter__(self):
        global _ROUNDING
        self._was, _ROUNDING = _ROUNDING, True
        return self

    def __exit__(self, *exc):
        global _ROUNDING
        _ROUNDING = self._was
        return False


def _pre_gelu_is_stored(m, n, k):
    """Whether the product's mixer stores the pre-activation in bf16 and applies GELU to the stored value (the shapes
    its persistent GEMM takes, sfcvit/functional.py:_fast_gemm_shape) or applies GELU to the fp32 accumulator."""
    return m >= 192 and n % 256 == 0 and k % 128 == 0 and k >= 256


@dataclass
class OracleConfig:
    tokenizer: str          # "hilbert1d" | "morton1d" | "raster1d" | "sfc"
    img_size: int
    patch_size: int         # pixels per token (1-D tokenizers) or group size g ("sfc")
    in_channels: int
    embed_dim: int
    depth: int
    n_heads: int
    mlp_dim: int
    num_classes: int
    variant: str = "1d"     # "1d" = VisionTransformer1D (with mixer), "2d" = VisionTransformer
    pre_patch_size: int = 1  # p of SFCEmbedding1D
    curve: str = "hilbert"   # curve of SFCEmbedding1D

    @property
    def n_patches(self):
        if self.tokenizer == "sfc":
            grid = self.img_size // self.pre_patch_size
            return grid * grid // self.patch_size
        return self.img_size * self.img_size // self.patch_size

    @property
    def input_dim(self):
        if self.tokenizer == "sfc":
            return self.in_channels * self.pre_patch_size ** 2 * self.patch_size
        return self.in_channels * self.patch_size

    @property
    def table_key(self):
        return {"hilbert1d": "hilbert_indices", "morton1d": "z_indices",
                "raster1d": None, "sfc": "sfc_indices"}[self.tokenizer]


# ----------------------------------------------------------------------------
# tokenizers
# ----------------------------------------------------------------------------
def curve_buffer(cfg):
    """The registered index buffer of the tokenizer, as the reference builds it."""
    if cfg.tokenizer == "hilbert1d":
        return torch.from_numpy(_curves.embed_and_prune_sfc("hilbert", cfg.img_size, cfg.img_size))
    if cfg.tokenizer == "morton1d":
        return torch.from_numpy(_curves.embed_and_prune_sfc("z", cfg.img_size, cfg.img_size))
    if cfg.tokenizer == "sfc":
        grid = cfg.img_size // cfg.pre_patch_size
        return torch.from_numpy(_curves.flat_table(cfg.curve, grid))
    return None


def tokens_1d(x, idx_rc, patch_size):
    """hilbert_embedding1D.py:36-40: gather along the curve, 'b c n -> b n c',
    reshape to [B, N, ps*C] (feature index = k*C + c)."""
    b, c = x.shape[0], x.shape[1]
    g = x[:, :, idx_rc[:, 0], idx_rc[:, 1]]          # [B, C, H*W]
    g = g.permute(0, 2, 1)                            # [B, H*W, C]
    return g.reshape(b, -1, patch_size * c)


def tokens_raster(x, patch_size):
    """zigzag_embedding1D.py:36-38."""
    b, c = x.shape[0], x.shape[1]
    g = x.flatten(2).transpose(1, 2)
    return g.reshape(b, -1, patch_size * c)


def tokens_sfc(x, flat_idx, p, g):
    """multi_hilbert.py:78-82: p x p pre-patches with feature order (p1 p2 c),
    permuted by the flat table, g consecutive pre-patches per token."""
    b, c, h, w = x.shape
    gh, gw = h // p, w // p
    t = x.reshape(b, c, gh, p, gw, p).permute(0, 2, 4, 3, 5, 1)   # b h w p1 p2 c
    t = t.reshape(b, gh * gw, p * p * c)
    t = t[:, flat_idx]
    return t.reshape(b, gh * gw // g, g * p * p * c)


def tokenize(x, sd, cfg, prefix="patch_embed."):
    if cfg.tokenizer in ("hilbert1d", "morton1d"):
        t = tokens_1d(x, sd[prefix + cfg.table_key], cfg.patch_size)
    elif cfg.tokenizer == "raster1d":
        t = tokens_raster(x, cfg.patch_size)
    elif cfg.tokenizer == "sfc":
        t = tokens_sfc(x, sd[prefix + "sfc_indices"], cfg.pre_patch_size, cfg.patch_size)
    else:
        raise ValueError(cfg.tokenizer)
    return _r(_r(t) @ sd[prefix + "proj.weight"].t() + sd[prefix + "proj.bias"])


def hierarchical_tokens(x, sd, img_size, patch_size_list, curve, prefix=""):
    """HierarchicalHilbertEmbedding / HierarchicalMortonEmbedding.forward
    (src/tokenizers/multiscale/multi_hilbert.py:31-40): level i = SFCEmbedding1D(pre_patch 2**i,
    group patch_size_list[i]); coarser levels are linearly resampled to the first level's length,
    concatenated on the feature axis and fused by a Linear.  `sd` keys: levels.{i}.sfc_indices,
    levels.{i}.proj.{weight,bias}, fusion.{weight,bias}."""
    outs = []
    for i, g in enumerate(patch_size_list):
        p = 2 ** i
        t = tokens_sfc(x, sd[f"{prefix}levels.{i}.sfc_indices"], p, g)
        outs.append(t @ sd[f"{prefix}levels.{i}.proj.weight"].t() + sd[f"{prefix}levels.{i}.proj.bias"])
    n_tokens = outs[0].shape[1]
    for i in range(1, len(outs)):
        outs[i] = torch.nn.functional.interpolate(outs[i].transpose(1, 2), size=n_tokens, mode="linear",
                                                  align_corners=False).transpose(1, 2)
    cat = torch.cat(outs, dim=-1)
    return cat @ sd[prefix + "fusion.weight"].t() + sd[prefix + "fusion.bias"]


def hierarchical_state(img_size, in_channels, patch_size_list, embed_dim, curve):
    """Formula-valued state of a hierarchical tokenizer (curve tables from the C oracle)."""
    from . import formula
    sd = {}
    for i, g in enumerate(patch_size_list):
        p = 2 ** i
        grid = img_size // p
        sd[f"levels.{i}.sfc_indices"] = torch.from_numpy(_curves.flat_table(curve, grid))
        sd[f"levels.{i}.proj.weight"] = formula.param_value(f"levels.{i}.proj.weight", (embed_dim, in_channels * p * p * g))
        sd[f"levels.{i}.proj.bias"] = formula.param_value(f"levels.{i}.proj.bias", (embed_dim,))
    d = embed_dim * len(patch_size_list)
    sd["fusion.weight"] = formula.param_value("fusion.weight", (d, d))
    sd["fusion.bias"] = formula.param_value("fusion.bias", (d,))
    return sd


def _case_table(curve, grid, seed=None):
    if curve == "randperm":       # RandomEmbedding.forward draws it per call (random_embedding.py:33)
        torch.manual_seed(seed)
        return torch.randperm(grid * grid)
    return torch.from_numpy(_curves.flat_table(curve, grid))


def tokenizer_case_state(args, kind):
    """Formula-valued state_dict of one oracle.cases.TOKENIZER_CASES entry, with the reference's key names."""
    from . import formula
    img = args[0]
    sd = {}
    if kind[0] == "grouped":
        _, curve, p, g, buf = kind
        dim = args[-1]
        if buf is not None:
            flat = _case_table(curve, img // p)
            if buf.endswith(":rc"):           # the _1D classes register (row, col) pairs
                sd[buf[:-3]] = torch.stack([flat // (img // p), flat % (img // p)], dim=1)
            else:
                sd[buf] = flat
        sd["proj.weight"] = formula.param_value("proj.weight", (dim, 3 * p * p * g))
        sd["proj.bias"] = formula.param_value("proj.bias", (dim,))
    elif kind[0] == "conv":
        _, curve, p = kind
        dim = args[-1]
        sd["proj.weight"] = formula.param_value("proj.weight", (dim, 3, p, p))
        sd["proj.bias"] = formula.param_value("proj.bias", (dim,))
    else:
        _, curve, plist, buf = kind
        dim = args[-1]
        for i, g in enumerate(plist):
            p = 2 ** i
            if buf is not None:
                sd[f"levels.{i}.{buf}"] = _case_table(curve, img // p)
            sd[f"levels.{i}.proj.weight"] = formula.param_value(f"levels.{i}.proj.weight", (dim, 3 * p * p * g))
            sd[f"levels.{i}.proj.bias"] = formula.param_value(f"levels.{i}.proj.bias", (dim,))
        d = dim * len(plist)
        sd["fusion.weight"] = formula.param_value("fusion.weight", (d, d))
        sd["fusion.bias"] = formula.param_value("fusion.bias", (d,))
    return sd


def tokenizer_case_forward(x, sd, args, kind, seed=None):
    """Forward of one TOKENIZER_CASES entry.  grouped: onion_embedding1D.py:55-76, multi_onion.py:90-103,
    multi_zigzag.py:85-97 (and the moore / peano twins of hilbert_embedding1D.py:30-44); conv:
    _2D/zigzag_embedding.py:23-30, hilbert_embedding.py:81-92, random_embedding.py:23-37; hier: multi_*.py:30-40."""
    img = args[0]
    if kind[0] == "grouped":
        _, curve, p, g, _ = kind
        flat = _case_table(curve, img // p)
        return tokens_sfc(x, flat, p, g) @ sd["proj.weight"].t() + sd["proj.bias"]
    if kind[0] == "conv":
        _, curve, p = kind
        y = torch.nn.functional.conv2d(x, sd["proj.weight"], sd["proj.bias"], stride=p)   # [B, D, gh, gw]
        y = y.flatten(2).transpose(1, 2)
        if curve == "raster":
            return y
        return y[:, _case_table(curve, img // p, seed)]
    _, curve, plist, _ = kind
    outs = []
    for i, g in enumerate(plist):
        p = 2 ** i
        t = tokens_sfc(x, _case_table(curve, img // p), p, g)
        outs.append(t @ sd[f"levels.{i}.proj.weight"].t() + sd[f"levels.{i}.proj.bias"])
    for i in range(1, len(outs)):
        outs[i] = torch.nn.functional.interpolate(outs[i].transpose(1, 2), size=outs[0].shape[1], mode="linear",
                                                  align_corners=False).transpose(1, 2)
    return torch.cat(outs, dim=-1) @ sd["fusion.weight"].t() + sd["fusion.bias"]


# ----------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------
def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)      # biased
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def mixer_block(x, sd, prefix="mlp_mixer."):
    """vit.py:272: x + channel_mix(channel_mix_ln(x)); the token-mix branch is
    commented out in the reference (:269-271) and its parameters are unused."""
    z = _r(layer_norm(x, sd[prefix + "channel_mix_ln.weight"], sd[prefix + "channel_mix_ln.bias"]))
    w1 = sd[prefix + "channel_mix.0.weight"]
    u = z @ w1.t() + sd[prefix + "channel_mix.0.bias"]
    if _pre_gelu_is_stored(z.numel() // z.shape[-1], w1.shape[0], w1.shape[1]):
        u = _r(u)
    h = _r(gelu_erf(u))
    return _r(x + h @ sd[prefix + "channel_mix.2.weight"].t() + sd[prefix + "channel_mix.2.bias"])


def attention(x, w_in, b_in, w_out, b_out, n_heads):
    b, n, d = x.shape
    hd = d // n_heads
    qkv = _r(x @ w_in.t() + b_in)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    q = q.reshape(b, n, n_heads, hd).transpose(1, 2)
    k = k.reshape(b, n, n_heads, hd).transpose(1, 2)
    v = v.reshape(b, n, n_heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if _ROUNDING:
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        o = (_r(e) @ v) / e.sum(dim=-1, keepdim=True)
    else:
        o = torch.softmax(s, dim=-1) @ v
    o = _r(o.transpose(1, 2).reshape(b, n, d))
    return o @ w_out.t() + b_out


def encoder_layer(x, sd, prefix, n_heads):
    """Post-norm layer, eval mode (dropout off): torch:nn/modules/transformer.py:951-958."""
    a = attention(x, sd[prefix + "self_attn.in_proj_weight"], sd[prefix + "self_attn.in_proj_bias"],
                  sd[prefix + "self_attn.out_proj.weight"], sd[prefix + "self_attn.out_proj.bias"],
                  n_heads)
    x = _r(layer_norm(_r(x + a), sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"]))
    f = _r(torch.relu(x @ sd[prefix + "linear1.weight"].t() + sd[prefix + "linear1.bias"]))
    f = f @ sd[prefix + "linear2.weight"].t() + sd[prefix + "linear2.bias"]
    return _r(layer_norm(_r(x + f), sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"]))


def head(x, sd, prefix="mlp_head."):
    """vit.py:303-319 with n_layers=2: LN -> FactorisedLinear -> GELU -> (Dropout) -> Linear."""
    z = _r(layer_norm(x, sd[prefix + "0.weight"], sd[prefix + "0.bias"]))
    h = _r(torch.einsum("bnd,rd->bnr", z, sd[prefix + "1.W_emb"]))
    y = _r(torch.einsum("bnr,onr->bo", h, sd[prefix + "1.W_seq"]))
    y = _r(gelu_erf(y))
    return _r(y @ sd[prefix + "4.weight"].t() + sd[prefix + "4.bias"])


def forward(x, sd, cfg, return_intermediates=False):
    inter = {}
    if _ROUNDING:                                   # the product's parameters are bf16 tensors
        sd = {k: (_r(v) if torch.is_floating_point(v) else v) for k, v in sd.items()}
    t = tokenize(x, sd, cfg)
    inter["tokens"] = t
    if cfg.variant == "1d":
        t = mixer_block(t, sd)
        inter["mixer"] = t
    for layer in range(cfg.depth):
        t = encoder_layer(t, sd, f"encoder.transformer.layers.{layer}.", cfg.n_heads)
    inter["encoded"] = t
    logits = head(t, sd)
    return (logits, inter) if return_intermediates else logits


def soft_target_ce(logits, targets):
    """main.py:49-51."""
    return -(targets * torch.log_softmax(logits, dim=-1)).sum(-1).mean()


# ----------------------------------------------------------------------------
# state
# ----------------------------------------------------------------------------
def state_shapes(cfg):
    """Reference state_dict keys -> shapes, duplicates included."""
    d, n, f, c = cfg.embed_dim, cfg.n_patches, cfg.mlp_dim, cfg.num_classes
    s = {}
    pe = {}
    if cfg.tokenizer in ("hilbert1d", "morton1d"):
        pe[cfg.table_key] = (cfg.img_size ** 2, 2)
    elif cfg.tokenizer == "sfc":
        pe["sfc_indices"] = ((cfg.img_size // cfg.pre_patch_size) ** 2,)
    pe["proj.weight"] = (d, cfg.input_dim)
    pe["proj.bias"] = (d,)
    for k, v in pe.items():
        s["patch_embed." + k] = v
    if cfg.variant == "1d":
        m = "mlp_mixer."
        s[m + "token_mix_ln.weight"] = (d,)
        s[m + "token_mix_ln.bias"] = (d,)
        s[m + "channel_mix_ln.weight"] = (d,)
        s[m + "channel_mix_ln.bias"] = (d,)
        s[m + "token_mix.0.weight"] = (2 * d, n)
        s[m + "token_mix.0.bias"] = (2 * d,)
        s[m + "token_mix.2.weight"] = (n, 2 * d)
        s[m + "token_mix.2.bias"] = (n,)
        s[m + "channel_mix.0.weight"] = (2 * d, d)
        s[m + "channel_mix.0.bias"] = (2 * d,)
        s[m + "channel_mix.2.weight"] = (d, 2 * d)
        s[m + "channel_mix.2.bias"] = (d,)
    for layer in range(cfg.depth):
        p = f"encoder.transformer.layers.{layer}."
        s[p + "self_attn.in_proj_weight"] = (3 * d, d)
        s[p + "self_attn.in_proj_bias"] = (3 * d,)
        s[p + "self_attn.out_proj.weight"] = (d, d)
        s[p + "self_attn.out_proj.bias"] = (d,)
        s[p + "linear1.weight"] = (f, d)
        s[p + "linear1.bias"] = (f,)
        s[p + "linear2.weight"] = (d, f)
        s[p + "linear2.bias"] = (d,)
        s[p + "norm1.weight"] = (d,)
        s[p + "norm1.bias"] = (d,)
        s[p + "norm2.weight"] = (d,)
        s[p + "norm2.bias"] = (d,)
    for k, v in pe.items():
        s["encoder.to_patch_embedding." + k] = v       # vit.py:221 (same tensors)
    s["mlp_head.0.weight"] = (d,)
    s["mlp_head.0.bias"] = (d,)
    s["mlp_head.1.W_emb"] = (64, d)
    s["mlp_head.1.W_seq"] = (2 * d, n, 64)
    s["mlp_head.4.weight"] = (c, 2 * d)
    s["mlp_head.4.bias"] = (c,)
    return s


def formula_state(cfg):
    """Formula-valued fp32 state for `cfg` (curve buffers from the C oracle)."""
    from . import formula
    buf = curve_buffer(cfg)
    sd = {}
    for k, shape in state_shapes(cfg).items():
        if k.startswith("encoder.to_patch_embedding."):
            continue
        if cfg.table_key and k.endswith(cfg.table_key):
            sd[k] = buf.clone()
        else:
            sd[k] = formula.param_value(k, shape)
    for k in list(sd):
        if k.startswith("patch_embed."):
            sd["encoder.to_patch_embedding." + k[len("patch_embed."):]] = sd[k]
    return sd


def random_state(cfg, seed=0, dtype=torch.float32):
    """Random-init fp32 state of the right shapes (for CPU-baseline timing only;
    the init distribution is not the reference's)."""
    g = torch.Generator().manual_seed(seed)
    buf = curve_buffer(cfg)
    sd = {}
    for k, shape in state_shapes(cfg).items():
        if k.startswith("encoder.to_patch_embedding."):
            continue
        if cfg.table_key and k.endswith(cfg.table_key):
            sd[k] = buf.clone()
        elif k.endswith("weight") and len(shape) == 1:
            sd[k] = torch.ones(shape, dtype=dtype)
        elif len(shape) == 1:
            sd[k] = torch.zeros(shape, dtype=dtype)
        else:
            fan_in = 1
            for v in shape[1:]:
                fan_in *= v
            sd[k] = (torch.randn(shape, generator=g, dtype=dtype) / math.sqrt(fan_in))
    return sd


UNUSED_PREFIXES = ("mlp_mixer.token_mix.", "mlp_mixer.token_mix_ln.")


def trainable(sd):
    """Leaf fp32 tensors that receive a gradient (token-mix params never do, vit.py:269-272)."""
    out = {}
    for k, v in sd.items():
        if not torch.is_floating_point(v) or k.startswith("encoder.to_patch_embedding."):
            continue
        out[k] = v
    return out


def train_step(x, targets, sd, cfg, opt, clip=1.0):
    """One step with the semantics of train.py:153-167 (no AMP on CPU, fp32):
    zero_grad -> forward -> soft-target CE -> backward -> clip_grad_norm_(1.0) ->
    AdamW.step.  `sd` holds leaf tensors with requires_grad=True for trainable keys."""
    opt.zero_grad()
    logits = forward(x, sd, cfg)
    loss = soft_target_ce(logits, targets)
    loss.backward()
    params = [p for p in opt.param_groups[0]["params"]]
    torch.nn.utils.clip_grad_norm_(params, clip, foreach=False)
    opt.step()
    return loss.detach(), logits.detach()


# ----------------------------------------------------------------------------
# altvit (src/models/altvit.py): SimpleViT / HilbertViT
# ----------------------------------------------------------------------------
def altvit_patches(x, p):
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (altvit.py:88-90,176-177)."""
    b, c, h, w = x.shape
    return x.reshape(b, c, h // p, p, w // p, p).permute(0, 2, 4, 3, 5, 1).reshape(b, (h // p) * (w // p), p * p * c)


def altvit_pos_embedding(kind, grid, dim, T=4, h_param=3.0):
    """SimpleViT: posemb_sincos_1d (altvit.py:16-41); HilbertViT: the Hilbert-index encoding (altvit.py:229-251)."""
    n = grid * grid
    if kind == "simple":
        pe = torch.zeros(n, dim)
        position = torch.arange(n, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-math.log(10000.0) / dim))
        pe[:, 0::2] = torch.sin(position * div)
        pe[:, 1::2] = torch.cos(position * div)
        return pe
    pos = torch.from_numpy(_curves.flat_table("hilbert", grid)).to(torch.float32).unsqueeze(1)
    i_ar = torch.arange(dim // 2, dtype=torch.float32).unsqueeze(0)
    two_pi = 2 * math.pi
    arg = (2.0 * i_ar * grid ** 2 * pos * two_pi) / (T * n * dim) + h_param * (2.0 * i_ar * pos * two_pi) / dim
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)


def altvit_forward(x, sd, kind, patch, heads, depth):
    """kind 'simple' | 'hilbert' (altvit.py:196-205, 260-268); pre-norm blocks altvit.py:116-160."""
    pre = "to_patch_embedding."
    names = ("1.", "2.", "3.") if kind == "simple" else ("layernorm1.", "linear.", "layernorm2.")
    t = altvit_patches(x, patch)
    if kind == "hilbert":
        grid = x.shape[-1] // patch
        t = t[:, torch.from_numpy(_curves.flat_table("hilbert", grid))]      # private recursion == src/curves order (2^k grids)
    t = layer_norm(t, sd[pre + names[0] + "weight"], sd[pre + names[0] + "bias"])
    t = t @ sd[pre + names[1] + "weight"].t() + sd[pre + names[1] + "bias"]
    t = layer_norm(t, sd[pre + names[2] + "weight"], sd[pre + names[2] + "bias"])
    t = t + sd["pos_embedding"]
    B, N, D = t.shape
    for i in range(depth):
        a, f = f"transformer.layers.{i}.0.", f"transformer.layers.{i}.1.net."
        z = layer_norm(t, sd[a + "norm.weight"], sd[a + "norm.bias"])
        qkv = z @ sd[a + "to_qkv.weight"].t()
        inner = qkv.shape[-1] // 3
        hd = inner // heads
        q, k, v = (u.reshape(B, N, heads, hd).transpose(1, 2) for u in qkv.split(inner, dim=-1))
        p = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1)
        o = (p @ v).transpose(1, 2).reshape(B, N, inner)
        t = o @ sd[a + "to_out.weight"].t() + t
        z = layer_norm(t, sd[f + "0.weight"], sd[f + "0.bias"])
        z = gelu_erf(z @ sd[f + "1.weight"].t() + sd[f + "1.bias"])
        t = z @ sd[f + "3.weight"].t() + sd[f + "3.bias"] + t
    t = layer_norm(t, sd["transformer.norm.weight"], sd["transformer.norm.bias"])
    return t.mean(dim=1) @ sd["linear_head.weight"].t() + sd["linear_head.bias"]


def altvit_state(kind, image, patch, dim, depth, heads, mlp, classes, dim_head=64):
    """Formula-valued state_dict with the reference's keys."""
    from . import formula
    names = ("1", "2", "3") if kind == "simple" else ("layernorm1", "linear", "layernorm2")
    pd, inner = 3 * patch * patch, heads * dim_head
    shapes = {f"to_patch_embedding.{names[0]}.weight": (pd,), f"to_patch_embedding.{names[0]}.bias": (pd,),
              f"to_patch_embedding.{names[1]}.weight": (dim, pd), f"to_patch_embedding.{names[1]}.bias": (dim,),
              f"to_patch_embedding.{names[2]}.weight": (dim,), f"to_patch_embedding.{names[2]}.bias": (dim,),
              "transformer.norm.weight": (dim,), "transformer.norm.bias": (dim,),
              "linear_head.weight": (classes, dim), "linear_head.bias": (classes,)}
    for i in range(depth):
        a, f = f"transformer.layers.{i}.0.", f"transformer.layers.{i}.1.net."
        shapes.update({a + "norm.weight": (dim,), a + "norm.bias": (dim,), a + "to_qkv.weight": (3 * inner, dim),
                       a + "to_out.weight": (dim, inner), f + "0.weight": (dim,), f + "0.bias": (dim,),
                       f + "1.weight": (mlp, dim), f + "1.bias": (mlp,), f + "3.weight": (dim, mlp), f + "3.bias": (dim,)})
    sd = {k: formula.param_value(k, shp) for k, shp in shapes.items()}
    sd["pos_embedding"] = altvit_pos_embedding(kind, image // patch, dim)
    return sd
