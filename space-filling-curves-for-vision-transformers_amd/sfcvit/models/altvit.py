"""The reference's second model family (src/models/altvit.py: SimpleViT / HilbertViT -- pre-norm blocks, bias-free
attention projections, sin-cos or Hilbert-index positional embedding, mean pooling) on the HIP kernels.

Same constructors, module tree and state_dict keys as the reference classes (fixtures: tests/golden/altvit.json).
LayerNorm, every Linear (GELU fused into the first FFN GEMM), and the attention core run through sfcvit.functional;
patch extraction / Hilbert reordering, residual adds and the mean pool are torch plumbing -- this family is not the
benchmark path."""
import math

import torch
from torch import nn

from .. import functional as F
from .. import ops
from ..curves.space_filling_curves import curve_table, hilbert_curve


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def posemb_sincos_1d(n_pos, dim, temperature: float = 10000.0, dtype=torch.float32):
    """Sinusoidal table of the reference's SimpleViT (altvit.py:16-41): column 2i = sin(p w_i), column 2i + 1 = cos(p w_i),
    w_i = temperature^(-2i / dim).  Angles are formed in float64 and the table is rounded once."""
    if dim % 2:
        raise ValueError("posemb_sincos_1d: dim must be even")
    freq = torch.pow(torch.tensor(float(temperature), dtype=torch.float64),
                     -torch.arange(0, dim, 2, dtype=torch.float64) / dim)                  # [dim / 2]
    angle = torch.outer(torch.arange(n_pos, dtype=torch.float64), freq)                      # [n_pos, dim / 2]
    return torch.stack((angle.sin(), angle.cos()), dim=-1).reshape(n_pos, dim).to(dtype)    # interleave sin / cos


def _patches(x, p1, p2):
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)'"""
    b, c, hh, ww = x.shape
    gh, gw = hh // p1, ww // p2
    return x.reshape(b, c, gh, p1, gw, p2).permute(0, 2, 4, 3, 5, 1).reshape(b, gh * gw, p1 * p2 * c)


class _Patchify(nn.Module):
    """Stands where einops' Rearrange sits in SimpleViT.to_patch_embedding (index 0, no parameters)."""

    def __init__(self, p1, p2):
        super().__init__()
        self.p1, self.p2 = p1, p2

    def forward(self, x):
        return _patches(x, self.p1, self.p2)


class HilbertPatchEmbedding(nn.Module):
    """altvit.py:46-99: patches in Hilbert order -> LayerNorm -> Linear -> LayerNorm."""

    def __init__(self, *, image_size, patch_size, channels, dim):
        super().__init__()
        image_height, image_width = pair(image_size)
        patch_height, patch_width = pair(patch_size)
        self.grid_h = image_height // patch_height
        self.grid_w = image_width // patch_width
        assert self.grid_h == self.grid_w and (self.grid_h & (self.grid_h - 1)) == 0, \
            "Hilbert curve requires square grid size that is a power of 2."
        patch_dim = channels * patch_height * patch_width
        self.patch_height, self.patch_width, self.channels = patch_height, patch_width, channels
        self.layernorm1 = nn.LayerNorm(patch_dim)
        self.linear = nn.Linear(patch_dim, dim)
        self.layernorm2 = nn.LayerNorm(dim)
        # the class's private integer recursion equals the src/curves Hilbert order on power-of-two grids
        self.hilbert_indices = torch.from_numpy(curve_table(hilbert_curve, self.grid_h)).long()

    def forward(self, x):
        x = _patches(x, self.patch_height, self.patch_width)[:, self.hilbert_indices.to(x.device)]
        x = F.layer_norm(x, self.layernorm1.weight, self.layernorm1.bias, self.layernorm1.eps)
        x = F.linear(x, self.linear.weight, self.linear.bias)
        return F.layer_norm(x, self.layernorm2.weight, self.layernorm2.bias, self.layernorm2.eps)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, dim))

    def forward(self, x):
        ln, l1, _, l2 = self.net
        x = F.layer_norm(x, ln.weight, ln.bias, ln.eps)
        x = F.linear(x, l1.weight, l1.bias, act=ops.ACT_GELU)
        return F.linear(x, l2.weight, l2.bias)


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.attend = nn.Softmax(dim=-1)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)

    def forward(self, x):
        x = F.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        o = F.attention(F.linear(x, self.to_qkv.weight), self.heads)
        return F.linear(o, self.to_out.weight)


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([Attention(dim, heads=heads, dim_head=dim_head), FeedForward(dim, mlp_dim)]))

    def forward(self, x):
        for attn, ff in self.layers:
            x = attn(x) + x
            x = ff(x) + x
        return F.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)


class _PooledViT(nn.Module):
    def _encode(self, x):
        x = x + self.pos_embedding.to(x.device, dtype=x.dtype)
        x = self.transformer(x)
        x = x.float().mean(dim=1).to(x.dtype)
        x = self.to_latent(x)
        return F.linear(x, self.linear_head.weight, self.linear_head.bias)


class SimpleViT(_PooledViT):
    """altvit.py:163-205."""

    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, channels=3, dim_head=64):
        super().__init__()
        image_height, image_width = pair(image_size)
        patch_height, patch_width = pair(patch_size)
        assert image_height % patch_height == 0 and image_width % patch_width == 0, \
            'Image dimensions must be divisible by the patch size.'
        patch_dim = channels * patch_height * patch_width
        self.to_patch_embedding = nn.Sequential(_Patchify(patch_height, patch_width), nn.LayerNorm(patch_dim),
                                                nn.Linear(patch_dim, dim), nn.LayerNorm(dim))
        self.posemb = posemb_sincos_1d(n_pos=(image_height // patch_height) * (image_width // patch_width), dim=dim)
        self.register_buffer("pos_embedding", self.posemb)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim)
        self.pool = "mean"
        self.to_latent = nn.Identity()
        self.linear_head = nn.Linear(dim, num_classes)

    def forward(self, img):
        pat, ln1, lin, ln2 = self.to_patch_embedding
        x = F.layer_norm(pat(img), ln1.weight, ln1.bias, ln1.eps)
        x = F.layer_norm(F.linear(x, lin.weight, lin.bias), ln2.weight, ln2.bias, ln2.eps)
        return self._encode(x)


class HilbertViT(_PooledViT):
    """altvit.py:208-268: Hilbert-ordered patches and a positional embedding built from the Hilbert indices."""

    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, channels=3, dim_head=64,
                 T=4, h_param=3.0):
        super().__init__()
        image_height, image_width = pair(image_size)
        patch_height, patch_width = pair(patch_size)
        assert image_height % patch_height == 0 and image_width % patch_width == 0, \
            'Image dimensions must be divisible by the patch size.'
        self.grid_h = image_height // patch_height
        self.grid_w = image_width // patch_width
        assert self.grid_h == self.grid_w and (self.grid_h & (self.grid_h - 1)) == 0, \
            "Hilbert embedding requires square grid size that is a power of 2."
        self.to_patch_embedding = HilbertPatchEmbedding(image_size=image_size, patch_size=patch_size, channels=channels,
                                                        dim=dim)
        hilbert_indices = self.to_patch_embedding.hilbert_indices
        n = hilbert_indices.numel()
        N = int(math.sqrt(n))
        assert N * N == n, "Hilbert indices must form a square grid."
        assert dim % 2 == 0, "Feature dimension must be even."
        # altvit.py:229-251 adds two phase terms per (position h, frequency i): 2 pi (2i) h N^2 / (T n dim) and
        # 2 pi h_param (2i) h / dim.  With n = N^2 they are one product, angle = h * i * (4 pi / dim) * (1 / T + h_param);
        # first half of the feature axis = sin, second half = cos.
        rate = (4.0 * math.pi / dim) * (1.0 / T + h_param) * torch.arange(dim // 2, dtype=torch.float64)
        angle = hilbert_indices.to(torch.float64).reshape(-1, 1) * rate.reshape(1, -1)
        self.register_buffer("pos_embedding", torch.cat((angle.sin(), angle.cos()), dim=1).to(torch.float32))
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim)
        self.pool = "mean"
        self.to_latent = nn.Identity()
        self.linear_head = nn.Linear(dim, num_classes)

    def forward(self, img):
        return self._encode(self.to_patch_embedding(img))
