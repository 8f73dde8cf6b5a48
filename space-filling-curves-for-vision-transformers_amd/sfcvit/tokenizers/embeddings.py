"""Curve-ordered tokenizers on the fused gather + patchify + projection HIP kernel.

Same constructor signatures, attributes (`n_patches`, `input_dim`, `embed_dim`,
`proj`) and state_dict keys (index buffers included) as the reference classes:
  HilbertEmbedding1D     src/tokenizers/_1D/hilbert_embedding1D.py:9-44
  MortonEmbedding1D      src/tokenizers/_1D/morton_embedding1D.py:9-44
  MooreEmbedding1D / PeanoEmbedding1D   src/tokenizers/_1D/{moore,peano}_embedding1D.py
  RasterScan1DEmbedding  src/tokenizers/_1D/zigzag_embedding1D.py:5-39
  OnionEmbedding1D       src/tokenizers/_1D/onion_embedding1D.py:10-76 (spiral; no registered buffer)
  SFCEmbedding1D         src/tokenizers/multiscale/multi_hilbert.py:43-84
  OnionGroupedEmbedding1D / RasterScan1DGroupedEmbedding   multiscale/multi_onion.py:46-103, multi_zigzag.py:55-97
  ZigzagEmbedding / HilbertEmbedding / RandomEmbedding     src/tokenizers/_2D/*.py (Conv2d-weight tokenizers)
The registered int64 buffer stays the source of truth (it is part of the
state_dict); the kernel's per-token pixel table is derived from it on the host by
sfcvit_pixel_table and cached on the device.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import functional as F
from .._lib import lib, check
from ..curves.space_filling_curves import (curve_table, curve_table_rc, hilbert_curve, hilbert_t_curve, moore_curve,
                                           peano_curve, spiral_curve, z_curve)
from .base_patch_embedding import BasePatchEmbedding


def _pixel_table(flat, img, p, g):
    flat = np.ascontiguousarray(flat, dtype=np.int32)
    grid = img // p
    out = np.empty((grid * grid // g, g * p * p), dtype=np.int32)
    check(lib.sfcvit_pixel_table(flat.ctypes.data_as(ctypes.c_void_p), img, p, g,
                                 out.ctypes.data_as(ctypes.c_void_p)), "sfcvit_pixel_table")
    return out


class _FusedTokenizer(BasePatchEmbedding):
    """Holds `proj` and the device copy of the pixel table; subclasses say how the
    flat curve table is obtained from their registered buffer."""

    def _setup(self, img_size, pre_patch, group, in_channels, embed_dim):
        self._geom = (img_size, pre_patch, group)
        self._pix = None
        self._pix_key = None
        self._desc = None
        self._order = None
        self.proj = nn.Linear(in_channels * pre_patch * pre_patch * group, embed_dim)

    def _flat_table(self):          # -> 1-D integer array of length grid*grid, or None for raster
        raise NotImplementedError

    def _key_buffer(self):          # the long-lived tensor the table derives from (None: raster order); subclasses whose
        return self._flat_table()   # _flat_table() builds a temporary override this with the registered buffer

    def _pix_table(self, device):
        # keyed on the registered buffer itself (storage, version counter, device): a hit costs three attribute reads
        # and no kernel; load_state_dict / .to() / an in-place edit of the buffer change the key.  The flat table
        # (a derived temporary for the [n, 2] row/col buffers) is only built on a miss.
        reg = self._key_buffer()
        key = (None if reg is None else (reg.data_ptr(), reg._version, str(reg.device)), str(device))
        if self._pix is None or self._pix_key != key:
            buf = self._flat_table()
            img, p, g = self._geom
            grid = img // p
            flat = np.arange(grid * grid, dtype=np.int32) if buf is None else buf.detach().cpu().numpy()
            pix_host = _pixel_table(flat, img, p, g)
            self._pix = torch.from_numpy(pix_host).to(device)
            # are the tokens 16 x 16 pixel tiles (Hilbert / Z at 256 pixels per token) or 256-pixel strips (raster)?  Then
            # the coalesced kernels apply (csrc/patch_embed_tiled.hip); decided once, on the host, with the table
            from .. import ops
            self._desc = ops.tile_descriptor(pix_host, img, device) if torch.device(device).type == "cuda" else None
            # ... and the pairing order of the gather kernels (tokens by lowest pixel offset): kept with the table it
            # belongs to, not in a cache keyed by the table's address
            self._order = torch.from_numpy(ops.gather_order(pix_host)).to(device) if self._static_order else None
            self._pix_key = key
        return self._pix

    def _apply(self, fn, *args, **kwargs):
        """`.to(device)` / `.cuda()`: build the device tables right away, so that a model that is moved and then wrapped in
        torch.compile (the reference's order, main.py:252-284) never has to build them inside a traced forward."""
        out = super()._apply(fn, *args, **kwargs)
        dev = self.proj.weight.device
        if dev.type == "cuda" and getattr(self, "_geom", None) is not None and self._static_order:
            self._pix_table(dev)
        return out

    _static_order = True            # False: the token order changes from call to call (RandomEmbedding): nothing to prebuild

    def forward(self, x):
        img = self._geom[0]
        if x.dim() != 4 or x.shape[2] != img or x.shape[3] != img:
            raise ValueError(f"expected [B, C, {img}, {img}] input, got {tuple(x.shape)}")
        if torch.compiler.is_compiling():
            # the cache key reads data_ptr() / _version, which Dynamo cannot trace: the tables exist (built by .to(device))
            if self._pix is None or self._pix.device != x.device:
                raise RuntimeError("tokenizer device tables are missing: move the model to the device (model.to(device)) "
                                   "before wrapping it in torch.compile")
            pix = self._pix
        else:
            pix = self._pix_table(x.device)
        return F.patch_embed(x, pix, self.proj.weight, self.proj.bias, self._desc, self._order)


class _Curve1D(_FusedTokenizer):
    _curve = None
    _buffer = None

    def __init__(self, img_size, patch_size, in_channels, embed_dim):
        super().__init__()
        num_pixels = img_size * img_size
        assert num_pixels % patch_size == 0, "Image must be divisible into 1D patches"
        self.n_patches = num_pixels // patch_size
        self.input_dim = in_channels * patch_size
        self.register_buffer(self._buffer, torch.from_numpy(curve_table_rc(self._curve, img_size)).long())
        self.embed_dim = embed_dim
        self._img_size = img_size
        self._setup(img_size, 1, patch_size, in_channels, embed_dim)

    def _key_buffer(self):
        return getattr(self, self._buffer)

    def _flat_table(self):
        rc = getattr(self, self._buffer)
        return rc[:, 0] * self._img_size + rc[:, 1]


class HilbertEmbedding1D(_Curve1D):
    _curve, _buffer = hilbert_curve, "hilbert_indices"


class MortonEmbedding1D(_Curve1D):
    _curve, _buffer = z_curve, "z_indices"


class MooreEmbedding1D(_Curve1D):
    _curve, _buffer = moore_curve, "moore_indices"


class PeanoEmbedding1D(_Curve1D):
    _curve, _buffer = peano_curve, "peano_indices"


class RasterScan1DEmbedding(_FusedTokenizer):
    def __init__(self, img_size, patch_size, in_channels, embed_dim):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.embed_dim = embed_dim
        num_pixels = img_size * img_size
        assert num_pixels % patch_size == 0, "Image must be divisible into 1D patches"
        self.n_patches = num_pixels // patch_size
        self.input_dim = patch_size * in_channels
        self._setup(img_size, 1, patch_size, in_channels, embed_dim)

    def _flat_table(self):
        return None


class OnionEmbedding1D(RasterScan1DEmbedding):
    """Pixels in inward-spiral order from the bottom-left corner.  Like the reference class it registers no
    index buffer (the walk is recomputed from the input size there), so state_dict holds `proj.*` only."""

    def __init__(self, img_size, patch_size, in_channels, embed_dim):
        super().__init__(img_size, patch_size, in_channels, embed_dim)
        self._spiral = torch.from_numpy(curve_table(spiral_curve, img_size))

    def _flat_table(self):
        return self._spiral


class SFCEmbedding1D(_FusedTokenizer):
    _buffer = "sfc_indices"

    def __init__(self, img_size, pre_patch_size, group_patch_size, in_channels, embed_dim,
                 curve_fn=hilbert_curve):
        super().__init__()
        assert img_size % pre_patch_size == 0, "Image size must be divisible by pre_patch_size"
        self.img_size = img_size
        self.pre_patch_size = pre_patch_size
        self.group_patch_size = group_patch_size
        self.in_channels = in_channels
        self.embed_dim = embed_dim
        self.curve_fn = curve_fn
        self.grid_size = img_size // pre_patch_size
        self.n_pre_patches = self.grid_size * self.grid_size
        self.n_final_patches = self.n_pre_patches // group_patch_size
        self.n_patches = self.n_final_patches       # what VisionTransformer{,1D} reads (vit.py:354,422)
        self.pre_patch_dim = in_channels * pre_patch_size * pre_patch_size
        self.input_dim = self.pre_patch_dim * group_patch_size
        if self._buffer is not None:
            self.register_buffer(self._buffer, torch.from_numpy(curve_table(curve_fn, self.grid_size)).long())
        self._setup(img_size, pre_patch_size, group_patch_size, in_channels, embed_dim)

    def _flat_table(self):
        return None if self._buffer is None else getattr(self, self._buffer)


class OnionGroupedEmbedding1D(SFCEmbedding1D):
    """multiscale/multi_onion.py:46-103 (named OnionEmbedding1D there; buffer `onion_indices`)."""
    _buffer = "onion_indices"

    def __init__(self, img_size, pre_patch_size, group_patch_size, in_channels, embed_dim):
        super().__init__(img_size, pre_patch_size, group_patch_size, in_channels, embed_dim, spiral_curve)


class RasterScan1DGroupedEmbedding(SFCEmbedding1D):
    """multiscale/multi_zigzag.py:55-97: pre-patches in raster order, no buffer."""
    _buffer = None

    def __init__(self, img_size, pre_patch_size, group_patch_size, in_channels, embed_dim):
        super().__init__(img_size, pre_patch_size, group_patch_size, in_channels, embed_dim, None)


class _Conv2dTokenizer(_FusedTokenizer):
    """p x p patches projected by an nn.Conv2d(kernel = stride = p) weight [D, C, p, p]; the fused kernel wants
    the features of a patch ordered (p1, p2, c), so the weight is viewed through a permute (1.2 MB at ViT-B,
    differentiable) instead of running a convolution."""

    def __init__(self, img_size, patch_size, in_channels, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.img_size = img_size
        self.embed_dim = embed_dim
        self.grid_size = img_size // patch_size
        self.n_patches = self.grid_size ** 2
        self._geom = (img_size, patch_size, 1)
        self._pix = None
        self._pix_key = None

    def _flat_table(self):
        return None

    def forward(self, x):
        img = self._geom[0]
        if x.dim() != 4 or x.shape[2] != img or x.shape[3] != img:
            raise ValueError(f"expected [B, C, {img}, {img}] input, got {tuple(x.shape)}")
        w = self.proj.weight
        w2 = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
        return F.patch_embed(x, self._pix_table(x.device), w2, self.proj.bias)


class ZigzagEmbedding(_Conv2dTokenizer):
    """src/tokenizers/_2D/zigzag_embedding.py:5-30: the standard ViT patch embedding, raster token order."""


class HilbertEmbedding(_Conv2dTokenizer):
    """src/tokenizers/_2D/hilbert_embedding.py:9-92: ViT patches visited along a Hilbert curve (power-of-two
    grids; `hilbert_indices` is a plain attribute there, not a buffer, and stays one here)."""

    def __init__(self, img_size, patch_size, in_channels, embed_dim):
        super().__init__(img_size, patch_size, in_channels, embed_dim)
        self.hilbert_indices = torch.from_numpy(curve_table(hilbert_t_curve, self.grid_size)).long()

    def _flat_table(self):
        return self.hilbert_indices


class RandomEmbedding(_Conv2dTokenizer):
    """src/tokenizers/_2D/random_embedding.py:6-37: a fresh torch.randperm(N) of the patches on every call
    (drawn from torch's CPU generator, as in the reference)."""
    _static_order = False

    def forward(self, x):
        self._perm = torch.randperm(self.n_patches)
        self._pix_key = None                     # a new order every call: never reuse the cached pixel table
        return super().forward(x)

    def _flat_table(self):
        return self._perm
