"""Hierarchical (multi-scale) curve tokenizers: several SFCEmbedding1D levels, resampled to a common
length, concatenated on the feature axis and fused by a Linear
(reference: src/tokenizers/multiscale/multi_hilbert.py:9-40, multi_morton.py:9-40; what the reference's
main.py:269-274 instantiates).  When every level has the same token count -- the reference's own configuration
[16, 4, 1]: the resampling is then the identity -- the whole forward is ONE kernel (csrc/hier_tokenizer.hip: gather,
level projections, concatenation in LDS, fusion GEMM).  Otherwise every level and the fusion run on their own HIP
kernels, with the reference's linear resampling + concatenation as one more (sfcvit_hier_resample_concat)."""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as F
from ..curves.space_filling_curves import hilbert_curve, moore_curve, peano_curve, z_curve
from .embeddings import OnionGroupedEmbedding1D, RasterScan1DGroupedEmbedding, SFCEmbedding1D


class _Hierarchical(nn.Module):
    _default_curve = hilbert_curve

    def __init__(self, img_size, in_channels, patch_size_list, embed_dim, curve_fn=None):
        super().__init__()
        curve_fn = curve_fn or type(self)._default_curve
        self.levels = nn.ModuleList()
        pre_patch_size, pre_patch_list = 1, []
        for patch_size in patch_size_list:
            self.levels.append(self._level(img_size, pre_patch_size, patch_size, in_channels, embed_dim, curve_fn))
            pre_patch_list.append(pre_patch_size)
            pre_patch_size *= 2
        self.patch_list = [int(((img_size // pre) // np.sqrt(ps)) ** 2) for pre, ps in zip(pre_patch_list, patch_size_list)]
        self.embed_dim = embed_dim * len(patch_size_list)
        self.depth = len(patch_size_list)
        self.n_patches = self.patch_list[0]
        self.fusion = nn.Linear(self.embed_dim, self.embed_dim)

    @staticmethod
    def _level(img_size, pre_patch_size, patch_size, in_channels, embed_dim, curve_fn):
        return SFCEmbedding1D(img_size, pre_patch_size, patch_size, in_channels, embed_dim, curve_fn)

    def _fusable(self, x):
        """One common token count, a CUDA input and shapes inside the fused kernel's envelope (decided on the host)."""
        if not x.is_cuda or len(set(self.patch_list)) != 1 or any(lv.n_patches != self.patch_list[0] for lv in self.levels):
            return False
        key = (x.shape[1],)
        if getattr(self, "_fuse_key", None) != key:
            from .. import ops
            D = self.levels[0].embed_dim
            ppt = [lv.input_dim // x.shape[1] for lv in self.levels]
            ok = all(lv.embed_dim == D and lv.input_dim % x.shape[1] == 0 for lv in self.levels)
            # the fused kernel also needs every level's tokens to cover the image exactly (N * P_l == H * W, checked again by
            # sfcvit_hier_tokenizer_fwd); a level whose grouped pre-patches do not takes the composed path
            img = self.levels[0]._geom[0]
            ok = ok and all(lv.n_patches * p == img * img for lv, p in zip(self.levels, ppt))
            self._fuse_ok = ok and ops.hier_tokenizer_supported(len(self.levels), D, x.shape[1], ppt)
            self._fuse_key = key
        return self._fuse_ok

    def forward(self, x, one_kernel=None):
        if self._fusable(x):
            img = self.levels[0]._geom[0]
            if x.dim() != 4 or x.shape[2] != img or x.shape[3] != img:
                raise ValueError(f"expected [B, C, {img}, {img}] input, got {tuple(x.shape)}")
            return F.hier_tokenizer(x, [lv._pix_table(x.device) for lv in self.levels],
                                    [lv.proj.weight for lv in self.levels], [lv.proj.bias for lv in self.levels],
                                    self.fusion.weight, self.fusion.bias, one_kernel)
        return self.forward_unfused(x)

    def forward_unfused(self, x):
        """Level kernels, the linear resampling to the first level's token count + concatenation as one kernel
        (sfcvit_hier_resample_concat), fusion GEMM: any token counts."""
        patches = [level(x) for level in self.levels]
        widths = {p.shape[2] for p in patches}
        if not x.is_cuda or len(widths) != 1 or next(iter(widths)) % 8:
            raise RuntimeError("hierarchical tokenizer: the HIP path needs a CUDA (ROCm) input and level widths that are "
                               "one multiple of 8; there is no CPU fallback")
        return F.linear(F.hier_resample_concat(patches), self.fusion.weight, self.fusion.bias)


class HierarchicalHilbertEmbedding(_Hierarchical):
    _default_curve = hilbert_curve


class HierarchicalMortonEmbedding(_Hierarchical):
    _default_curve = z_curve


class HierarchicalMooreEmbedding(_Hierarchical):
    """multiscale/multi_moore.py:9-40."""
    _default_curve = moore_curve


class HierarchicalPeanoEmbedding(_Hierarchical):
    """multiscale/multi_peano.py:9-40."""
    _default_curve = peano_curve


class HierarchicalOnionEmbedding(_Hierarchical):
    """multiscale/multi_onion.py:8-43 (levels carry `onion_indices`)."""

    def __init__(self, img_size, in_channels, patch_size_list, embed_dim):
        super().__init__(img_size, in_channels, patch_size_list, embed_dim)

    @staticmethod
    def _level(img_size, pre_patch_size, patch_size, in_channels, embed_dim, curve_fn):
        return OnionGroupedEmbedding1D(img_size, pre_patch_size, patch_size, in_channels, embed_dim)


class HierarchicalRasterScanEmbedding(_Hierarchical):
    """multiscale/multi_zigzag.py:7-52 (levels have no index buffer)."""

    def __init__(self, img_size, in_channels, patch_size_list, embed_dim):
        super().__init__(img_size, in_channels, patch_size_list, embed_dim)

    @staticmethod
    def _level(img_size, pre_patch_size, patch_size, in_channels, embed_dim, curve_fn):
        return RasterScan1DGroupedEmbedding(img_size, pre_patch_size, patch_size, in_channels, embed_dim)
