from .base_patch_embedding import BasePatchEmbedding  # noqa: F401
from .embeddings import (HilbertEmbedding, HilbertEmbedding1D, MooreEmbedding1D, MortonEmbedding1D,  # noqa: F401
                         OnionEmbedding1D, OnionGroupedEmbedding1D, PeanoEmbedding1D, RandomEmbedding,
                         RasterScan1DEmbedding, RasterScan1DGroupedEmbedding, SFCEmbedding1D, ZigzagEmbedding)
from .multiscale import (HierarchicalHilbertEmbedding, HierarchicalMooreEmbedding,  # noqa: F401
                         HierarchicalMortonEmbedding, HierarchicalOnionEmbedding, HierarchicalPeanoEmbedding,
                         HierarchicalRasterScanEmbedding)
