from .base_patch_embedding import BasePatchEmbedding  # noqa: F401
from .embeddings import (HilbertEmbedding1D, MortonEmbedding1D, MooreEmbedding1D, PeanoEmbedding1D,  # noqa: F401
                         RasterScan1DEmbedding, SFCEmbedding1D)
