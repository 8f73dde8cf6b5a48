from sfcvit.tokenizers.embeddings import SFCEmbedding1D  # noqa: F401
from sfcvit.tokenizers.multiscale import HierarchicalHilbertEmbedding  # noqa: F401
