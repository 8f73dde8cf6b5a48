from sfcvit.tokenizers.embeddings import SFCEmbedding1D  # noqa: F401
from sfcvit.tokenizers.multiscale import HierarchicalMooreEmbedding  # noqa: F401
