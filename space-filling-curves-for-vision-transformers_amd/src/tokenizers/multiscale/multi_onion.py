from sfcvit.tokenizers.embeddings import OnionGroupedEmbedding1D as OnionEmbedding1D  # noqa: F401
from sfcvit.tokenizers.multiscale import HierarchicalOnionEmbedding  # noqa: F401
