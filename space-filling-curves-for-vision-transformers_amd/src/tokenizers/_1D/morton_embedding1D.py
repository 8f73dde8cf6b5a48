from sfcvit.tokenizers.embeddings import MortonEmbedding1D  # noqa: F401
