from sfcvit.tokenizers.embeddings import HilbertEmbedding  # noqa: F401
