from sfcvit.tokenizers.embeddings import HilbertEmbedding1D  # noqa: F401
