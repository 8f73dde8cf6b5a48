from sfcvit.tokenizers.embeddings import MooreEmbedding1D  # noqa: F401
