from sfcvit.tokenizers.embeddings import OnionEmbedding1D  # noqa: F401
