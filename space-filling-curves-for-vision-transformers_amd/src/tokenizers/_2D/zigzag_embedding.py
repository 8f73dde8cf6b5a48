from sfcvit.tokenizers.embeddings import ZigzagEmbedding  # noqa: F401
