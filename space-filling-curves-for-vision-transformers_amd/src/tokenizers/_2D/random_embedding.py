from sfcvit.tokenizers.embeddings import RandomEmbedding  # noqa: F401
