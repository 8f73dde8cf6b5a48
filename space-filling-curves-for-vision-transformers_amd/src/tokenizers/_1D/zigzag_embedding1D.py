from sfcvit.tokenizers.embeddings import RasterScan1DEmbedding  # noqa: F401
