from sfcvit.tokenizers.embeddings import PeanoEmbedding1D  # noqa: F401
