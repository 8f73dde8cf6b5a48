from sfcvit.tokenizers.base_patch_embedding import BasePatchEmbedding  # noqa: F401
