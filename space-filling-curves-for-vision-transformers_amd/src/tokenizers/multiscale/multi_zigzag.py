from sfcvit.tokenizers.embeddings import RasterScan1DGroupedEmbedding  # noqa: F401
from sfcvit.tokenizers.multiscale import HierarchicalRasterScanEmbedding  # noqa: F401
