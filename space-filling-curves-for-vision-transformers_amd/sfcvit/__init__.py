"""sfcvit -- MI355X (gfx950) native hot path of Space-Filling-Curves-for-Vision-Transformers.

Layout mirrors the reference's `src/` package:
  sfcvit.curves.space_filling_curves   curve tables (native integer generators)
  sfcvit.tokenizers                    HilbertEmbedding1D, MortonEmbedding1D, RasterScan1DEmbedding, SFCEmbedding1D
  sfcvit.models.vit                    VisionTransformer, VisionTransformer1D and their building blocks
  sfcvit.training                      train step, fused AdamW, data-parallel gradient reducer
Everything computes through libsfcvit_hip.so (include/sfcvit.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
