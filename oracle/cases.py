"""Fixture configurations shared by tools/make_golden.py and tests/ (TEST INFRASTRUCTURE)."""
from .vit_oracle import OracleConfig

# name -> (config, batch)
MODEL_CASES = {
    # HilbertEmbedding1D(32,256,3,128) + VisionTransformer1D
    "hilbert32_1d": (OracleConfig("hilbert1d", 32, 256, 3, 128, depth=2, n_heads=2, mlp_dim=256,
                                  num_classes=10, variant="1d"), 4),
    # MortonEmbedding1D at the B/16@224 token grid (N=196, tail masking in attention)
    "morton224_1d": (OracleConfig("morton1d", 224, 256, 3, 128, depth=1, n_heads=2, mlp_dim=256,
                                  num_classes=10, variant="1d"), 2),
    # HilbertEmbedding1D at 224: four intra-tile pixel orders
    "hilbert224_1d": (OracleConfig("hilbert1d", 224, 256, 3, 128, depth=1, n_heads=2, mlp_dim=256,
                                   num_classes=10, variant="1d"), 2),
    # ViT-Tiny-like raster model without the mixer (VisionTransformer), BASELINE config 1 shape
    "raster32_2d": (OracleConfig("raster1d", 32, 256, 3, 192, depth=2, n_heads=3, mlp_dim=768,
                                 num_classes=10, variant="2d"), 4),
    # SFCEmbedding1D(p=16, g=1): standard ViT/16 patches in Hilbert order, 4x4 token grid
    "sfc64_p16": (OracleConfig("sfc", 64, 1, 3, 128, depth=1, n_heads=2, mlp_dim=256,
                               num_classes=10, variant="1d", pre_patch_size=16, curve="hilbert"), 3),
    # SFCEmbedding1D(p=2, g=16) with the Z curve: general (non tile-aligned feature order) path
    "sfc32_p2g16_z": (OracleConfig("sfc", 32, 16, 3, 64, depth=1, n_heads=1, mlp_dim=128,
                                   num_classes=7, variant="2d", pre_patch_size=2, curve="z"), 3),
}

CURVE_SMALL_N = (2, 4, 8, 14, 16, 24, 32)
CURVE_SHA_N = (224, 384)
CURVE_KINDS = ("hilbert", "z", "moore", "peano")
