from sfcvit.models.vit import (FactorisedLinear, MixerBlock, MultiLayerPredictor, TransformerSeqEncoder,  # noqa: F401
                               VisionTransformer, VisionTransformer1D)
