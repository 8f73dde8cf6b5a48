from sfcvit.models.altvit import *  # noqa: F401,F403
from sfcvit.models.altvit import (Attention, FeedForward, HilbertPatchEmbedding, HilbertViT, SimpleViT,  # noqa: F401
                                  Transformer, pair, posemb_sincos_1d)
