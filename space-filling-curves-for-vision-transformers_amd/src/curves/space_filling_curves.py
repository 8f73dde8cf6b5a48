from sfcvit.curves.space_filling_curves import *  # noqa: F401,F403
from sfcvit.curves.space_filling_curves import embed_and_prune_sfc, grid_size, hilbert_curve, moore_curve, peano_curve, raster_curve, z_curve  # noqa: F401
