from .space_filling_curves import (embed_and_prune_sfc, grid_size, hilbert_curve, moore_curve,  # noqa: F401
                                   peano_curve, raster_curve, z_curve, curve_table, curve_table_rc)
