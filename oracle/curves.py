"""ctypes front end of curves_oracle.c (TEST INFRASTRUCTURE; see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcurves_oracle.so")
KINDS = {"hilbert": 0, "z": 1, "moore": 2, "peano": 3, "onion": 4}
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "curves_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcurves_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_embed_and_prune_sfc.restype = ctypes.c_int64
        _lib.oracle_embed_and_prune_sfc.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        for fn in (_lib.oracle_spiral_flat, _lib.oracle_hilbert2d_flat):
            fn.restype = ctypes.c_int64
            fn.argtypes = [ctypes.c_int, ctypes.c_void_p]
    return _lib


def embed_and_prune_sfc(kind, width, height):
    """(i, j) pairs exactly as the reference's embed_and_prune_sfc returns them
    (space_filling_curves.py:471-491): int64 array [n, 2]."""
    out = np.zeros((width * height, 2), dtype=np.int64)
    n = _load().oracle_embed_and_prune_sfc(KINDS[kind], width, height, out.ctypes.data)
    if n < 0:
        raise ValueError(kind)
    if n > width * height:
        raise ValueError(f"{kind} {width}x{height}: {n} points kept (not a permutation)")
    return out[:n]


def flat_table(kind, n):
    """Flat r*n+c table as SFCEmbedding1D._sfc_indices builds it
    (src/tokenizers/multiscale/multi_hilbert.py:68-72); 'raster' is the identity
    (RasterScan1DEmbedding has no table, zigzag_embedding1D.py:30-39)."""
    if kind == "raster":
        return np.arange(n * n, dtype=np.int64)
    if kind == "spiral":        # OnionEmbedding1D.onion_indices (onion_embedding1D.py:35-53), any n
        out = np.zeros(n * n, dtype=np.int64)
        _load().oracle_spiral_flat(n, out.ctypes.data)
        return out
    if kind == "hilbert_t":     # _2D/HilbertEmbedding._get_hilbert_indices (hilbert_embedding.py:30-45), n = 2^k
        out = np.zeros(n * n, dtype=np.int64)
        cnt = _load().oracle_hilbert2d_flat(n, out.ctypes.data)
        if cnt != n * n:
            raise ValueError(f"hilbert_t: grid {n} is not a power of two")
        return out
    ij = embed_and_prune_sfc(kind, n, n)
    return ij[:, 0] * n + ij[:, 1]
