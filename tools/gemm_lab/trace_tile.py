#!/usr/bin/env python3
"""Prints the shader-clock stamps a -DP8_LAB_TRACE build of gemm8p.hip records for one workgroup (waves 0 and 4):
per tile, one stamp after each k-tile's phase-3 wait + barrier, then epilogue start and end.  SFCVIT_LIB names the
lab library.  Usage: trace_tile.py N K"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import _lib, ops  # noqa: E402

M = 50176
N, K = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn((M, K), device="cuda", generator=g).bfloat16()
b = torch.randn((N, K), device="cuda", generator=g).bfloat16()
for _ in range(3):
    ops.gemm(a, b)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
_lib.lib.sfcvit_lab_trace.argtypes = [ctypes.c_void_p]
rc = _lib.lib.sfcvit_lab_trace(buf)
assert rc == 0, rc
KT = K // 64
per = KT + 2
for w in range(2):
    st = [buf[w * 256 + i] for i in range(256)]
    st = [x for x in st if x]
    print(f"wave group {w}: {len(st)} stamps, {len(st) // per} tiles; clocks between stamps (k = k-tile, E = epilogue)")
    for t in range(len(st) // per):
        row = st[t * per:(t + 1) * per]
        prev = st[t * per - 1] if t else row[0]
        d = [row[0] - prev] + [row[i] - row[i - 1] for i in range(1, per)]
        # order within a tile: k-tile stamps 0..KT-1, then E start, E end (group 1: E start/end come before the last bar)
        print(f"  tile {t}: " + " ".join(f"{x}" for x in d))
