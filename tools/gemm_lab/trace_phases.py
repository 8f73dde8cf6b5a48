#!/usr/bin/env python3
"""Per-phase variant of trace_tile.py for a -DP8_LAB_TRACE=<wg> -DP8_LAB_TRACE_PHASE build (QKV forward shape): prints
the clock intervals between consecutive stamps, 8 per line = the 4 phases of two k-tiles."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "space-filling-curves-for-vision-transformers_amd"))
import torch
from sfcvit import _lib, ops
M, N, K = 50176, 2304, 768
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn((M, K), device="cuda", generator=g).bfloat16(); b = torch.randn((N, K), device="cuda", generator=g).bfloat16()
for _ in range(3): ops.gemm(a, b)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
_lib.lib.sfcvit_lab_trace.argtypes = [ctypes.c_void_p]
assert _lib.lib.sfcvit_lab_trace(buf) == 0
for w in range(2):
    st = [buf[w * 256 + i] for i in range(256)]; st = [x for x in st if x]
    d = [st[i] - st[i - 1] for i in range(1, len(st))]
    print("group", w, len(st))
    # stamps per k-tile: p0, p1, p2, p3 ; per tile 48 + 2 epilogue stamps
    for t in range(0, min(len(d), 150), 8):
        print("  ", d[t:t + 8])
