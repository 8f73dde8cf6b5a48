#!/usr/bin/env python3
"""Cycle budget of a k-tile of the persistent GEMM's two-phase schedule, from the stamped build (trace_build.py; SFCVIT_LIB
names it).  Per wave group (waves 0 and 4 of one workgroup): clocks of  X mma | X mma end -> Y mma start (= the Y load
section of this group under the other group's MFMAs) | Y mma | Y end -> next X mma start, steady-state k-tiles averaged,
tile boundaries (the k-tile that ends with the epilogue, and the one after it) listed apart.
    SFCVIT_LIB=tools/probe/lib_trace.so python tools/gemm_lab/trace_2phase.py N K [M]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import _lib, ops  # noqa: E402

N, K = int(sys.argv[1]), int(sys.argv[2])
M = int(sys.argv[3]) if len(sys.argv) > 3 else 50176
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn((M, K), device="cuda", generator=g).bfloat16()
b = torch.randn((N, K), device="cuda", generator=g).bfloat16()
for _ in range(3):
    ops.gemm(a, b)
torch.cuda.synchronize()
print("kernel:", ops.last_gemm_kernel(), f"M={M} N={N} K={K}")
buf = (ctypes.c_ulonglong * 512)()
_lib.lib.sfcvit_lab_trace.argtypes = [ctypes.c_void_p]
assert _lib.lib.sfcvit_lab_trace(buf) == 0
KT = K // 64
for w in range(2):
    st = [buf[w * 256 + i] for i in range(256) if buf[w * 256 + i]]
    nk = len(st) // 4
    rows = []
    for k in range(nk - 1):
        s0, s1, s2, s3 = st[4 * k:4 * k + 4]
        rows.append((k % KT, s1 - s0, s2 - s1, s3 - s2, st[4 * k + 4] - s3))
    steady = [r for r in rows if 1 <= r[0] < KT - 1]
    avg = [sum(r[i] for r in steady) / len(steady) for i in range(1, 5)]
    print(f"wave group {w}: {nk} k-tiles stamped; steady-state k-tile ({len(steady)} averaged): X mma {avg[0]:.0f} | -> Y mma {avg[1]:.0f} | "
          f"Y mma {avg[2]:.0f} | -> next X mma {avg[3]:.0f} | total {sum(avg):.0f} clocks")
    for r in rows:
        if r[0] in (KT - 1, 0):
            print(f"    k-tile {r[0]:2d} ({'last of a tile: Y mma section includes the epilogue' if r[0] == KT - 1 else 'first of a tile'}): "
                  f"{r[1]} | {r[2]} | {r[3]} | {r[4]}  = {sum(r[1:])}")
