#!/usr/bin/env python3
"""Where an epilogue of the persistent GEMM spends its clocks (TRACE_EPI=1 build of trace_build.py): per tile of the traced
workgroup and wave group, arithmetic (accumulators -> packed bf16 rows, incl. the loads of residual rows) | exchange through the
LDS patch (one asm block, one wait) | issue of the store instructions.
    TRACE_EPI=1 python tools/gemm_lab/trace_build.py && SFCVIT_LIB=tools/probe/lib_trace.so python tools/gemm_lab/trace_epilogue.py N K [residual dropout]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import _lib, ops  # noqa: E402

N, K = int(sys.argv[1]), int(sys.argv[2])
res, drop = (len(sys.argv) > 3 and sys.argv[3] == "1"), (float(sys.argv[4]) if len(sys.argv) > 4 else 0.0)
M = 50176
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn((M, K), device="cuda", generator=g).bfloat16()
b = torch.randn((N, K), device="cuda", generator=g).bfloat16()
bias = torch.randn(N, device="cuda", generator=g).bfloat16()
r = torch.randn((M, N), device="cuda", generator=g).bfloat16() if res else None
for _ in range(3):
    ops.gemm(a, b, bias=bias, residual=r, dropout_p=drop, dropout_seed=5)
torch.cuda.synchronize()
print("kernel:", ops.last_gemm_kernel(), f"M={M} N={N} K={K}")
buf = (ctypes.c_ulonglong * 480)()
_lib.lib.sfcvit_lab_trace_epilogue.argtypes = [ctypes.c_void_p]
assert _lib.lib.sfcvit_lab_trace_epilogue(buf) == 0
for w in range(2):
    rows = []
    for t in range(60):
        s = [buf[(w * 60 + t) * 4 + k] for k in range(4)]
        if all(s):
            rows.append((s[1] - s[0], s[2] - s[1], s[3] - s[2]))
    if not rows:
        continue
    print(f"wave group {w}: {len(rows)} epilogues;   arithmetic | LDS exchange | store issue   (clocks)")
    for i, x in enumerate(rows[:8]):
        print(f"    tile {i}: {x[0]:6d} | {x[1]:6d} | {x[2]:6d}   = {sum(x)}")
    late = rows[3:]
    if late:
        print("    mean from tile 3 on: " + " | ".join(f"{sum(x[k] for x in late) / len(late):6.0f}" for k in range(3)))
