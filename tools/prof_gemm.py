#!/usr/bin/env python3
"""Few launches of chosen GEMM shapes/kernels for rocprofv3 --pmc runs (see profiles/).
usage: prof_gemm.py shape[,shape...] mode    (mode = force_generic value of include/sfcvit.h: 0 auto, 1 generic, 6 ring 256x128, 8 / 9 persistent 8-phase)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import ops  # noqa: E402

M = 50176
SHAPES = {
    "fwd_qkv": (M, 2304, 768, False, False),
    "fwd_ffn2": (M, 768, 3072, False, False),
    "dx_ffn2": (M, 3072, 768, False, True),
    "dw_ffn1": (3072, 768, M, True, True),
}
which = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator(device="cuda").manual_seed(0)
for name in which:
    m, n, k, akm, bkm = SHAPES[name]
    a = torch.randn((k, m) if akm else (m, k), device="cuda", generator=g).bfloat16()
    b = torch.randn((k, n) if bkm else (n, k), device="cuda", generator=g).bfloat16()
    for _ in range(3):
        ops.gemm(a, b, a_kmajor=akm, b_kmajor=bkm, force_generic=mode)
    torch.cuda.synchronize()
