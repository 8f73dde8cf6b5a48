#!/usr/bin/env python3
"""What a GEMM costs while some CUs are held by another kernel (an RCCL all-reduce overlapped with backward does
that on a multi-GPU node).  A hog kernel occupies `n` CUs on a side stream for the whole measurement; the 8-phase
GEMM draws its tiles from a queue, so it should slow down by ~256 / (256 - n), not by 2x as a fixed tile list would."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import ops  # noqa: E402

# the hog kernel is a lab helper with its own tiny library (tools/occupy/occupy.hip), not part of the product ABI
_occ_dir = os.path.join(ROOT, "tools", "occupy")
_occ_so = os.path.join(_occ_dir, "liboccupy.so")
if not os.path.exists(_occ_so):
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC",
                           os.path.join(_occ_dir, "occupy.hip"), "-o", _occ_so])
occ = ctypes.CDLL(_occ_so)
occ.lab_occupy.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]

M = 50176
g = torch.Generator(device="cuda").manual_seed(0)
side = torch.cuda.Stream()
sink = torch.zeros(4, device="cuda", dtype=torch.int32)


def run(n_busy, N, K, reps=10):
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.1).bfloat16()
    for _ in range(3):
        ops.gemm(a, w)
    torch.cuda.synchronize()
    if n_busy:
        assert occ.lab_occupy(n_busy, 40_000_000, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream)) == 0
        torch.cuda._sleep(2_000_000)          # let the hog get onto its CUs first
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(a, w)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for N, K in [(768, 768), (2304, 768), (768, 3072)]:
    base = run(0, N, K)
    row = f"N={N:5d} K={K:5d}: all CUs free {base:7.1f} us"
    for n in (16, 32, 64):
        t = run(n, N, K)
        row += f" | {n} busy {t:7.1f} us (x{t / base:.2f}, ideal x{256 / (256 - n):.2f})"
    print(row)
