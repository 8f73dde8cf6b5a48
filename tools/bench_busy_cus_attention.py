#!/usr/bin/env python3
"""The persistent attention backward while some CUs are held by another kernel (what an RCCL all-reduce beside backward
does): items dealt from a counter (default) against a fixed stride (SFCVIT_ATTN_BWD_QUEUE=0), one process per setting.
The hog is tools/occupy/occupy.hip, as in tools/bench_busy_cus.py."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import ops  # noqa: E402

_occ_dir = os.path.join(ROOT, "tools", "occupy")
_occ_so = os.path.join(_occ_dir, "liboccupy.so")
if not os.path.exists(_occ_so):
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(_occ_dir, "occupy.hip"), "-o", _occ_so])
occ = ctypes.CDLL(_occ_so)
occ.lab_occupy.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
B, N, D, H, p = 256, 196, 768, 12, 0.1
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3 * D, device="cuda", generator=g).bfloat16()
dout = torch.randn(B, N, D, device="cuda", generator=g).bfloat16()
out, lse = ops.attention_fwd(qkv, H, p, 5)
side = torch.cuda.Stream()
sink = torch.zeros(4, device="cuda", dtype=torch.int32)
ref = ops.attention_bwd(qkv, out, lse, dout, H, p, 5)


def run(n_busy, reps=10):
    torch.cuda.synchronize()
    if n_busy:
        assert occ.lab_occupy(n_busy, 40_000_000, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream)) == 0
        torch.cuda._sleep(2_000_000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dq = ops.attention_bwd(qkv, out, lse, dout, H, p, 5)
    e1.record()
    torch.cuda.synchronize()
    assert torch.equal(dq, ref)
    return e0.elapsed_time(e1) / reps * 1e3


base = run(0)
row = f"queue={os.environ.get('SFCVIT_ATTN_BWD_QUEUE', '1')}: all CUs free {base:7.1f} us"
for n in (8, 32, 64):
    t = run(n)
    row += f" | {n} busy {t:7.1f} us (x{t / base:.2f}, ideal x{256 / (256 - n):.2f})"
print(row)
