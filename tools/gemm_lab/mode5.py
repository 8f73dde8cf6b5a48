import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch
from sfcvit import ops
M, N, K = 50176, 2304, 768
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
b = torch.randn(N, K, device="cuda", generator=g).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("zero_ 231MB          %.1f us" % t(lambda: out.zero_()))
print("mode5 fresh out      %.1f us" % t(lambda: ops.gemm(a, b, force_generic=5)))
print("mode5 preallocated   %.1f us" % t(lambda: ops.gemm(a, b, force_generic=5, out=out)))
az = torch.zeros_like(a); bz = torch.zeros_like(b)
print("mode5 zeros inputs   %.1f us" % t(lambda: ops.gemm(az, bz, force_generic=5, out=out)))
print("mode4 zeros inputs   %.1f us" % t(lambda: ops.gemm(az, bz, force_generic=4, out=out)))
print("mode4 random inputs  %.1f us" % t(lambda: ops.gemm(a, b, force_generic=4, out=out)))
