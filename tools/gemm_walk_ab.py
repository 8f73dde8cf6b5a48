#!/usr/bin/env python3
"""One process per SFCVIT_GEMM_WALK value: the wide forward GEMMs of a ViT-B layer (M = 50 176) on the persistent kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "space-filling-curves-for-vision-transformers_amd"))
import torch
from sfcvit import ops
M = 50176
g = torch.Generator(device="cuda").manual_seed(0)
out = []
for name, n, k in (("ffn1 fwd N3072 K768", 3072, 768), ("qkv fwd N2304 K768", 2304, 768), ("ffn2 fwd N768 K3072", 768, 3072)):
    a = torch.randn(M, k, device="cuda", generator=g).bfloat16()
    w = (torch.randn(n, k, device="cuda", generator=g) / 28).bfloat16()
    b = torch.zeros(n, device="cuda").bfloat16()
    for _ in range(3):
        ops.gemm(a, w, bias=b)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm(a, w, bias=b)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    out.append(f"{name} {sorted(ts)[3]:.1f} us [{ops.last_gemm_kernel()}]")
print("WALK=" + os.environ.get("SFCVIT_GEMM_WALK", "0") + ": " + " | ".join(out))
