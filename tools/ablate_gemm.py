import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch
from sfcvit import ops
M = 50176
SH = {"fwd_qkv": (M, 2304, 768, False, False), "dx_ffn2": (M, 3072, 768, False, True), "dw_ffn1": (3072, 768, M, True, True), "fwd_ffn2": (M,768,3072,False,False)}
g = torch.Generator(device="cuda").manual_seed(0)
for name,(m,n,k,akm,bkm) in SH.items():
    a = torch.randn((k, m) if akm else (m, k), device="cuda", generator=g).bfloat16()
    b = torch.randn((k, n) if bkm else (n, k), device="cuda", generator=g).bfloat16()
    res = {}
    for mode in (4, 1, 2, 3, 5):
        for _ in range(2): ops.gemm(a, b, a_kmajor=akm, b_kmajor=bkm, force_generic=mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.gemm(a, b, a_kmajor=akm, b_kmajor=bkm, force_generic=mode)
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 5 * 1e3
    print(f"{name:10s} big {res[4]:7.1f} us | generic {res[1]:7.1f} | DMA-only {res[2]:7.1f} | MFMA+LDS-only {res[3]:7.1f} | epilogue-only {res[5]:7.1f}")
