#!/usr/bin/env python3
"""Race screen for the 8-phase GEMM kernels (cdna_hip_programming.md: a new sync structure is screened over many runs
at several sizes): every shape is run `reps` times and must reproduce its first result bit for bit, and the first
result must match the generic kernel (kc form) / fp32 math (k-major form)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator(device="cuda").manual_seed(0)
bad = 0
for (M, N, K) in [(50176, 768, 768), (50176, 2304, 768), (50176, 768, 3072), (256 * 300, 256, 256), (224 * 17, 512, 384),
                  (36864, 1024, 1024), (256, 256, 256), (256 * 3, 256 * 5, 512)]:
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.1).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    r = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    first = ops.gemm(a, w, bias=b, residual=r, dropout_p=0.1, dropout_seed=7)
    ok = torch.equal(first, ops.gemm(a, w, bias=b, residual=r, dropout_p=0.1, dropout_seed=7, force_generic=1))
    n_diff = sum(int(not torch.equal(first, ops.gemm(a, w, bias=b, residual=r, dropout_p=0.1, dropout_seed=7))) for _ in range(reps))
    print(f"kc {M}x{N}x{K}: matches generic {ok}, {n_diff}/{reps} runs differ")
    bad += (not ok) + n_diff
for (M, N, K) in [(768, 768, 50176), (3072, 768, 50176), (768, 2304, 12544), (256, 256, 1024), (1024, 4096, 36864)]:
    a = torch.randn(K, M, device="cuda", generator=g).bfloat16()
    x = torch.randn(K, N, device="cuda", generator=g).bfloat16()
    first = ops.gemm(a, x, a_kmajor=True, b_kmajor=True)
    ref = a.float().t() @ x.float()
    err = float((first.float() - ref).abs().max() / ref.abs().max())
    n_diff = sum(int(not torch.equal(first, ops.gemm(a, x, a_kmajor=True, b_kmajor=True))) for _ in range(reps))
    print(f"km {M}x{N}x{K}: rel err vs fp32 {err:.4f}, {n_diff}/{reps} runs differ")
    bad += (err > 0.02) + n_diff
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
