#!/usr/bin/env python3
"""The fused clip + AdamW kernel alone (28 B of state per parameter: fp32 master / m / v read and written, bf16 gradient
read, bf16 parameter written) at ViT-B (86.6 M) and ViT-L (304 M) sizes; optional A/B against another build of the
library (SFCVIT_LIB).    python tools/bench_adamw.py [n_params ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import ops  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [86_567_656, 304_330_000]
for n in sizes:
    n = n // 8 * 8
    master = torch.randn(n, device="cuda")
    param = master.bfloat16()
    grad = (torch.randn(n, device="cuda") * 1e-2).bfloat16()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    sumsq = torch.ones(1, device="cuda")

    def step(i=[0]):
        i[0] += 1
        ops.adamw_step(param, master, grad, m, v, sumsq, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-5,
                       max_norm=1.0, step=i[0])

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    med = sorted(ts)[len(ts) // 2]
    print(f"n = {n / 1e6:7.1f} M   {med:8.1f} us   {28.0 * n / med / 1e6:6.2f} TB/s of 28 B/param")
