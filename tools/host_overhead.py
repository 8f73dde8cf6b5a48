#!/usr/bin/env python3
"""How much host time a train step takes to ENQUEUE (Python + ctypes + torch allocator), next to the GPU time of the
step: the margin by which the step is GPU-bound.  ViT-B/16 @ 224, batch 256 by default."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from sfcvit.training import FusedAdamW, mixup_soft_targets, train_step  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "vit_b16_224_hilbert"
img, patch, D, depth, heads, mlp, classes, batch = bench.WORKLOADS[workload]
model = bench.build(workload, 0.1).to("cuda", dtype=torch.bfloat16).train()
opt = FusedAdamW(model.parameters(), lr=3e-4, weight_decay=5e-5)
x = torch.randn(batch, 3, img, img, device="cuda")
t = mixup_soft_targets(torch.randint(0, classes, (batch,), device="cuda"), classes)
for _ in range(5):
    train_step(model, x, t, opt)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    train_step(model, x, t, opt)
t1 = time.perf_counter()                       # everything enqueued (the queue is deep enough not to block for 10 steps?)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{workload}: enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, total {1e3 * (t2 - t0) / n:.2f} ms/step")
# enqueue-only rate with the GPU idle at the start of every step: an upper bound on the host cost
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    a = time.perf_counter()
    train_step(model, x, t, opt)
    ts.append(time.perf_counter() - a)
    torch.cuda.synchronize()
print(f"host time of one step with an empty queue: {1e3 * min(ts):.2f} ms (min of 5)")
