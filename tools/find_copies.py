#!/usr/bin/env python3
"""Where do the copy / fill / cast launches of one training step come from?  Runs the benched step (bench.py's workload)
under torch.profiler with Python stacks and prints, per (op, innermost repository source line), how many launches one step
makes and how many bytes they move.
    python tools/find_copies.py [--workload vit_b16_224_hilbert] [--batch 256]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="vit_b16_224_hilbert")
ap.add_argument("--batch", type=int, default=0)
args = ap.parse_args()

from sfcvit.training import FusedAdamW, mixup_soft_targets, train_step  # noqa: E402

tok, img, patch, D, depth, heads, mlp, classes, batch, _ = bench.WORKLOADS[args.workload]
batch = args.batch or batch
dev = torch.device("cuda:0")
model = bench.build(args.workload, 0.1).to(dev, dtype=torch.bfloat16).train()
opt = FusedAdamW(model.parameters(), lr=3e-4, weight_decay=5e-5, max_grad_norm=1.0)
images = torch.randn(batch, 3, img, img, device=dev)
targets = mixup_soft_targets(torch.randint(0, classes, (batch,), device=dev), classes, lam=0.7)
for _ in range(3):
    train_step(model, images, targets, opt)
torch.cuda.synchronize()

with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            with_stack=True, record_shapes=True) as prof:
    train_step(model, images, targets, opt)
    torch.cuda.synchronize()

# 1. device side: every launch that is not one of this repository's kernels
dev_rows = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA and "sfcvit" not in ev.name:
        dev_rows[ev.name[:100]] += 1
print("device launches outside libsfcvit_hip.so, one step:")
for name, n in dev_rows.most_common():
    print(f"{n:5d} x {name}")

# 2. host side: the aten ops that launched them, by shape and innermost repository frame
print("\naten ops with a device launch underneath, one step:")
rows = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    if not any("sfcvit" not in k.name for k in ev.kernels):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::") and ev.cpu_parent.kernels:
        continue                                           # count the outermost aten op only
    where = "?"
    for fr in ev.stack or []:                              # innermost first
        if ("sfcvit" in fr or "bench.py" in fr) and "find_copies" not in fr:
            where = fr.split("space-filling-curves-for-vision-transformers_amd/")[-1]
            break
    shapes = tuple(tuple(s) for s in (ev.input_shapes or []) if s)
    rows[(ev.name, where, str(shapes[:2]), ",".join(sorted({k.name[:40] for k in ev.kernels})))] += 1
for (name, where, shapes, kern), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d} x {name:18s} {shapes:44s} {where:60s} {kern}")
