#!/usr/bin/env python3
"""Per-kernel in-step durations of two bench.py JSON lines side by side (roofline_detail: HIP events around the launches of
the timed steps): does an isolated kernel gain survive inside the training step?
    python tools/ab_step_kernels.py a.json b.json"""
import json
import sys

a, b = (json.load(open(f)) for f in sys.argv[1:3])
print(f"{'':44s} {sys.argv[1]:>22s} {sys.argv[2]:>22s}")
print(f"{'images/s':44s} {a['value']:22.1f} {b['value']:22.1f}")
print(f"{'ms/step':44s} {a['ms_per_step']:22.3f} {b['ms_per_step']:22.3f}")
da, db = a["roofline_detail"], b["roofline_detail"]
for k in da:
    if k in db and isinstance(da[k], dict) and "avg_launch_ms" in da[k]:
        x, y = da[k]["avg_launch_ms"] * 1e3, db[k]["avg_launch_ms"] * 1e3
        print(f"{k:44s} {x:15.1f} us x{da[k]['launches']:<4d} {y:15.1f} us x{db[k]['launches']:<4d} {(x / y - 1) * 100:+6.1f} %")
