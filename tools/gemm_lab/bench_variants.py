#!/usr/bin/env python3
"""Times sfcvit_gemm's persistent kernel on the ViT-B forward / dX shapes with whatever library SFCVIT_LIB names
(lab builds of gemm8p.hip with -DP8_LAB_*; see README.md).  One line per shape: median us and TFLOP/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import ops  # noqa: E402

M = 50176
SHAPES = [("qkv", 2304, 768), ("out", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072), ("dqkv", 768, 2304)]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    cases = []
    for name, n, k in SHAPES:
        a = torch.randn((M, k), device="cuda", generator=g).bfloat16()
        b = torch.randn((n, k), device="cuda", generator=g).bfloat16()
        cases.append((name, n, k, a, b))
        ops.gemm(a, b)
    torch.cuda.synchronize()
    times = {c[0]: [] for c in cases}
    for _ in range(7):
        for name, n, k, a, b in cases:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm(a, b)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / 3)
    row = os.path.basename(os.environ.get("SFCVIT_LIB", "product")) + ":"
    for name, n, k, *_ in cases:
        t = sorted(times[name])[3]
        row += f"  {name} {t * 1e3:6.1f} us {2.0 * M * n * k / t / 1e9:5.0f} TF"
    print(row)


if __name__ == "__main__":
    main()
