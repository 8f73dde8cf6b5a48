#!/usr/bin/env python3
"""Micro-benchmark of sfcvit_gemm on the ViT-B/16@224 (batch 256) shapes, all three layouts.
Interleaved rounds in one process (cdna_hip_programming.md §5.4 rule 24), random data."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import ops  # noqa: E402

M = 50176
SHAPES = [  # (name, M, N, K, a_kmajor, b_kmajor)
    ("fwd qkv      x[M,768]  W[2304,768]", M, 2304, 768, False, False),
    ("fwd out      x[M,768]  W[768,768] ", M, 768, 768, False, False),
    ("fwd ffn1     x[M,768]  W[3072,768]", M, 3072, 768, False, False),
    ("fwd ffn2     x[M,3072] W[768,3072]", M, 768, 3072, False, False),
    ("dx  ffn2     dy[M,768] W[768,3072]", M, 3072, 768, False, True),
    ("dx  ffn1     dy[M,3072] W[3072,768]", M, 768, 3072, False, True),
    ("dx  qkv      dy[M,2304] W[2304,768]", M, 768, 2304, False, True),
    ("dW  ffn1     dy[M,3072]^T x[M,768]", 3072, 768, M, True, True),
    ("dW  ffn2     dy[M,768]^T h[M,3072]", 768, 3072, M, True, True),
    ("dW  out      dy[M,768]^T x[M,768] ", 768, 768, M, True, True),
    ("dW  qkv      dy[M,2304]^T x[M,768]", 2304, 768, M, True, True),
]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    g = torch.Generator(device="cuda").manual_seed(0)
    cases = []
    for name, m, n, k, akm, bkm in SHAPES:
        a = torch.randn((k, m) if akm else (m, k), device="cuda", generator=g).bfloat16()
        b = torch.randn((k, n) if bkm else (n, k), device="cuda", generator=g).bfloat16()
        cases.append((name, m, n, k, akm, bkm, a, b))
    modes = {"auto": 0, "generic": 1, "ring 256x128": 6, "torch.matmul": -1}

    def run(a, b, akm, bkm, md):
        if md == -1:      # the vendor library through torch, as a known-good yardstick on the same device (not product)
            return torch.matmul(a.t() if akm else a, b if bkm else b.t())
        return ops.gemm(a, b, a_kmajor=akm, b_kmajor=bkm, force_generic=md)
    for c in cases:
        for md in modes.values():
            run(c[6], c[7], c[4], c[5], md)
    torch.cuda.synchronize()
    times = {(c[0], k): [] for c in cases for k in modes}
    for _ in range(rounds):
        for name, m, n, k, akm, bkm, a, b in cases:
            for mk, md in modes.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    run(a, b, akm, bkm, md)
                e1.record()
                torch.cuda.synchronize()
                times[(name, mk)].append(e0.elapsed_time(e1) / 3)
    tot = {k: 0.0 for k in modes}
    for name, m, n, k, *_ in cases:
        row = f"{name:38s}"
        for mk in modes:
            ts = times[(name, mk)]
            t = sorted(ts)[len(ts) // 2]
            tot[mk] += t
            row += f" | {mk} {t * 1e3:6.1f} us {2.0 * m * n * k / t / 1e9:6.0f} TF"
        print(row)
    print("sum ms: " + ", ".join(f"{k} {v:.3f}" for k, v in tot.items()))


if __name__ == "__main__":
    main()
