#!/usr/bin/env python3
"""Attention kernels alone at a BASELINE shape (default ViT-B/16 @ 224: B 256, N 196, H 12, hd 64), interleaved A/B of
the fused single-pass backward against the two-kernel form in ONE process (cdna_hip_programming.md rule 24).
    python tools/bench_attention.py [B N H] [--drop 0.1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import ops  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B, N, H = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (256, 196, 12)
p = float(sys.argv[sys.argv.index("--drop") + 1]) if "--drop" in sys.argv else 0.1
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).bfloat16()
dout = torch.randn(B, N, H * 64, device="cuda", generator=g).bfloat16()
out, lse = ops.attention_fwd(qkv, H, p, 5)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fl = 4.0 * B * H * N * N * 64
byt_f = B * N * 4 * H * 64 * 2
byt_b = B * N * 8 * H * 64 * 2
csum = torch.empty(3 * H * 64, device="cuda", dtype=torch.bfloat16)
rows = {"fwd": [], "fwd tiled (SFCVIT_ATTN_LONG=0)": [], "bwd fused": [], "bwd fused + in_proj bias column sums (as in the training step)": [],
        "bwd fused + column sums, dQ's from a separate pass (SFCVIT_ATTN_DQSUM=pass, round 3)": [],
        "bwd fused, two-slot start-up stagger (SFCVIT_ATTN_STAGGER_BWD=2,450)": [], "bwd fused, one workgroup per item (SFCVIT_ATTN_BWD_PERSIST=0)": [], "bwd two-kernel": [], "bwd two-kernel, tiled (SFCVIT_ATTN_LONG=0)": []}
for rnd in range(5):
    rows["fwd"].append(timeit(lambda: ops.attention_fwd(qkv, H, p, 5)))
    os.environ["SFCVIT_ATTN_LONG"] = "0"
    rows["fwd tiled (SFCVIT_ATTN_LONG=0)"].append(timeit(lambda: ops.attention_fwd(qkv, H, p, 5)))
    os.environ["SFCVIT_ATTN_LONG"] = "1"
    os.environ["SFCVIT_ATTN_BWD_FUSED"] = "1"
    rows["bwd fused"].append(timeit(lambda: ops.attention_bwd(qkv, out, lse, dout, H, p, 5)))
    rows["bwd fused + in_proj bias column sums (as in the training step)"].append(timeit(lambda: ops.attention_bwd(qkv, out, lse, dout, H, p, 5, colsum=csum)))
    os.environ["SFCVIT_ATTN_DQSUM"] = "pass"
    rows["bwd fused + column sums, dQ's from a separate pass (SFCVIT_ATTN_DQSUM=pass, round 3)"].append(timeit(lambda: ops.attention_bwd(qkv, out, lse, dout, H, p, 5, colsum=csum)))
    del os.environ["SFCVIT_ATTN_DQSUM"]
    os.environ["SFCVIT_ATTN_STAGGER_BWD"] = "2,450"
    rows["bwd fused, two-slot start-up stagger (SFCVIT_ATTN_STAGGER_BWD=2,450)"].append(timeit(lambda: ops.attention_bwd(qkv, out, lse, dout, H, p, 5)))
    del os.environ["SFCVIT_ATTN_STAGGER_BWD"]
    os.environ["SFCVIT_ATTN_BWD_PERSIST"] = "0"
    rows["bwd fused, one workgroup per item (SFCVIT_ATTN_BWD_PERSIST=0)"].append(timeit(lambda: ops.attention_bwd(qkv, out, lse, dout, H, p, 5)))
    os.environ["SFCVIT_ATTN_BWD_PERSIST"] = "1"
    os.environ["SFCVIT_ATTN_BWD_FUSED"] = "0"
    rows["bwd two-kernel"].append(timeit(lambda: ops.attention_bwd(qkv, out, lse, dout, H, p, 5)))
    os.environ["SFCVIT_ATTN_LONG"] = "0"
    rows["bwd two-kernel, tiled (SFCVIT_ATTN_LONG=0)"].append(timeit(lambda: ops.attention_bwd(qkv, out, lse, dout, H, p, 5)))
    os.environ["SFCVIT_ATTN_LONG"] = "1"
os.environ["SFCVIT_ATTN_BWD_FUSED"] = "1"
print(f"B={B} N={N} H={H} hd=64 dropout={p}")
for k, v in rows.items():
    v = sorted(v)
    med = v[len(v) // 2]
    f, by = (fl, byt_f) if k.startswith("fwd") else (2.5 * fl, byt_b)
    print(f"{k:86s} median {med:7.1f} us  min {v[0]:7.1f}   {f / med / 1e6:7.1f} TFLOP/s   {by / med / 1e3:7.1f} GB/s algorithmic "
          f"({by / med / 1e3 / 8000 * 100:4.1f} % of 8 TB/s)")
