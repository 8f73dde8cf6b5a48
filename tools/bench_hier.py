#!/usr/bin/env python3
"""The hierarchical tokenizer alone at the reference's default shape (main.py:269-274: HierarchicalMortonEmbedding(32, 3,
[16, 4, 1], 256), CIFAR batch 512): the one-kernel forward (csrc/hier_tokenizer.hip) against the composed path (three
level kernels + cat + fusion GEMM), forward and forward + backward, interleaved in one process.
    python tools/bench_hier.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit.tokenizers import HierarchicalMortonEmbedding  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
mod = HierarchicalMortonEmbedding(32, 3, [16, 4, 1], 256).to("cuda", dtype=torch.bfloat16)
x = torch.randn(B, 3, 32, 32, device="cuda")
assert mod._fusable(x)
r = torch.randn(B, 64, 768, device="cuda").bfloat16()


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


def fwd(f):
    with torch.no_grad():
        return f(x)


def fwd_bwd(f):
    for p in mod.parameters():
        p.grad = None
    f(x).backward(r)


cases = {
    "forward, one kernel": lambda: fwd(lambda t: mod(t, one_kernel=True)),
    "forward, levels+concat kernel, fusion GEMM": lambda: fwd(lambda t: mod(t, one_kernel=False)),
    "forward, composed (3 level kernels + cat + GEMM)": lambda: fwd(mod.forward_unfused),
    "forward + backward, levels+concat kernel + GEMM": lambda: fwd_bwd(lambda t: mod(t, one_kernel=False)),
    "forward + backward, composed": lambda: fwd_bwd(mod.forward_unfused),
}
res = {k: [] for k in cases}
for rnd in range(3):
    for k, fn in cases.items():
        res[k].append(timeit(fn))
M, E = B * 64, 768
fl = 2.0 * M * E * E + 3 * 2.0 * M * 48 * 256
print(f"batch {B}: {M} token rows, {fl / 1e9:.1f} GFLOP forward; host-inclusive times (eager launches)")
for k, v in res.items():
    med = sorted(v)[len(v) // 2]
    print(f"{k:52s} {med:8.1f} us")
