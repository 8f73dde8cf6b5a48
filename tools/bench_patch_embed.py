#!/usr/bin/env python3
"""The tokenizer kernels alone at a BASELINE shape (default ViT-B/16 @ 224 Hilbert, 256 images, D = 768): tiled
(csrc/patch_embed_tiled.hip) against generic (csrc/patch_embed.hip), forward and backward, interleaved in one process.
    python tools/bench_patch_embed.py [curve img D B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sfcvit import ops  # noqa: E402
from sfcvit.curves import curve_table, hilbert_curve, z_curve  # noqa: E402
from sfcvit.tokenizers.embeddings import _pixel_table  # noqa: E402

curve, img, D, B = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("hilbert", 224, 768, 256)
flat = np.arange(img * img, dtype=np.int32) if curve == "raster" else curve_table({"hilbert": hilbert_curve, "z": z_curve}[curve], img)
pix_h = _pixel_table(flat, img, 1, 256)
pix = torch.from_numpy(pix_h).cuda()
desc = ops.tile_descriptor(pix_h, img, "cuda")
N = pix.shape[0]
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, 3, img, img, device="cuda", generator=g)
xb = x.bfloat16()
w = (torch.randn(D, 768, device="cuda", generator=g) / 28).bfloat16()
b = torch.zeros(D, device="cuda").bfloat16()
dy = torch.randn(B, N, D, device="cuda", generator=g).bfloat16()


order = torch.from_numpy(ops.gather_order(pix_h)).cuda()
tok = ops.gather_tokens(x, pix, desc, order)
dy2 = dy.view(B * N, D)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = {
    "fwd tiled (fp32 image)": lambda: ops.patch_embed_fwd(x, pix, w, b, desc),
    "fwd tiled (bf16 image)": lambda: ops.patch_embed_fwd(xb, pix, w, b, desc),
    "fwd generic (bf16 image)": lambda: ops.patch_embed_fwd(xb, pix, w, b, None),
    "fp32 -> bf16 cast of the image": lambda: x.bfloat16(),
    "gather, tile kernel (fp32 image -> bf16 tokens)": lambda: ops.gather_tokens(x, pix, desc, order),
    "gather, per-pixel kernel (fp32 image)": lambda: ops.gather_tokens(x, pix, None, order),
    "gather (bf16 image)": lambda: ops.gather_tokens(xb, pix),
    "projection GEMM on the tokens": lambda: ops.gemm(tok, w, bias=b),
    "two-stage fwd (gather + GEMM)": lambda: ops.gemm(ops.gather_tokens(x, pix, desc, order), w, bias=b),
    "two-stage bwd (dW GEMM + dbias)": lambda: (ops.gemm(dy2, tok, a_kmajor=True, b_kmajor=True), ops.colsum(dy2)),
    "bwd tiled (fp32 image)": lambda: ops.patch_embed_bwd(x, pix, dy, D, True, desc),
    "bwd generic (bf16 image)": lambda: ops.patch_embed_bwd(xb, pix, dy, D, True, None),
}
res = {k: [] for k in cases}
for rnd in range(3):
    for k, fn in cases.items():
        res[k].append(timeit(fn))
# the gather as the step sees it: the image and the token buffer cold (a 1-GB fill between launches evicts L2 and the
# 256-MB memory-side cache), one launch timed at a time
scrub = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
scrub.fill_(1.0)
for label, fn in (("tile kernel", lambda: ops.gather_tokens(x, pix, desc, order)), ("per-pixel kernel", lambda: ops.gather_tokens(x, pix, None, order)),
                  ("image cast", lambda: x.bfloat16())):
    ts = []
    for _ in range(7):
        scrub.sum()                                 # caches full of CLEAN lines: nothing to write back during the launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    med = sorted(ts)[len(ts) // 2]
    nbytes = x.numel() * 4 + tok.numel() * 2
    print(f"gather, {label}, caches full of clean lines: {med:7.1f} us = {nbytes / med / 1e6:.2f} TB/s of {nbytes / 1e6:.0f} MB")
for label, fn in (("tile kernel", lambda: ops.gather_tokens(x, pix, desc, order)), ("per-pixel kernel", lambda: ops.gather_tokens(x, pix, None, order)),
                  ("image cast", lambda: x.bfloat16())):
    ts = []
    for _ in range(7):
        scrub.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    med = sorted(ts)[len(ts) // 2]
    nbytes = x.numel() * 4 + tok.numel() * 2
    print(f"gather, {label}, caches full of dirty lines: {med:7.1f} us = {nbytes / med / 1e6:.2f} TB/s of {nbytes / 1e6:.0f} MB")
fl = 2.0 * B * N * 768 * D
print(f"{curve} {img}px D={D} B={B}: {fl / 1e9:.1f} GFLOP per pass, image {x.numel() * 4 / 1e6:.0f} MB fp32, tokens {B * N * D * 2 / 1e6:.0f} MB")
for k, v in res.items():
    med = sorted(v)[len(v) // 2]
    print(f"{k:34s} {med:8.1f} us   {fl / med / 1e6:7.1f} TFLOP/s")
