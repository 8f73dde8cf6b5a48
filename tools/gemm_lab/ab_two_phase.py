#!/usr/bin/env python3
"""A/B of the persistent GEMM's k-tile schedules in ONE process: four phases (8 barriers per k-tile, the product default)
against two phases (SFCVIT_GEMM_2PHASE=1: 4 barriers), on the forward / dX GEMMs of a ViT-B layer at batch 256 with their
real epilogues.  The two must agree bit for bit (same accumulation order)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 50176
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()      # noqa: E731
D, F3, FF = 768, 2304, 3072
x, wqkv, wo, w1, w2 = r(M, D), r(F3, D) * 0.05, r(D, D) * 0.05, r(FF, D) * 0.05, r(D, FF) * 0.03
bq, bo, b1, b2 = r(F3), r(D), r(FF), r(D)
h = torch.relu(r(M, FF))
res = r(M, D)
dyq, dyf = r(M, F3), r(M, D)
bits = torch.empty((M, FF // 8), device="cuda", dtype=torch.uint8)
ops.gemm(x, w1, bias=b1, act=ops.ACT_RELU, dropout_p=0.1, dropout_seed=3, actmask=bits)
w1t, w2t, wqkvt = w1.t().contiguous(), w2.t().contiguous(), wqkv.t().contiguous()
CASES = {
    "qkv fwd      <.,0>  N2304 K768 ": (lambda: ops.gemm(x, wqkv, bias=bq), 2.0 * M * F3 * D),
    "out fwd      <.,6>  N768  K768 ": (lambda: ops.gemm(x, wo, bias=bo, residual=res, dropout_p=0.1, dropout_seed=5), 2.0 * M * D * D),
    "ffn1 fwd     <.,35> N3072 K768 ": (lambda: ops.gemm(x, w1, bias=b1, act=ops.ACT_RELU, dropout_p=0.1, dropout_seed=3, actmask=bits), 2.0 * M * FF * D),
    "ffn2 fwd     <.,6>  N768  K3072": (lambda: ops.gemm(h, w2, bias=b2, residual=res, dropout_p=0.1, dropout_seed=7), 2.0 * M * D * FF),
    "ffn2 dX      <.,56> N3072 K768 ": (lambda: ops.gemm(dyf, w2t, aux_in=h, dact=ops.ACT_RELU, dact_scale=1 / 0.9, colsum=True, actmask=bits)[0], 2.0 * M * FF * D),
    "ffn1 dX      <.,4>  N768  K3072": (lambda: ops.gemm(h, w1t, residual=res), 2.0 * M * D * FF),
    "qkv dX       <.,4>  N768  K2304": (lambda: ops.gemm(dyq, wqkvt, residual=res), 2.0 * M * D * F3),
    "out dW   km  768 x 768         ": (lambda: ops.gemm(dyf, x, a_kmajor=True, b_kmajor=True), 2.0 * M * D * D),
    "qkv dW   km  2304 x 768        ": (lambda: ops.gemm(dyq, x, a_kmajor=True, b_kmajor=True), 2.0 * M * F3 * D),
    "ffn1 dW  km  3072 x 768        ": (lambda: ops.gemm(h, x, a_kmajor=True, b_kmajor=True), 2.0 * M * FF * D),
    "ffn2 dW  km  768 x 3072        ": (lambda: ops.gemm(dyf, h, a_kmajor=True, b_kmajor=True), 2.0 * M * FF * D),
}


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


STAG = os.environ.get("AB_STAGGER")           # e.g. "4,50;8,30": compare start-up staggers instead of the schedules
if STAG:
    for name, (fn, fl) in CASES.items():
        if " km " in name:
            continue
        modes = ["1,0"] + STAG.split(";")
        t = {m: [] for m in modes}
        for rnd in range(7):
            for m in modes:
                os.environ["SFCVIT_GEMM_STAGGER"] = m
                t[m].append(timeit(fn))
        print(name, "  ".join(f"[{m}] {sorted(t[m])[3]:7.1f} us" for m in modes), flush=True)
    sys.exit(0)
for name, (fn, fl) in CASES.items():
    os.environ["SFCVIT_GEMM_2PHASE"] = "0"
    ref = fn().clone()
    os.environ["SFCVIT_GEMM_2PHASE"] = "1"
    got = fn()
    same = torch.equal(ref, got)
    t = {"0": [], "1": []}
    for rnd in range(7):
        for mode in ("0", "1"):
            os.environ["SFCVIT_GEMM_2PHASE"] = mode
            t[mode].append(timeit(fn))
    a, b = sorted(t["0"])[3], sorted(t["1"])[3]
    print(f"{name}  4-phase {a:7.1f} us {fl / a / 1e6:6.0f} TF   2-phase {b:7.1f} us {fl / b / 1e6:6.0f} TF   {(a / b - 1) * 100:+5.1f} %   bit-identical: {same}", flush=True)
