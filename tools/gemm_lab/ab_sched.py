#!/usr/bin/env python3
"""A/B of the persistent GEMM's schedules in ONE process (SFCVIT_GEMM_SCHED is read at every launch): 1 = two phases with the
epilogue between tiles, 2 = two phases with the epilogue dealt out to the load sections of the boundary k-tiles -- on the
forward / dX GEMMs of a ViT-B (or, with M D F given, any) layer at batch 256 with their real epilogues.  Results must agree
bit for bit (same accumulation order, same epilogue arithmetic).  AB_SCHEDS="1,2" (default), AB_FORCE=1 lets the
side-operand variants take schedule 2 too (lab builds that do not spill only).
Usage: ab_sched.py [M [D F]]      (with SFCVIT_LIB=<variant .so> to time another build, e.g. the LDS-patch exchange)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch  # noqa: E402
from sfcvit import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 50176
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
FF = int(sys.argv[3]) if len(sys.argv) > 3 else 4 * D
F3 = 3 * D
if os.environ.get("AB_FORCE") == "1":
    os.environ["SFCVIT_GEMM_SCHED_FORCE"] = "1"
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()      # noqa: E731
x, wqkv, wo, w1, w2 = r(M, D), r(F3, D) * 0.05, r(D, D) * 0.05, r(FF, D) * 0.05, r(D, FF) * 0.03
bq, bo, b1, b2 = r(F3), r(D), r(FF), r(D)
h = torch.relu(r(M, FF))
res = r(M, D)
dyq, dyf = r(M, F3), r(M, D)
bits = torch.empty((M, FF // 8), device="cuda", dtype=torch.uint8)
os.environ["SFCVIT_GEMM_SCHED"] = "1"
ops.gemm(x, w1, bias=b1, act=ops.ACT_RELU, dropout_p=0.1, dropout_seed=3, actmask=bits)
w1t, w2t, wqkvt = w1.t().contiguous(), w2.t().contiguous(), wqkv.t().contiguous()


def ffn1():
    y = ops.gemm(x, w1, bias=b1, act=ops.ACT_RELU, dropout_p=0.1, dropout_seed=3, actmask=bits)
    return y


def ffn2dx():
    out = ops.gemm(dyf, w2t, aux_in=h, dact=ops.ACT_RELU, dact_scale=1 / 0.9, colsum=True, actmask=bits)
    return torch.cat([out[0].float().flatten(), out[1].float().flatten()])


CASES = {
    "qkv fwd   <0>  N%d K%d" % (F3, D): (lambda: ops.gemm(x, wqkv, bias=bq), 2.0 * M * F3 * D),
    "out fwd   <6>  N%d K%d" % (D, D): (lambda: ops.gemm(x, wo, bias=bo, residual=res, dropout_p=0.1, dropout_seed=5), 2.0 * M * D * D),
    "ffn1 fwd  <35> N%d K%d" % (FF, D): (ffn1, 2.0 * M * FF * D),
    "ffn2 fwd  <6>  N%d K%d" % (D, FF): (lambda: ops.gemm(h, w2, bias=b2, residual=res, dropout_p=0.1, dropout_seed=7), 2.0 * M * D * FF),
    "ffn2 dX   <56> N%d K%d" % (FF, D): (ffn2dx, 2.0 * M * FF * D),
    "ffn1 dX   <4>  N%d K%d" % (D, FF): (lambda: ops.gemm(h, w1t, residual=res), 2.0 * M * D * FF),
    "qkv dX    <4>  N%d K%d" % (D, F3): (lambda: ops.gemm(dyq, wqkvt, residual=res), 2.0 * M * D * F3),
    "out dX    <0>  N%d K%d" % (D, D): (lambda: ops.gemm(dyf, wo.t().contiguous()), 2.0 * M * D * D),
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


scheds = os.environ.get("AB_SCHEDS", "1,2").split(",")
print(f"M {M} D {D} F {FF}; lib {os.environ.get('SFCVIT_LIB', 'in-tree')}", flush=True)
for name, (fn, fl) in CASES.items():
    outs, names, t = {}, {}, {m: [] for m in scheds}
    for m in scheds:
        os.environ["SFCVIT_GEMM_SCHED"] = m
        bits.zero_()
        outs[m] = (fn().clone(), bits.clone())
        names[m] = ops.last_gemm_kernel()
    same = all(torch.equal(outs[m][0], outs[scheds[0]][0]) and torch.equal(outs[m][1], outs[scheds[0]][1]) for m in scheds)
    for rnd in range(7):
        for m in scheds:
            os.environ["SFCVIT_GEMM_SCHED"] = m
            t[m].append(timeit(fn))
    med = {m: sorted(t[m])[3] for m in scheds}
    line = "  ".join(f"[{names[m]}] {med[m]:7.1f} us {fl / med[m] / 1e6:5.0f} TF" for m in scheds)
    print(f"{name:28s} {line}   {(med[scheds[0]] / med[scheds[-1]] - 1) * 100:+5.1f} %   bit-identical: {same}", flush=True)
