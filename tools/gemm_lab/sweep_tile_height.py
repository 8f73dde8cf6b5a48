#!/usr/bin/env python3
"""Which row-tile height (256 / 224 / 192 rows) is fastest for each forward / dX GEMM of a ViT-B layer at M = 50 176, with the
real epilogues?  `force_generic` 8 / 9 / 10 pins the height (0 = the dispatcher's own choice); one process, interleaved."""
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
    "qkv fwd      N2304 K768 ": lambda f: ops.gemm(x, wqkv, bias=bq, force_generic=f),
    "out fwd      N768  K768 ": lambda f: ops.gemm(x, wo, bias=bo, residual=res, dropout_p=0.1, dropout_seed=5, force_generic=f),
    "ffn1 fwd     N3072 K768 ": lambda f: ops.gemm(x, w1, bias=b1, act=ops.ACT_RELU, dropout_p=0.1, dropout_seed=3, actmask=bits, force_generic=f),
    "ffn2 fwd     N768  K3072": lambda f: ops.gemm(h, w2, bias=b2, residual=res, dropout_p=0.1, dropout_seed=7, force_generic=f),
    "ffn2 dX      N3072 K768 ": lambda f: ops.gemm(dyf, w2t, aux_in=h, dact=ops.ACT_RELU, dact_scale=1 / 0.9, colsum=True, actmask=bits, force_generic=f)[0],
    "ffn1 dX      N768  K3072": lambda f: ops.gemm(h, w1t, residual=res, force_generic=f),
    "qkv dX       N768  K2304": lambda f: ops.gemm(dyq, wqkvt, residual=res, force_generic=f),
    "pe / mixer   N768  K768 (bias only)": lambda f: ops.gemm(x, wo, bias=bo, force_generic=f),
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


for name, fn in CASES.items():
    modes = {0: "auto", 8: "256", 9: "224", 10: "192"}
    t = {m: [] for m in modes}
    chosen = None
    for rnd in range(5):
        for m in modes:
            t[m].append(timeit(lambda: fn(m)))
            if m == 0 and chosen is None:
                chosen = ops.last_gemm_kernel()
    print(f"{name:38s}", "  ".join(f"[{modes[m]}] {sorted(t[m])[2]:7.1f} us" for m in modes), "  auto =", chosen, flush=True)
