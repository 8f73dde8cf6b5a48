#!/usr/bin/env python3
"""Where do two schedules of the persistent GEMM disagree?  Prints the mismatching (tile, wave group, row fragment, column block) set."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch
from sfcvit import ops
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (9472, 1792, 256)
fg = int(sys.argv[4]) if len(sys.argv) > 4 else 8
BM = {8: 256, 9: 224, 10: 192}[fg]
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
w = torch.randn(N, K, device="cuda", generator=g).bfloat16()
b = torch.randn(N, device="cuda", generator=g).bfloat16()
for trial in range(4):
    os.environ["SFCVIT_GEMM_SCHED"] = "1"
    ref = ops.gemm(a, w, bias=b, force_generic=fg)
    os.environ["SFCVIT_GEMM_SCHED"] = "2"
    got = ops.gemm(a, w, bias=b, force_generic=fg)
    name = ops.last_gemm_kernel()
    bad = (ref != got).nonzero()
    print(f"trial {trial} {name}: {bad.shape[0]} mismatches", flush=True)
    if bad.shape[0]:
        r, c = bad[:, 0], bad[:, 1]
        keys = {}
        for rr, cc in zip(r.tolist(), c.tolist()):
            k = (rr // BM, cc // 256, (rr % BM) // (BM // 2), ((rr % BM) % (BM // 2)) // 16, (cc % 256) // 64)
            keys[k] = keys.get(k, 0) + 1
        for k, v in sorted(keys.items())[:6]:
            print("  row tile %d col tile %d group %d row frag %d wave col %d: %d" % (*k, v))
        k0 = sorted(keys.items())[0][0]
        sel = [(rr, cc) for rr, cc in zip(r.tolist(), c.tolist()) if (rr // BM, cc // 256, (rr % BM) // (BM // 2), ((rr % BM) % (BM // 2)) // 16, (cc % 256) // 64) == k0]
        print("   first group, (row in fragment, col in wave block): got / ref", [((rr % BM) % 16, cc % 64, float(got[rr, cc]), float(ref[rr, cc])) for rr, cc in sel][:16])
