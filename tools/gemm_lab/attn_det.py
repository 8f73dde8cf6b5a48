import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch
from sfcvit import ops
for (B, N, H) in [(4, 196, 2), (16, 196, 12), (2, 64, 1), (3, 130, 4), (256, 196, 12)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).bfloat16()
    dout = torch.randn(B, N, H * 64, device="cuda", generator=g).bfloat16()
    o0, l0 = ops.attention_fwd(qkv, H)
    d0 = ops.attention_bwd(qkv, o0, l0, dout, H)
    nf = nb = 0
    for i in range(20):
        o, l = ops.attention_fwd(qkv, H)
        d = ops.attention_bwd(qkv, o0, l0, dout, H)
        nf += int(not (torch.equal(o, o0) and torch.equal(l, l0)))
        nb += int(not torch.equal(d, d0))
        if not torch.equal(o, o0) and nf == 1:
            bad = (o != o0).nonzero()
            print("  first fwd mismatch rows:", bad[:5].tolist(), "count", len(bad))
    print(B, N, H, "fwd mismatches", nf, "bwd mismatches", nb, "nan", bool(torch.isnan(o0.float()).any()))
