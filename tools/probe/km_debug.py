import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "space-filling-curves-for-vision-transformers_amd"))
import torch
from sfcvit import ops
torch.manual_seed(0)
for (M, N, K, sk) in [(256, 256, 256, 2), (256, 256, 512, 2), (256, 256, 4096, 4)]:
    g = torch.Generator(device="cuda").manual_seed(31)
    a = torch.randn(K, M, device="cuda", generator=g).bfloat16()
    b = torch.randn(K, N, device="cuda", generator=g).bfloat16()
    ref = a.float().t() @ b.float()
    c = ops.gemm(a, b, a_kmajor=True, b_kmajor=True, splitk=sk, out_f32=True)
    bad = (c - ref).abs() > 0.05 * ref.abs() + 0.5
    print(M, N, K, sk, "bad", int(bad.sum()))
    rows = bad.any(1).nonzero().flatten().tolist()
    cols = bad.any(0).nonzero().flatten().tolist()
    print(" bad rows", rows[:40], len(rows))
    print(" bad cols", cols[:40], len(cols))
    # which k contributions are missing: probe with one-hot k
    for k in (0, 5, 33, 70, 130, K - 1):
        a1 = torch.zeros_like(a); a1[k] = a[k]
        c1 = ops.gemm(a1, b, a_kmajor=True, b_kmajor=True, splitk=sk, out_f32=True)
        r1 = a1.float().t() @ b.float()
        print("  k", k, "bad", int(((c1 - r1).abs() > 0.02).sum()))
