import os, sys, math
sys.path.insert(0, "space-filling-curves-for-vision-transformers_amd")
import torch
from sfcvit import ops
bf = lambda t: t.to(torch.bfloat16)
g = torch.Generator(device="cuda").manual_seed(5)
B, N, D, H = 64, 196, 768, 12
qkv = bf(torch.randn(B, N, 3 * D, device="cuda", generator=g))
q, k, v = qkv.float().split(D, dim=-1)
sp = lambda t: t.reshape(B, N, H, 64).transpose(1, 2)
P = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / 8.0, -1)
for p in (1e-6, 0.001, 0.1, 0.5):
    for seed in (11, 12):
        ma = ops.dropout_mask(B * H * N, N, p, seed).float().view(B, H, N, N)
        ref = ((P * ma) @ sp(v)).transpose(1, 2).reshape(B, N, D)
        out, lse = ops.attention_fwd(qkv, H, p, seed)
        bad = ((out.float() - ref).abs() > (ref.abs() / 32 + ref.pow(2).mean().sqrt() / 24))
        rows = bad.view(B, N, H, 64).any(-1)
        idx = rows.nonzero()
        print(f"p {p} seed {seed} [{ops.last_attn_kernel()}]: {int(bad.sum())} elements off in {idx.shape[0]} rows; q//16 {torch.bincount(idx[:,1]//16, minlength=13).tolist() if idx.shape[0] else []}", flush=True)
# smaller batch: 21 batches = 252 workgroups (one per CU)
for Bs in (21, 22, 32):
    ma = ops.dropout_mask(Bs * H * N, N, 0.1, 11).float().view(Bs, H, N, N)
    ref = ((P[:Bs] * ma) @ sp(v)[:Bs]).transpose(1, 2).reshape(Bs, N, D)
    out, lse = ops.attention_fwd(qkv[:Bs].contiguous(), H, 0.1, 11)
    bad = ((out.float() - ref).abs() > (ref.abs() / 32 + ref.pow(2).mean().sqrt() / 24))
    print(f"batch {Bs}: {int(bad.sum())} off", flush=True)
print("---- which key fragment of a bad row is wrong, and how")
p, seed = 0.1, 11
ma = ops.dropout_mask(B * H * N, N, p, seed).float().view(B, H, N, N)
ref = (P * ma) @ sp(v)
out, lse = ops.attention_fwd(qkv, H, p, seed)
got = out.float().view(B, N, H, 64).transpose(1, 2)
bad = ((got - ref).abs() > (ref.abs() / 32 + ref.pow(2).mean().sqrt() / 24)).any(-1)
idx = bad.nonzero()
for (b0, h0, q0) in idx[::max(1, idx.shape[0] // 6)][:6].tolist():
    Pr, V, m0 = P[b0, h0, q0].double(), sp(v)[b0, h0].double(), ma[b0, h0, q0].double()
    r = got[b0, h0, q0].double() - (Pr * m0) @ V           # residual to explain
    best = None
    for kf in range(13):
        sl = slice(16 * kf, min(16 * kf + 16, N))
        A = (Pr[sl, None] * V[sl]).t()                    # [64, <=16]
        sol = torch.linalg.lstsq(A, r[:, None]).solution[:, 0]
        res = float((A @ sol - r).abs().max())
        if best is None or res < best[0]: best = (res, kf, (m0[sl] + sol).tolist())
    print(f"row (b {b0} h {h0} q {q0}): residual {float(r.abs().max()):.4f}; best single key fragment {best[1]} leaves {best[0]:.5f}; its effective keep factors: {[round(x, 2) for x in best[2]]}; reference: {[round(float(x), 2) for x in m0[16*best[1]:16*best[1]+16]]}")
