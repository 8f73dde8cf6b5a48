#!/usr/bin/env python3
"""A few launches of attention fwd/bwd at the ViT-B shape for rocprofv3 runs. usage: prof_attn.py [dropout_p]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch
from sfcvit import ops
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
B, N, H = 256, 196, 12
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).bfloat16()
dout = torch.randn(B, N, H * 64, device="cuda", generator=g).bfloat16()
for _ in range(3):
    out, lse = ops.attention_fwd(qkv, H, p, 123)
    dq = ops.attention_bwd(qkv, out, lse, dout, H, p, 123)
torch.cuda.synchronize()
