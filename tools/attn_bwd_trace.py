#!/usr/bin/env python3
"""Shader-clock stamps of workgroup 5 of the persistent fused attention backward (a -DSFCVIT_ATTN_TRACE build loaded through
SFCVIT_LIB): item start, the barrier that ends each of the 7 steps, the last (dQ-only) step, the end of the post-loop code, the
end of the item switch.  Prints the cycle budget of an item."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import numpy as np, torch
from sfcvit import ops, _lib
B, N, H, p = 256, 196, 12, 0.1
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).bfloat16()
dout = torch.randn(B, N, H * 64, device="cuda", generator=g).bfloat16()
out, lse = ops.attention_fwd(qkv, H, p, 5)
for _ in range(3):
    ops.attention_bwd(qkv, out, lse, dout, H, p, 5)
torch.cuda.synchronize()
buf = np.zeros(512, dtype=np.uint64)
lib = ctypes.CDLL(os.environ["SFCVIT_LIB"])
assert lib.sfcvit_debug_attn_trace(ctypes.c_void_p(buf.ctypes.data)) == 0
t = buf.astype(np.int64).reshape(32, 16)
for it in range(1, 11):
    r = t[it]
    steps = np.diff(r[0:8])
    print(f"item {it}: steps {steps.tolist()}  last step: dQ + fetch issue {int(r[8]-r[7])}, wait + barrier {int(r[9]-r[8])}  post-loop {int(r[10]-r[9])}  switch: first barrier {int(r[12]-r[10])}, delta {int(r[13]-r[12])}, second barrier {int(r[11]-r[13])}  | item total {int(t[it+1][0]-r[0])}")

w0 = buf[288:304].astype(np.int64); w1 = buf[256:272].astype(np.int64)
print("item 1, per wave: clocks from the last step's barrier to the arrival at the item switch's first barrier:", (w1 - w0).tolist())
