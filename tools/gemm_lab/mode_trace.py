import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))
import torch
from sfcvit import ops
M, N, K = 50176, 2304, 768
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
b = torch.randn(N, K, device="cuda", generator=g).bfloat16()
for mode in (4, 5, 2, 3, 1):
    for _ in range(3):
        ops.gemm(a, b, force_generic=mode)
    torch.cuda.synchronize()
