"""HBM traffic check of the norm family under rocprofv3 PMC (FETCH_SIZE / WRITE_SIZE in separate passes): GroupNorm(+SiLU) and
LayerNorm forward at the 64x64 level (B = 8, C = 320) -- the two heaviest signatures of bench.py's
roofline.norm_elementwise_family -- launched a few times each.  Not part of the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlora_amd import hip  # noqa: E402

B, HW, C = 8, 4096, 320
x = torch.randn(B * HW, C, device="cuda").to(torch.bfloat16)
y = torch.empty_like(x)
gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
stats = torch.empty(B, 32, 2, device="cuda"); ws = torch.empty(hip.groupnorm_ws(B, HW, C), device="cuda")
lst = torch.empty(B * HW, 2, device="cuda")
for _ in range(6):
    hip.groupnorm_fwd(x, y, gamma, beta, B, HW, 1e-5, True, stats, ws)
    hip.layernorm_fwd(x, y, gamma, beta, 1e-5, lst)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", 2 * x.numel() * 2)
