"""Isolated timing of a 3x3 conv's weight gradient as nine single-tap problems vs three row problems (bf16, HIP events)."""
import sys
import torch
sys.path.insert(0, ".")
from ctrlora_amd import hip


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for B, H, C in [(8, 64, 320), (8, 32, 640), (8, 16, 1280), (8, 8, 1280)]:
    x = torch.randn(B * H * H, C, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B * H * H, C, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(C, 9 * C, device="cuda")
    p9 = [(dy, x, dW[:, t * C:(t + 1) * C], 1.0, (t, H, H, H, H, 1, 1)) for t in range(9)]
    p3 = [(dy, x, dW[:, 3 * k * C:(3 * k + 1) * C], 1.0, (16 + k, H, H, H, H, 1, 1)) for k in range(3)]
    t9, t3 = timed(lambda: hip.weight_grad_tn_group(p9)), timed(lambda: hip.weight_grad_tn_group(p3))
    fl = 2 * B * H * H * C * C * 9
    print(f"B={B} {H}x{H} C={C}: nine single taps {t9:7.1f} us ({fl / t9 / 1e6:5.0f} TF/s) | three rows {t3:7.1f} us ({fl / t3 / 1e6:5.0f} TF/s)  x{t9 / t3:.2f}", flush=True)
