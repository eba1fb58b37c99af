"""Downsample data gradient (nine-tap T2 vs phase form T2P) with and without an accumulate operand, isolated, bf16 -- the in-step
duration of these launches (126 us in profiles/r06_final3/train_shapes_in_step.txt against 41 us here) is the overlap with the
weight-gradient stream, not the residual read."""
import sys, torch
sys.path.insert(0, ".")
from ctrlora_amd import hip
from ctrlora_amd.engine import blocks
from ctrlora_amd.engine.packing import Conv3W
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
dt = torch.bfloat16
ctx = blocks.Ctx(dt, torch.device("cuda"), False)
for B, H, C in [(8, 32, 320), (8, 16, 640)]:
    cw = Conv3W(torch.randn(C, C, 3, 3) * 0.01, torch.zeros(C), dt, "cuda", True)
    dy = torch.randn(B * H * H, C, device="cuda").to(dt)
    out = torch.empty(4 * B * H * H, C, dtype=dt, device="cuda")
    acc = torch.randn(4 * B * H * H, C, device="cuda").to(dt)
    for ph in (True, False):
        blocks.CONV_PHASE = ph
        t0 = timed(lambda: blocks.conv3_bwd_data(ctx, cw, dy, B, H, H, fwd_mode=hip.CONV_S2, out=out))
        t1 = timed(lambda: blocks.conv3_bwd_data(ctx, cw, dy, B, H, H, fwd_mode=hip.CONV_S2, out=out, accum=acc))
        t2 = timed(lambda: blocks.conv3_bwd_data(ctx, cw, dy, B, H, H, fwd_mode=hip.CONV_S2, out=acc, accum=acc))
        print(f"B={B} {H}x{H} C={C} phase={ph}: plain {t0:.1f} us, +accum {t1:.1f} us, in-place accum {t2:.1f} us", flush=True)
