"""Time the phase-decomposed Upsample conv / Downsample data gradient against the nine-tap modes at the shapes of the training
step (B = 8) and of the DDIM step (B = 32 = 16 x 2 guidance passes), HIP events, bf16.  Usage: python tools/time_conv_phase.py
Prints one line per (shape, form, tile configuration); cfg -1 = the launcher's rule, 8 / 9 = 256 x {160, 128} tiles,
10 / 11 = 128 x {160, 128}."""
import sys

import torch

sys.path.insert(0, ".")
from ctrlora_amd import hip                                   # noqa: E402
from ctrlora_amd.engine import blocks                          # noqa: E402
from ctrlora_amd.engine.packing import Conv3W                  # noqa: E402


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    dt = torch.bfloat16
    ctx = blocks.Ctx(dt, torch.device("cuda"), False)
    L = hip.lib()
    rows = []
    for kind, B, H, C in [("up2", 8, 8, 1280), ("up2", 8, 16, 1280), ("up2", 8, 32, 640),
                          ("up2", 32, 8, 1280), ("up2", 32, 16, 1280), ("up2", 32, 32, 640),
                          ("t2", 8, 32, 320), ("t2", 8, 16, 640), ("t2", 8, 8, 1280),
                          ("up2d", 8, 8, 1280), ("up2d", 8, 16, 1280), ("up2d", 8, 32, 640)]:
        w = torch.randn(C, C, 3, 3) * 0.01
        cw = Conv3W(w, torch.zeros(C), dt, "cuda", True)
        x = torch.randn(B * H * H, C, device="cuda").to(dt)
        out = torch.empty(4 * B * H * H, C, dtype=dt, device="cuda")
        if kind == "up2":
            fn = lambda: blocks.conv3_fwd(ctx, cw, x, B, H, H, mode=hip.CONV_UP2, out=out)
            macs9 = 4 * B * H * H * C * 9 * C
        elif kind == "t2":
            fn = lambda: blocks.conv3_bwd_data(ctx, cw, x, B, H, H, fwd_mode=hip.CONV_S2, out=out)
            macs9 = 4 * B * H * H * C * 9 * C
        else:     # data gradient of the Upsample conv: dy on the 2H x 2H grid -> dx on H x H
            dyu = torch.randn(4 * B * H * H, C, device="cuda").to(dt)
            dxl = torch.empty(B * H * H, C, dtype=dt, device="cuda")
            fn = lambda: blocks.conv3_bwd_data(ctx, cw, dyu, B, 2 * H, 2 * H, fwd_mode=hip.CONV_UP2, out=dxl)
            macs9 = 4 * B * H * H * C * 9 * C
        blocks.CONV_PHASE = False
        t_old = timed(fn)
        blocks.CONV_PHASE = True
        res = {}
        for cfg in (-1, 8, 10):
            L.cl_gemm_force_config(cfg)
            try:
                res[cfg] = timed(fn)
            except Exception as e:           # a configuration the shape does not admit
                res[cfg] = float("nan")
            L.cl_gemm_force_config(-1)
        best = min(v for v in res.values() if v == v)
        print(f"{kind} B={B} src {H}x{H} C={C}: nine-tap {t_old:7.1f} us ({2 * macs9 / t_old / 1e6:6.0f} TF/s nominal) | phase rule {res[-1]:7.1f}"
              f"  256-row {res[8]:7.1f}  128-row {res[10]:7.1f}  -> x{t_old / best:.2f}", flush=True)
        rows.append((kind, B, H, C, t_old, res))
    return rows


if __name__ == "__main__":
    main()
