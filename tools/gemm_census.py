"""Census of the dense contractions of one training step (and one DDIM forward): records every
hip.gemm call's shape, then times each unique shape in isolation with HIP events.
Usage (GPU box): python tools/gemm_census.py [--ddim] > gpurun_out/gemm_census.txt"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ctrlora_amd import hip  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = bench.build_model("ctrlora_finetune_sd15_rank128.yaml", 0).to(device).train()
    model.set_engine_dtype(torch.bfloat16)
    model.learning_rate = 1e-5
    opt = model.configure_optimizers()
    data = bench.synth(8, 64, model.control_model.context_dim, device, 1234, 1)

    def step():
        opt.zero_grad()
        cond = {"c_crossattn": [data["ctx"][0]], "c_concat": [data["hint"][0]]}
        loss, _ = model.p_losses(data["z"][0], cond, data["t"][0], noise=data["noise"][0])
        loss.backward()
        opt.step()

    ddim = "--ddim" in sys.argv
    if ddim:   # one CFG-batched forward at B = 2 x 16 through the inference engine
        del model, opt
        torch.cuda.empty_cache()
        minf = bench.build_model("inference/ctrlora_sd15_rank128_1lora.yaml", 0).to(device).eval()
        minf.set_engine_dtype(torch.bfloat16)
        d32 = bench.synth(32, 64, minf.control_model.context_dim, device, 7, 1)

        def step():   # noqa: F811
            with torch.no_grad():
                minf.apply_model(d32["z"][0], d32["t"][0], {"c_crossattn": [d32["ctx"][0]], "c_concat": [d32["hint"][0]]})

    step()
    torch.cuda.synchronize()
    calls = collections.OrderedDict()
    orig = hip.gemm

    def rec(a1, w1, out, **kw):
        M = out.shape[0] if kw.get("M") is None else kw["M"]
        N = out.shape[1] if kw.get("N") is None else kw["N"]
        k1 = a1.shape[1] if kw.get("k1") is None else kw["k1"]
        mode = kw.get("mode", hip.LINEAR)
        a2 = kw.get("a2")
        key = (mode, M, N, k1, 0 if a2 is None else a2.shape[1], kw.get("conv"), kw.get("residual") is not None,
               bool(kw.get("atomic", False)), bool(kw.get("out_f32", False)))
        if key not in calls:
            calls[key] = [0, (a1, w1, out, kw)]
        calls[key][0] += 1
        return orig(a1, w1, out, **kw)

    hip.gemm = rec
    import ctrlora_amd.engine.blocks as blocks
    import ctrlora_amd.engine.nets as nets
    step()
    torch.cuda.synchronize()
    hip.gemm = orig

    rows = []
    for key, (cnt, (a1, w1, out, kw)) in calls.items():
        mode, M, N, k1, k2, conv, res, atomic, of32 = key
        if atomic:
            continue
        for _ in range(3):
            orig(a1, w1, out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            orig(a1, w1, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        taps = 1 if mode == hip.LINEAR else 9
        fl = 2.0 * M * N * (taps * k1 + k2)
        rows.append((cnt * us, cnt, us, fl / us * 1e-6, key))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"unique shapes {len(rows)}, total {tot/1e3:.2f} ms per step (isolated timings, sum over calls)")
    print(f"{'tot_us':>9} {'n':>4} {'us':>8} {'TF/s':>7}  mode M N K1 K2 conv resid f32out")
    for t, cnt, us, tf, key in rows:
        mode, M, N, k1, k2, conv, res, atomic, of32 = key
        print(f"{t:9.0f} {cnt:4d} {us:8.1f} {tf:7.1f}  {mode} {M} {N} {k1} {k2} {conv} {int(res)} {int(of32)}")


if __name__ == "__main__":
    main()
