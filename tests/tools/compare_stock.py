"""'Same silicon, stock kernels' + same-precision comparator (SURVEY.md 8d, section 7 precision contract (ii)).

The oracle -- the functional restatement of the reference modules in plain torch ops (oracle/ref_model.py) --
is run on the GPU through PyTorch-ROCm's own kernels (rocBLAS / hipBLASLt / MIOpen / ATen) in fp32 and under
torch.autocast(bfloat16) (what the reference's `precision: bf16` trainer flag does), next to the HIP engine in its
fp32 parity mode and its bf16 mode, on the bench's shape (rank 128, latent 64x64):

  * three-way parity: rel-L2 of eps and of every trainable gradient (max / median over the 246 tensors) for
    {engine bf16, stock bf16-autocast, engine fp32} against stock fp32;
  * speed: forward + backward ms/step and images/s of the stock-kernel path (fp32 and bf16 autocast) at batch 8.

Test / measurement infrastructure only (imports oracle/):
    python tests/tools/compare_stock.py [--batch 8] [--steps 5] [--out gpurun_out/compare_precision.json]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def summarize(grads, ref):
    errs = sorted(((rel(grads[k], ref[k]), k) for k in ref), reverse=True)
    return dict(grad_max=errs[0][0], grad_max_name=errs[0][1], grad_median=errs[len(errs) // 2][0],
                grad_p90=errs[len(errs) // 10][0], n=len(errs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--parity-batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--rank-lora", type=int, default=128)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from ctrlora_amd.engine import CtrLoRAEngine, NetCfg
    from oracle import arch, ref_model as R
    dev = torch.device("cuda")
    cfg = arch.ArchCfg(lora_rank=args.rank_lora)
    ncfg = NetCfg(cfg.in_channels, cfg.out_channels, cfg.model_channels, cfg.channel_mult, cfg.num_res_blocks,
                  cfg.attention_resolutions, cfg.num_heads, cfg.context_dim)
    sd_cn_cpu = arch.make_state(arch.controlnet_shapes(cfg), 0)
    sd_un_cpu = arch.make_state(arch.unet_shapes(cfg), 0)
    sd_cn = {k: v.to(dev) for k, v in sd_cn_cpu.items()}
    sd_un = {k: v.to(dev) for k, v in sd_un_cpu.items()}
    train = [k for k in sd_cn if arch.is_trainable(k)]
    for k in train:
        sd_cn[k].requires_grad_(True)
    sched = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in R.make_schedule().items()}

    def inputs(B):
        g = torch.Generator().manual_seed(0)
        z, hint, noise = (torch.randn(B, 4, 64, 64, generator=g).to(dev) for _ in range(3))
        ctx = torch.randn(B, 77, cfg.context_dim, generator=g).to(dev)
        t = torch.randint(0, 1000, (B,), generator=g).to(dev)
        return z, hint, noise, ctx, t

    def stock(B, autocast):
        z, hint, noise, ctx, t = inputs(B)
        for k in train:
            sd_cn[k].grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            loss, eps = R.p_losses(sd_cn, sd_un, cfg, sched, z, t, ctx, hint, noise)
        loss.backward()
        return loss, eps

    out = dict(kind="oracle on PyTorch-ROCm stock kernels vs the HIP engine", rank=args.rank_lora,
               device=torch.cuda.get_device_name(0))
    # ---- parity, three-way
    Bp = args.parity_batch
    z, hint, noise, ctx, t = inputs(Bp)
    loss32, eps32 = stock(Bp, False)
    g32 = {k: sd_cn[k].grad.detach().clone() for k in train}
    loss16, eps16 = stock(Bp, True)
    g16 = {k: sd_cn[k].grad.detach().float().clone() for k in train}
    par = dict(batch=Bp, latent=64, reference="stock fp32 (oracle restatement, PyTorch-ROCm kernels)",
               loss_stock_fp32=float(loss32))
    par["stock_bf16_autocast"] = dict(eps=rel(eps16, eps32), loss=float(loss16), **summarize(g16, g32))
    x_noisy = R.q_sample(sched, z, t, noise)
    for name, dtype in (("engine_fp32", torch.float32), ("engine_bf16", torch.bfloat16)):
        eng = CtrLoRAEngine(sd_un_cpu, [sd_cn_cpu], ncfg, dtype=dtype, device="cuda")
        eps = eng.forward(x_noisy, t, ctx, [hint], record=True)
        eng.zero_grad()
        eng.backward(2.0 * (eps - noise) / eps.numel())
        torch.cuda.synchronize()
        ge = {t_.name: t_.grad.detach().clone() for t_ in eng.controls[0].tr.items}
        par[name] = dict(eps=rel(eps, eps32), loss=float(((eps - noise) ** 2).mean()), **summarize(ge, g32))
        if name == "engine_bf16":
            par["engine_bf16_vs_stock_bf16"] = dict(eps=rel(eps, eps16), **summarize(ge, g16))
        del eng
        torch.cuda.empty_cache()
    out["parity"] = par
    # ---- speed of the stock-kernel path at the bench's batch
    B = args.batch
    for name, autocast in (("fp32", False), ("bf16_autocast", True)):
        stock(B, autocast)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, _ = stock(B, autocast)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out["stock_" + name] = dict(batch=B, ms_per_step=round(dt * 1e3, 1), images_per_s=round(B / dt, 2), loss=float(loss),
                                    note="forward + backward only (no optimizer step); frozen-weight gradients are not "
                                         "formed (only the LoRA / zero-conv / norm tensors require grad); no activation "
                                         "checkpointing (the reference recomputes; this comparator is FASTER than the reference)")
    s = json.dumps(out)
    print(s)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
