"""'Same silicon, stock kernels' comparator: the oracle -- the functional restatement of the
reference modules in plain torch ops (oracle/ref_model.py) -- run on the GPU through PyTorch-ROCm's own kernels
(rocBLAS / hipBLASLt / MIOpen / ATen), fp32 and bf16-autocast, forward + backward of one LoRA fine-tuning step
at the bench's shape (rank 128, B per GPU 8, latent 64x64), timed next to the engine.

Test / measurement infrastructure only (imports oracle/): python tests/tools/compare_stock.py [--batch 8] [--steps 5]
Not run on hardware yet (GPU budget of round 1 was spent); intended for the next round's profiles/.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--rank-lora", type=int, default=128)
    args = ap.parse_args()
    from oracle import arch, ref_model as R
    dev = torch.device("cuda")
    cfg = arch.ArchCfg(lora_rank=args.rank_lora)
    sd_cn = {k: v.to(dev) for k, v in arch.make_state(arch.controlnet_shapes(cfg), 0).items()}
    sd_un = {k: v.to(dev) for k, v in arch.make_state(arch.unet_shapes(cfg), 0).items()}
    train = [k for k in sd_cn if arch.is_trainable(k)]
    for k in train:
        sd_cn[k].requires_grad_(True)
    B = args.batch
    g = torch.Generator().manual_seed(0)
    z, hint, noise = (torch.randn(B, 4, 64, 64, generator=g).to(dev) for _ in range(3))
    ctx = torch.randn(B, 77, cfg.context_dim, generator=g).to(dev)
    t = torch.randint(0, 1000, (B,), generator=g).to(dev)
    sched = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in R.make_schedule().items()}
    out = {}
    for name, autocast in (("fp32", None), ("bf16_autocast", torch.bfloat16)):
        def step():
            for k in train:
                sd_cn[k].grad = None
            with torch.autocast("cuda", dtype=autocast, enabled=autocast is not None):
                loss, _ = R.p_losses(sd_cn, sd_un, cfg, sched, z, t, ctx, hint, noise)
            loss.backward()
            return loss
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out[name] = dict(ms_per_step=round(dt * 1e3, 1), images_per_s=round(B / dt, 2), loss=float(loss),
                         note="forward + backward only (no optimizer); frozen-weight gradients are not formed "
                              "(only the LoRA / zero-conv / norm tensors require grad)")
    print(json.dumps(dict(kind="oracle on PyTorch-ROCm stock kernels", batch=B, rank=args.rank_lora, **out)))


if __name__ == "__main__":
    main()
