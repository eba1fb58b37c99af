"""Combine SD1.5 + Base ControlNet + one LoRA file into a single checkpoint
(CLI and behaviour of the reference's scripts/tool_combine_weights.py:1-50).

Later sources override earlier ones key by key (SD -> base ControlNet -> LoRA); `model_ema.*` entries of the SD
checkpoint are dropped and a zero `logvar` (1000,) is added, which is what `ControlFinetuneLDM` expects for a
strict load.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def combine(sd_ckpt: dict, base_ckpt: dict, lora_ckpt: dict) -> dict:
    out = {k: v for k, v in sd_ckpt.items() if not k.startswith("model_ema.")}
    out.update(base_ckpt)
    out.update(lora_ckpt)
    out["logvar"] = torch.zeros(1000)
    return out


def get_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--sd_ckpt", type=str, default="./ckpts/sd15/v1-5-pruned.ckpt", help="path to SD1.5 checkpoint")
    p.add_argument("--base_ckpt", type=str, default="./ckpts/ctrlora-basecn/ctrlora_sd15_basecn700k.ckpt",
                   help="path to Base ControlNet checkpoint")
    p.add_argument("--lora_ckpt", type=str, required=True, help="path to LoRA checkpoint")
    p.add_argument("--save_path", type=str, required=True, help="path to save combined weights")
    return p


def main(argv=None):
    from cldm.model import load_state_dict
    args = get_parser().parse_args(argv)
    ckpt = combine(load_state_dict(args.sd_ckpt, location="cpu"), load_state_dict(args.base_ckpt, location="cpu"),
                   load_state_dict(args.lora_ckpt, location="cpu"))
    os.makedirs(os.path.dirname(args.save_path) or ".", exist_ok=True)
    torch.save(ckpt, args.save_path)
    print(f"Saved combined weights to [{args.save_path}]")


if __name__ == "__main__":
    main()
