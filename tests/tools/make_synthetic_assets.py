"""Synthetic assets for running the training / sampling SCRIPTS end to end on a GPU box that has no datasets and no
checkpoints: narrow copies of the shipped YAML configs (same targets, model_channels 64, LoRA rank 32,
VAE ch 64, CLIP at full size), an 'SD checkpoint' and a 'Base-ControlNet checkpoint' with random weights in the layouts
the scripts read, a CustomDataset directory and a MultiGen-20M style directory with a few 512 x 512 image pairs.

    python tests/tools/make_synthetic_assets.py --out /tmp/ctrlora_synth [--n 8]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def narrow(src, dst, tasks=None):
    with open(os.path.join(ROOT, "configs", src)) as f:
        tree = yaml.safe_load(f)
    p = tree["model"]["params"]
    for k in ("control_stage_config", "unet_config"):
        p[k]["params"].update(model_channels=64)
    if "lora_rank" in p["control_stage_config"]["params"]:
        p["control_stage_config"]["params"]["lora_rank"] = 32
    if tasks is not None:
        p["control_stage_config"]["params"]["tasks"] = list(tasks)
    p["first_stage_config"]["params"]["ddconfig"]["ch"] = 64
    tree.pop("x-shared", None)
    with open(dst, "w") as f:
        yaml.safe_dump(tree, f)
    return dst


def image_pair(i, size=512):
    from tests.golden.make_golden_vae import test_image
    t = test_image(1, size, size)[0] * (0.8 + 0.05 * (i % 4))
    tgt = ((t.permute(1, 2, 0).numpy() * 0.5 + 0.5) * 255).clip(0, 255).astype(np.uint8)
    g = tgt.mean(-1)
    edge = (np.abs(np.diff(g, axis=0, prepend=g[:1])) + np.abs(np.diff(g, axis=1, prepend=g[:, :1])) > 6).astype(np.uint8) * 255
    return tgt, np.stack([edge] * 3, -1)


def main():
    from PIL import Image
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--n", type=int, default=8)
    args = ap.parse_args()
    out = args.out
    os.makedirs(out, exist_ok=True)
    tasks = ["hed", "canny"]
    cfg_ft = narrow("ctrlora_finetune_sd15_rank128.yaml", os.path.join(out, "finetune_narrow.yaml"))
    cfg_pt = narrow("ctrlora_pretrain_sd15_9tasks_rank128.yaml", os.path.join(out, "pretrain_narrow.yaml"), tasks)
    cfg_full = narrow("ctrlora_finetune_sd15_full.yaml", os.path.join(out, "finetune_full_narrow.yaml"))
    # ---- checkpoints
    from cldm.model import create_model
    torch.manual_seed(0)
    m = create_model(cfg_pt)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for _, prm in m.named_parameters():       # nothing trivially zero
            if prm.numel() and float(prm.abs().max()) == 0.0:
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.02)
    sd = m.state_dict()
    torch.save({"state_dict": {k: v for k, v in sd.items() if not k.startswith("control_model.")}}, os.path.join(out, "sd_synth.ckpt"))
    m.control_model.switch_lora("hed")
    torch.save({k: v for k, v in m.state_dict().items() if k.startswith("control_model.") and "loras_dict" not in k},
               os.path.join(out, "basecn_synth.ckpt"))
    del m
    # ---- datasets
    cust = os.path.join(out, "custom")
    for d in ("source", "target"):
        os.makedirs(os.path.join(cust, d), exist_ok=True)
    mg = os.path.join(out, "multigen")
    for d in ("json_files", "conditions", "images"):
        os.makedirs(os.path.join(mg, d), exist_ok=True)
    lines = []
    mg_lines = {t: [] for t in tasks}
    for i in range(args.n):
        tgt, cond = image_pair(i)
        Image.fromarray(tgt).save(os.path.join(cust, "target", f"{i:04d}.jpg"), quality=95)
        Image.fromarray(cond).save(os.path.join(cust, "source", f"{i:04d}.jpg"), quality=95)
        lines.append(json.dumps(dict(source=f"source/{i:04d}.jpg", target=f"target/{i:04d}.jpg", prompt=f"synthetic pattern number {i}")))
        Image.fromarray(tgt).save(os.path.join(mg, "images", f"img_{i:04d}.jpg"), quality=95)
        for t in tasks:
            Image.fromarray(cond if t == "canny" else 255 - cond).save(os.path.join(mg, "conditions", f"{t}_{i:04d}.jpg"), quality=95)
            mg_lines[t].append(json.dumps({"source": f"./img_{i:04d}.jpg", "prompt": f"synthetic pattern number {i}", f"control_{t}": f"{t}_{i:04d}.jpg"}))
    with open(os.path.join(cust, "prompt.json"), "w") as f:
        f.write("\n".join(lines) + "\n")
    for t in tasks:
        with open(os.path.join(mg, "json_files", f"aesthetics_plus_all_group_{t}_all.json"), "w") as f:
            f.write("\n".join(mg_lines[t]) + "\n")
    print(json.dumps(dict(finetune_config=cfg_ft, pretrain_config=cfg_pt, full_config=cfg_full, sd_ckpt=os.path.join(out, "sd_synth.ckpt"),
                          cn_ckpt=os.path.join(out, "basecn_synth.ckpt"), custom=cust, multigen=mg)))


if __name__ == "__main__":
    main()
