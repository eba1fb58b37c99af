"""Sample from a trained checkpoint over a dataset of condition images -- command line of the reference's
scripts/sample.py:22-113 (same flags and output layout: <save_dir>/{sample,control,img}/<idx>.png + prompt.txt),
with the denoising loop on the MI355X engine.

    python scripts/sample.py --dataroot ./data/my_condition --config ./configs/ctrlora_finetune_sd15_rank128.yaml \\
        --ckpt ./runs/x/.../N-Step-Checkpoint_epoch=0_global_step=999.ckpt --n_samples 10 --save_dir ./samples

Images are resized with Pillow (LANCZOS when enlarging, BOX when shrinking) to the reference's target size -- short
side `512`, both sides rounded to multiples of 64 -- where the reference uses OpenCV (LANCZOS4 / AREA): same
geometry, slightly different resampling kernels.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

TASKS = ["hed", "canny", "seg", "depth", "normal", "openpose", "hedsketch", "bbox", "outpainting", "inpainting", "blur",
         "grayscale"]


def get_parser():
    p = argparse.ArgumentParser(description="args")
    p.add_argument("--dataroot", type=str, required=True, help="path to dataset")
    p.add_argument("--multigen20m", action="store_true", default=False, help="use multigen20m dataset")
    p.add_argument("--task", type=str, choices=TASKS, help="task name")
    p.add_argument("--config", type=str, required=True, help="path to model config file")
    p.add_argument("--ckpt", type=str, required=True, help="path to trained checkpoint")
    p.add_argument("--n_samples", type=int, default=10, help="number of samples")
    p.add_argument("--save_dir", type=str, required=True, help="path to save samples")
    p.add_argument("--ddim_steps", type=int, default=50, help="number of DDIM steps")
    p.add_argument("--ddim_eta", type=float, default=0.0, help="DDIM eta")
    p.add_argument("--strength", type=float, default=1.0, help="strength of controlnet")
    p.add_argument("--cfg", type=float, default=7.5, help="unconditional guidance scale")
    p.add_argument("--empty_prompt", action="store_true", default=False, help="experimental: use empty prompt")
    return p


def target_size(h: int, w: int, resolution: int = 512):
    """Short side -> `resolution`, then both sides to the nearest multiple of 64 (annotator/util.py:28-38)."""
    k = float(resolution) / min(h, w)
    return int(np.round(h * k / 64.0)) * 64, int(np.round(w * k / 64.0)) * 64, k


def resize_image(img: np.ndarray, resolution: int = 512) -> np.ndarray:
    from PIL import Image
    H, W, k = target_size(img.shape[0], img.shape[1], resolution)
    return np.asarray(Image.fromarray(img).resize((W, H), Image.LANCZOS if k > 1 else Image.BOX))


def sample_dataset(model, sampler, dataset, args, device="cuda"):
    """The per-item loop of the reference script; `model` needs get_learned_conditioning / decode_first_stage /
    control_scales, `sampler` a DDIMSampler-like `.sample(...)`.  Returns the number of items written."""
    import torch
    from PIL import Image
    from api import hwc3
    for sub in ("sample", "control", "img"):
        os.makedirs(os.path.join(args.save_dir, sub), exist_ok=True)
    n = 0
    with torch.no_grad():
        for idx, item in enumerate(dataset):
            img = resize_image(hwc3(((item["jpg"] + 1.0) / 2.0 * 255.0).astype(np.uint8)), 512)
            prompt = "" if args.empty_prompt else item["txt"]
            control_u8 = resize_image(hwc3((item["hint"] * 255.0).astype(np.uint8)), 512)
            control = (torch.from_numpy(control_u8.copy()).float().to(device) / 255.0).permute(2, 0, 1)[None]
            H, W, _ = img.shape
            cond = {"c_concat": [control], "c_crossattn": [model.get_learned_conditioning([prompt])], "task": args.task}
            un_cond = {"c_concat": [control], "c_crossattn": [model.get_learned_conditioning([""])], "task": args.task}
            model.control_scales = [args.strength] * 13
            samples, _ = sampler.sample(args.ddim_steps, 1, (4, H // 8, W // 8), cond, verbose=False, eta=args.ddim_eta,
                                        unconditional_guidance_scale=args.cfg, unconditional_conditioning=un_cond)
            x = model.decode_first_stage(samples)[0]
            x = (x.permute(1, 2, 0) * 127.5 + 127.5).cpu().numpy().clip(0, 255).astype(np.uint8)
            Image.fromarray(x).save(os.path.join(args.save_dir, "sample", f"{idx}.png"))
            Image.fromarray(img).save(os.path.join(args.save_dir, "img", f"{idx}.png"))
            Image.fromarray(control_u8).save(os.path.join(args.save_dir, "control", f"{idx}.png"))
            with open(os.path.join(args.save_dir, "prompt.txt"), "a") as f:
                print(prompt.strip(), file=f)
            n += 1
    return n


def main(argv=None):
    from torch.utils.data import Subset
    from cldm.cldm_ctrlora_pretrain import ControlPretrainLDM
    from cldm.ddim_hacked import DDIMSampler
    from cldm.model import create_model, load_state_dict
    args = get_parser().parse_args(argv)
    if args.multigen20m:
        from datasets.multigen20m import MultiGen20M
        dataset = MultiGen20M(path_json=os.path.join(args.dataroot, "json_files", f"aesthetics_plus_all_group_{args.task}_all.json"),
                              path_meta=args.dataroot, task=args.task, drop_rate=0.0, random_cropping=False)
    else:
        from datasets.custom_dataset import CustomDataset
        dataset = CustomDataset(args.dataroot)
    if args.n_samples < len(dataset):
        dataset = Subset(dataset, range(args.n_samples))
    print("Dataset size:", len(dataset))
    model = create_model(args.config).cpu()
    pre = isinstance(model, ControlPretrainLDM)
    if pre:                                        # the checkpoint was saved with a task bank aliased into the tree
        model.control_model.switch_lora(args.task)
    model.load_state_dict(load_state_dict(args.ckpt, location="cpu"), strict=True)
    if pre:
        model.control_model.switch_lora(args.task)
    model = model.cuda().eval()
    print(f"Successfully load model ckpt from {args.ckpt}")
    os.makedirs(args.save_dir, exist_ok=True)
    sample_dataset(model, DDIMSampler(model), dataset, args)
    print("Done")


if __name__ == "__main__":
    main()
