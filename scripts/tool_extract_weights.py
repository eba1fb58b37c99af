"""Extract the Base-ControlNet ("control") or the per-condition LoRA file from a training checkpoint
(CLI and behaviour of the reference's scripts/tool_extract_weights.py:1-79).

    python scripts/tool_extract_weights.py -t lora    --ckpt last.ckpt --save_path lineart_rank128.ckpt
    python scripts/tool_extract_weights.py -t control --ckpt last.ckpt --save_path basecn.ckpt
    python scripts/tool_extract_weights.py -t lora --from_base --from_base_config configs/ctrlora_pretrain_sd15_9tasks_rank128.yaml \\
           --ckpt pretrain_last.ckpt --save_path loras_dir/          # one <task>.ckpt per task bank

A LoRA file is the ecosystem contract (the ComfyUI node and api.CtrLoRA.create_model consume it): every
`control_model.*` entry outside the task banks whose key names a LoRA layer, a zero conv
(`zero_convs` / `middle_block_out`) or a norm layer -- 246 tensors for the SD1.5 ControlNet, whatever the rank.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

LORA_FILE_MARKERS = ("lora_layer", "zero_convs", "middle_block_out", "norm")


def _in_control_tree(key: str) -> bool:
    # task banks (`loras_dict.<task>.<i>`) are reached through the tree after switch_lora, never saved as such
    return "control_model" in key and "loras_dict" not in key


def extract_lora(ckpt: dict) -> dict:
    return {k: v for k, v in ckpt.items() if _in_control_tree(k) and any(m in k for m in LORA_FILE_MARKERS)}


def extract_control(ckpt: dict) -> dict:
    return {k: v for k, v in ckpt.items() if _in_control_tree(k)}


def extract_task_loras(model, ckpt: dict) -> dict:
    """--from_base: load a pre-training checkpoint into a ControlPretrainLDM and return {task: LoRA file dict}.
    `switch_lora(task)` aliases the task's bank into the module tree, so the tree-addressed keys of a plain
    LoRA file come out of `state_dict()`."""
    from cldm.cldm_ctrlora_pretrain import ControlPretrainLDM
    assert isinstance(model, ControlPretrainLDM)
    model.control_model.switch_lora(model.control_model.tasks[0])   # the checkpoint was saved with a bank aliased in
    model.load_state_dict(ckpt, strict=True)
    files = {}
    for task in model.control_model.tasks:
        model.control_model.switch_lora(task)
        files[task] = {k: v.detach().clone() for k, v in extract_lora(model.state_dict()).items()}
    return files


def get_parser():
    p = argparse.ArgumentParser()
    p.add_argument("-t", "--type", type=str, required=True, choices=["control", "lora"], help="type of weights to extract")
    p.add_argument("--ckpt", type=str, required=True, help="path to trained checkpoint")
    p.add_argument("--save_path", type=str, required=True, help="path to save extracted weights")
    p.add_argument("--from_base", action="store_true", help="extract weights from the Base ControlNet")
    p.add_argument("--from_base_config", type=str, help="path to Base ControlNet config file")
    return p


def main(argv=None):
    from cldm.model import create_model, load_state_dict
    args = get_parser().parse_args(argv)
    ckpt = load_state_dict(args.ckpt, location="cpu")
    if args.type == "control":
        torch.save(extract_control(ckpt), args.save_path)
        print(f"Extracted weights saved to {args.save_path}")
    elif not args.from_base:
        torch.save(extract_lora(ckpt), args.save_path)
        print(f"Extracted weights saved to {args.save_path}")
    else:
        assert not os.path.isfile(args.save_path)
        os.makedirs(args.save_path, exist_ok=True)
        model = create_model(args.from_base_config).cpu()
        for task, sd in extract_task_loras(model, ckpt).items():
            path = os.path.join(args.save_path, f"{task}.ckpt")
            torch.save(sd, path)
            print(f"Extracted weights for task {task} saved to {path}")
    print("Done.")


if __name__ == "__main__":
    main()
