"""LoRA fine-tuning on a new condition -- command line of the reference's scripts/train_ctrlora_finetune.py:22-129
(same flags), on the MI355X engine with a Lightning-free loop.

    torchrun --nproc-per-node 8 scripts/train_ctrlora_finetune.py --dataroot ./data/my_condition \\
        --config ./configs/ctrlora_finetune_sd15_rank128.yaml --sd_ckpt ./ckpts/sd15/v1-5-pruned.ckpt \\
        --cn_ckpt ./ckpts/ctrlora-basecn/ctrlora_sd15_basecn700k.ckpt --bs 8 --precision 16 --max_steps 1000

One process per GPU (the reference lets Lightning spawn them: strategy='ddp', devices=-1); `--precision 32` runs the
fp32 parity mode of the engine, 16 / bf16 the bf16-storage mode.
"""
import argparse
import datetime
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TASKS = ["hed", "canny", "seg", "depth", "normal", "openpose", "hedsketch", "bbox", "outpainting", "inpainting", "blur",
         "grayscale"]


def get_parser():
    p = argparse.ArgumentParser(description="args")
    # dataset
    p.add_argument("--dataroot", type=str, required=True, help="path to dataset")
    p.add_argument("--drop_rate", type=float, default=0.3, help="drop rate for classifier-free guidance")
    p.add_argument("--multigen20m", action="store_true", default=False, help="use multigen20m dataset")
    p.add_argument("--task", type=str, choices=TASKS, help="task name")
    p.add_argument("--subset", type=int, default=0, help="train on a subset of the dataset")
    # model
    p.add_argument("--config", type=str, required=True, help="path to model config file")
    p.add_argument("--sd_ckpt", type=str, required=True, help="path to pretrained stable diffusion checkpoint")
    p.add_argument("--cn_ckpt", type=str, required=True, help="path to pretrained controlnet checkpoint")
    # training
    p.add_argument("-n", "--name", type=str, help="experiment name")
    p.add_argument("--lr", type=float, default=1e-5, help="learning rate")
    p.add_argument("--bs", type=int, default=1, help="batchsize per device")
    p.add_argument("--max_steps", type=int, default=100000, help="max training steps")
    p.add_argument("--gradacc", type=int, default=1, help="gradient accumulation")
    p.add_argument("--precision", type=int, default=32, help="precision")
    p.add_argument("--save_memory", action="store_true", default=False, help="accepted for compatibility (no effect: "
                   "the engine's attention never materialises the score matrix)")
    p.add_argument("--img_logger_freq", type=int, default=1000, help="img logger freq")
    p.add_argument("--ckpt_logger_freq", type=int, default=1000, help="ckpt logger freq")
    p.add_argument("--num_workers", type=int, default=None,
                   help="DataLoader worker processes (default: the reference's 16, capped at the host's cores); every epoch "
                        "forks them again, which is slow from a process with a large address space")
    return p


def init_weights(model, sd_weights: dict, control_weights: dict, report_dir: str = "./tmp"):
    """Initialise a fine-tune model from an SD checkpoint and a Base-ControlNet checkpoint
    (train_ctrlora_finetune.py:76-113): every SD tensor the model has is taken; of the ControlNet checkpoint every
    `control_model` tensor the model has EXCEPT keys containing 'lora' (fresh LoRA layers are trained: A ~ N(0, 1/r),
    B = 0).  Returns ((copied_sd, missing_sd), (copied_cn, missing_cn)); the four lists are also written to
    `report_dir` under the reference's file names."""
    scratch = model.state_dict()
    copied_sd = [k for k in sd_weights if k in scratch]
    missing_sd = [k for k in sd_weights if k not in scratch]
    for k in copied_sd:
        scratch[k] = sd_weights[k].clone()
    cn_keys = [k for k in control_weights if "control_model" in k]
    missing_cn = [k for k in cn_keys if k not in scratch]
    copied_cn = [k for k in cn_keys if k in scratch and "lora" not in k]
    for k in copied_cn:
        scratch[k] = control_weights[k].clone()
    model.load_state_dict(scratch, strict=True)
    if report_dir:
        os.makedirs(report_dir, exist_ok=True)
        for name, keys in (("finetune_missing_keys_sd", missing_sd), ("finetune_copied_keys_sd", copied_sd),
                           ("finetune_missing_keys_cn", missing_cn), ("finetune_copied_keys_cn", copied_cn)):
            with open(os.path.join(report_dir, name + ".txt"), "w") as f:
                f.write("\n".join(keys))
    return (copied_sd, missing_sd), (copied_cn, missing_cn)


def build_dataloader(args, world_size: int, rank: int):
    from torch.utils.data import DataLoader, DistributedSampler, Subset
    if args.multigen20m:
        from datasets.multigen20m import MultiGen20M
        dataset = MultiGen20M(path_json=os.path.join(args.dataroot, "json_files", f"aesthetics_plus_all_group_{args.task}_all.json"),
                              path_meta=args.dataroot, task=args.task, drop_rate=args.drop_rate)
    else:
        from datasets.custom_dataset import CustomDataset
        dataset = CustomDataset(args.dataroot, drop_rate=args.drop_rate)
    if args.subset > 0:
        dataset = Subset(dataset, range(args.subset))
    sampler = DistributedSampler(dataset, num_replicas=world_size, rank=rank, shuffle=True) if world_size > 1 else None
    workers = min(16, os.cpu_count() or 1) if getattr(args, "num_workers", None) is None else max(0, args.num_workers)
    loader = DataLoader(dataset, num_workers=workers, batch_size=args.bs, shuffle=sampler is None,
                        sampler=sampler, drop_last=True)
    return dataset, loader


def main(argv=None):
    import gc
    from cldm.logger import CheckpointEveryNSteps, ImageLogger
    from cldm.model import create_model, load_state_dict
    from ctrlora_amd.trainer import Trainer
    args = get_parser().parse_args(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dataset, loader = build_dataloader(args, world, rank)
    if rank == 0:
        print("Dataset size:", len(dataset))
        print("Number of devices:", world)
        print("Batch size per device:", args.bs)
        print("Gradient accumulation:", args.gradacc)
        print("Total batch size:", args.bs * world * args.gradacc)
    model = create_model(args.config).cpu()
    model.learning_rate = args.lr
    model.sd_locked = True
    model.only_mid_control = False
    init_weights(model, load_state_dict(args.sd_ckpt, location="cpu"), load_state_dict(args.cn_ckpt, location="cpu"))
    print(f"Successfully initialize SD from {args.sd_ckpt}")
    print(f"Successfully initialize ControlNet from {args.cn_ckpt}")
    gc.collect()
    name = args.name or datetime.datetime.now().strftime("%Y-%m-%d-%H-%M-%S")
    trainer = Trainer(max_steps=args.max_steps, accumulate_grad_batches=args.gradacc, precision=args.precision,
                      callbacks=[ImageLogger(batch_frequency=args.img_logger_freq),
                                 CheckpointEveryNSteps(save_step_frequency=args.ckpt_logger_freq)],
                      default_root_dir=os.path.join("runs", name))
    trainer.fit(model, loader)


if __name__ == "__main__":
    main()
