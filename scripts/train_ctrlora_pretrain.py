"""Multi-task Base-ControlNet pre-training -- command line of the reference's scripts/train_ctrlora_pretrain.py:22-140
(same flags), on the MI355X engine with the Lightning-free loop (BASELINE.json configs[3]).

    torchrun --nproc-per-node 8 scripts/train_ctrlora_pretrain.py --dataroot ./data/MultiGen-20M \\
        --config ./configs/ctrlora_pretrain_sd15_9tasks_rank128.yaml --sd_ckpt ./ckpts/sd15/v1-5-pruned.ckpt \\
        --cn_ckpt ./ckpts/control_sd15_init.pth --bs 4 --max_steps 700000

Per task one MultiGen20M dataset (json_files/aesthetics_plus_all_group_<task>_all.json), concatenated; the
BatchSchedulerSampler makes every mini-batch single-task (the task name travels in the batch and selects the LoRA
bank).  One process per GPU; the gradient exchange moves the shared (base) buffer plus the LoRA banks that are live on
some rank this step (ctrlora_amd.parallel.BankedGradAllReduce).  Initialisation: every SD tensor and every
`control_model` tensor of the ControlNet checkpoint that the model has (:79-112).
"""
import argparse
import datetime
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def get_parser():
    p = argparse.ArgumentParser(description="args")
    p.add_argument("--dataroot", type=str, required=True, help="path to dataset")
    p.add_argument("--config", type=str, required=True, help="path to model config file")
    p.add_argument("--sd_ckpt", type=str, required=True, help="path to pretrained stable diffusion checkpoint")
    p.add_argument("--cn_ckpt", type=str, required=True, help="path to pretrained controlnet checkpoint")
    p.add_argument("-n", "--name", type=str, help="experiment name")
    p.add_argument("--lr", type=float, default=1e-5, help="learning rate")
    p.add_argument("--bs", type=int, default=4, help="batchsize per device")
    p.add_argument("--max_steps", type=int, default=700000, help="max training steps")
    p.add_argument("--gradacc", type=int, default=1, help="gradient accumulation")
    p.add_argument("--precision", type=int, default=32, help="precision")
    p.add_argument("--save_memory", action="store_true", default=False, help="accepted for compatibility (no effect)")
    p.add_argument("--img_logger_freq", type=int, default=10000, help="img logger freq")
    p.add_argument("--ckpt_logger_freq", type=int, default=10000, help="ckpt logger freq")
    p.add_argument("--num_workers", type=int, default=16)
    return p


def init_weights(model, sd_weights: dict, control_weights: dict, report_dir: str = "./tmp"):
    """:76-112 -- every SD tensor the model has, and every `control_model` tensor of the ControlNet checkpoint."""
    scratch = model.state_dict()
    lists = {}
    for tag, src, keep in (("sd", sd_weights, lambda k: True), ("cn", control_weights, lambda k: "control_model" in k)):
        copied = [k for k in src if keep(k) and k in scratch]
        lists[tag] = (copied, [k for k in src if keep(k) and k not in scratch])
        for k in copied:
            scratch[k] = src[k].clone()
    model.load_state_dict(scratch, strict=True)
    if report_dir:
        os.makedirs(report_dir, exist_ok=True)
        for tag, (copied, missing) in lists.items():
            for kind, keys in (("copied", copied), ("missing", missing)):
                with open(os.path.join(report_dir, f"pretrain_{kind}_keys_{tag}.txt"), "w") as f:
                    f.write("\n".join(keys))
    return lists


def main(argv=None):
    import gc
    import yaml
    from torch.utils.data import ConcatDataset, DataLoader
    from cldm.logger import CheckpointEveryNSteps, ImageLogger
    from cldm.model import create_model, load_state_dict
    from ctrlora_amd.trainer import Trainer
    from datasets.dataset_collate import collate_fn
    from datasets.multi_task_scheduler import BatchSchedulerSampler
    from datasets.multigen20m import MultiGen20M
    args = get_parser().parse_args(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    with open(args.config) as f:
        tasks = yaml.safe_load(f)["model"]["params"]["control_stage_config"]["params"]["tasks"]
    dataset = ConcatDataset([
        MultiGen20M(path_json=os.path.join(args.dataroot, "json_files", f"aesthetics_plus_all_group_{task}_all.json"),
                    path_meta=args.dataroot, task=task, drop_rate=0.3) for task in tasks])
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():      # the sampler asks the process group for rank / world size
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            import torch
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl")
    loader = DataLoader(dataset=dataset, num_workers=args.num_workers, batch_size=args.bs,
                        persistent_workers=args.num_workers > 0, collate_fn=collate_fn,
                        sampler=BatchSchedulerSampler(dataset=dataset, batch_size=args.bs, distributed=world > 1, shuffle=True))
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
