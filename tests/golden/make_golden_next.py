"""Golden vectors for the SURVEY.md 8(f) "next" rows, from the UNMODIFIED reference (build container only).

    python tests/golden/make_golden_next.py       # writes tests/golden/next_rows.pt

  f2  checkpoint I/O: scripts/tool_extract_weights.py {extract_lora, extract_control} on a fine-tune model and,
      per task, on a pre-train model after switch_lora (the --from_base flow); the api.CtrLoRA.create_model load
      sequence on a 2-LoRA inference model (base ControlNet filtered by check_key, then per LoRA file:
      switch_lora(i) -> load_state_dict(strict=False) -> copy_weights_to_switchable()).
  f3  datasets/multi_task_scheduler.BatchSchedulerSampler index streams (plain and 2-rank distributed).

Every parameter value is a function of its state-dict key (key_tensor below), so the fixtures hold only
per-key checksums and the tests rebuild identical inputs without the reference.
"""
import hashlib
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def key_tensor(key: str, shape, salt: str = "") -> torch.Tensor:
    """Deterministic values for a state-dict entry: N(0,1) drawn from a generator seeded by the key."""
    seed = int.from_bytes(hashlib.sha256((salt + "|" + key).encode()).digest()[:7], "little")
    g = torch.Generator().manual_seed(seed)
    return torch.randn(tuple(shape), generator=g)


def checksum(t: torch.Tensor):
    f = t.detach().double().flatten()
    w = torch.arange(1, f.numel() + 1, dtype=torch.float64)
    return [float(f.sum()), float((f * w).sum() / max(1, f.numel()))]


def fill(module, salt):
    sd = module.state_dict()
    module.load_state_dict({k: key_tensor(k, v.shape, salt) for k, v in sd.items()}, strict=True)


def tiny_control_params(extra):
    p = dict(image_size=32, in_channels=4, model_channels=64, hint_channels=3, attention_resolutions=[4, 2, 1],
             num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
             transformer_depth=1, context_dim=96, use_checkpoint=True, legacy=False)
    p.update(extra)
    return p


def load_ref_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_ckpt_golden(out):
    from ldm.util import instantiate_from_config
    tool = load_ref_module("ref_tool_extract_weights", f"{REF}/scripts/tool_extract_weights.py")

    # ---- fine-tune model: which keys go into a LoRA file / a control file
    ft = instantiate_from_config(dict(target="cldm.cldm_ctrlora_finetune.ControlNetFinetune",
                                      params=tiny_control_params(dict(ft_with_lora=True, lora_rank=32,
                                                                      norm_trainable=True))))
    full = {"control_model." + k: v for k, v in ft.state_dict().items()}
    full.update({"model.diffusion_model.fake.weight": torch.zeros(1), "first_stage_model.norm.weight": torch.zeros(1),
                 "cond_stage_model.transformer.final_layer_norm.weight": torch.zeros(1), "logvar": torch.zeros(1)})
    out["finetune_extract_lora_keys"] = sorted(tool.extract_lora(full).keys())
    out["finetune_extract_control_keys"] = sorted(tool.extract_control(full).keys())

    # ---- pre-train model, two tasks: per-task LoRA files after switch_lora (bank -> tree aliasing order)
    pre = instantiate_from_config(dict(target="cldm.cldm_ctrlora_pretrain.ControlNetPretrain",
                                       params=tiny_control_params(dict(lora_rank=32, tasks=["hed", "canny"]))))
    fill(pre, "pretrain")
    per_task = {}
    for task in pre.tasks:
        pre.switch_lora(task)
        sd = {"control_model." + k: v for k, v in pre.state_dict().items()}
        ex = tool.extract_lora(sd)
        per_task[task] = {k: checksum(v) for k, v in ex.items()}
    out["pretrain_task_files"] = per_task
    out["pretrain_extract_control_keys"] = sorted(
        tool.extract_control({"control_model." + k: v for k, v in pre.state_dict().items()}).keys())

    # ---- inference model, 2 LoRAs: the api.CtrLoRA.create_model load sequence
    def check_key(k):   # api.py:28 (the file itself cannot be imported here: PIL / annotator dependencies)
        return 'lora_layer' in k or 'zero_convs' in k or 'middle_block_out' in k or 'norm' in k

    src = open(f"{REF}/api.py").read()
    assert "return 'lora_layer' in k or 'zero_convs' in k or 'middle_block_out' in k or 'norm' in k" in src
    inf = instantiate_from_config(dict(target="cldm.cldm_ctrlora_inference.ControlNetInference",
                                       params=tiny_control_params(dict(lora_rank=32, lora_num=2))))
    fill(inf, "inference-init")

    class Holder(torch.nn.Module):   # gives the "control_model." prefix the checkpoints carry
        def __init__(self, cm):
            super().__init__()
            self.control_model = cm

    holder = Holder(inf)
    # base ControlNet checkpoint = a fine-tune model's full state (values keyed by name), filtered as the API does
    base_sd = {"control_model." + k: key_tensor(k, v.shape, "basecn") for k, v in ft.state_dict().items()}
    base_sd = {k: v for k, v in base_sd.items() if k.startswith("control_model") and not check_key(k)}
    missing, unexpected = holder.load_state_dict(base_sd, strict=False)
    out["api_base_unexpected"] = sorted(unexpected)
    for i in range(2):
        lora_sd = {"control_model." + k: key_tensor(k, v.shape, f"lora{i}") for k, v in ft.state_dict().items()}
        lora_sd = {k: v for k, v in lora_sd.items() if check_key(k)}
        inf.switch_lora(i)
        holder.load_state_dict(lora_sd, strict=False)
        inf.copy_weights_to_switchable()
    out["api_final_state"] = {k: checksum(v) for k, v in inf.state_dict().items()}
    out["api_num_lora_keys"] = len(lora_sd)
    print(f"[golden-next] ckpt: lora file {len(out['finetune_extract_lora_keys'])} keys, control file "
          f"{len(out['finetune_extract_control_keys'])} keys, inference state {len(out['api_final_state'])} keys")


class _DS(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


def gen_sampler_golden(out):
    import torch.distributed as dist
    mts = load_ref_module("ref_multi_task_scheduler", f"{REF}/datasets/multi_task_scheduler.py")
    sizes = [5, 3, 7]
    ds = torch.utils.data.ConcatDataset([_DS(n) for n in sizes])
    res = {}
    for shuffle in (False, True):
        torch.manual_seed(1234); np.random.seed(4321)
        s = mts.BatchSchedulerSampler(ds, batch_size=2, distributed=False, shuffle=shuffle)
        res[f"plain_shuffle{int(shuffle)}"] = dict(len=len(s), idx=list(iter(s)))
    # distributed: DistributedSampler reads world size / rank from torch.distributed
    saved = (dist.is_available, dist.is_initialized, dist.get_world_size, dist.get_rank)
    try:
        for rank in (0, 1):
            dist.is_available = lambda: True
            dist.is_initialized = lambda: True
            dist.get_world_size = lambda group=None: 2
            dist.get_rank = lambda group=None, r=rank: r
            for shuffle in (False, True):
                torch.manual_seed(1234); np.random.seed(4321 + rank)
                s = mts.BatchSchedulerSampler(ds, batch_size=2, distributed=True, shuffle=shuffle)
                res[f"dist_rank{rank}_shuffle{int(shuffle)}"] = dict(len=len(s), idx=list(iter(s)))
    finally:
        dist.is_available, dist.is_initialized, dist.get_world_size, dist.get_rank = saved
    out["sampler"] = dict(sizes=sizes, batch_size=2, streams=res)
    print("[golden-next] sampler:", {k: len(v["idx"]) for k, v in res.items()})


if __name__ == "__main__":
    assert os.path.isdir(REF), "run in the build container (needs /root/reference)"
    from make_golden import install_stubs
    install_stubs()
    sys.path.insert(0, REF)
    os.chdir("/tmp")
    out = {}
    gen_ckpt_golden(out)
    gen_sampler_golden(out)
    torch.save(out, f"{HERE}/next_rows.pt")
    print("[golden-next] wrote", f"{HERE}/next_rows.pt", os.path.getsize(f"{HERE}/next_rows.pt"), "bytes")
