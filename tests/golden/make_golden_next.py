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
if HERE not in sys.path:
    sys.path.insert(0, HERE)      # make_golden helpers; the repo root is deliberately NOT added (see use_reference_packages)


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


def gen_variant_golden(out):
    """Reference outputs for the apply_model variants the fine-tune goldens do not reach (pins the oracle's
    apply_model_multi / only_mid_control / pre-train task switching):
      * ControlInferenceLDM, 2 LoRA banks, lora_weights (0.3, 0.7), non-trivial control_scales
        (cldm/cldm_ctrlora_inference.py:156-178), banks filled through the api.CtrLoRA load sequence;
      * ControlFinetuneLDM with only_mid_control=True (cldm/cldm.py:34-41);
      * ControlPretrainLDM.apply_model with cond['task'] switching between two banks (cldm_ctrlora_pretrain.py:95-111).
    Weights are oracle/arch.py key-addressed draws, inputs tests/golden/make_golden.py:inputs_for."""
    from ldm.util import instantiate_from_config
    from make_golden import inputs_for, ref_kwargs
    from oracle import arch
    cfg = arch.TINY
    check_key = lambda k: 'lora_layer' in k or 'zero_convs' in k or 'middle_block_out' in k or 'norm' in k

    def ldm(target_ldm, target_cn, cn_extra, drop=()):
        cn = ref_kwargs(cfg, True)
        for d in drop:
            cn.pop(d)
        cn.update(cn_extra)
        conf = dict(target=target_ldm, params=dict(
            linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000,
            first_stage_key="jpg", cond_stage_key="txt", control_key="hint", image_size=64, channels=4,
            cond_stage_trainable=False, conditioning_key="crossattn", monitor="val/loss_simple_ema",
            scale_factor=0.18215, use_ema=False, only_mid_control=False,
            control_stage_config=dict(target=target_cn, params=cn),
            unet_config=dict(target="cldm.cldm.ControlledUnetModel", params=ref_kwargs(cfg, False)),
            first_stage_config=dict(target="torch.nn.Identity"), cond_stage_config=dict(target="torch.nn.Identity")))
        m = instantiate_from_config(conf)
        m.encode_first_stage = lambda x: x
        m.get_first_stage_encoding = lambda z: z
        return m.eval()

    seed = 4
    inp = inputs_for(cfg, 2, 16, seed)
    sd_un = arch.make_state(arch.unet_shapes(cfg), seed)
    sd_a = arch.make_state(arch.controlnet_shapes(cfg), 4)
    sd_b5 = arch.make_state(arch.controlnet_shapes(cfg), 5)
    res = dict(meta=dict(B=2, H=16, seed=seed, seed_a=4, seed_b=5, weights=[0.3, 0.7],
                         scales=[0.5 + 0.1 * i for i in range(13)]))
    h2 = inp["hint_z"].flip(0)
    with torch.no_grad():
        # ---- 2-LoRA inference model: shared base (seed 4), bank 0 = LoRA file of seed 4, bank 1 = of seed 5
        m = ldm("cldm.cldm_ctrlora_inference.ControlInferenceLDM", "cldm.cldm_ctrlora_inference.ControlNetInference",
                dict(lora_rank=cfg.lora_rank, lora_num=2), drop=("ft_with_lora", "norm_trainable", "lora_rank"))
        m.model.diffusion_model.load_state_dict(sd_un, strict=True)
        pre = lambda sd: {"control_model." + k: v for k, v in sd.items()}
        m.load_state_dict({k: v for k, v in pre(sd_a).items() if not check_key(k)}, strict=False)
        for i, sd in enumerate((sd_a, sd_b5)):
            m.control_model.switch_lora(i)
            m.load_state_dict({k: v for k, v in pre(sd).items() if check_key(k)}, strict=False)
            m.control_model.copy_weights_to_switchable()
        m.lora_weights = list(res["meta"]["weights"])
        m.control_scales = list(res["meta"]["scales"])
        conds = [dict(c_crossattn=[inp["ctx"]], c_concat=[inp["hint_z"]]), dict(c_crossattn=[inp["ctx"]], c_concat=[h2])]
        res["eps_multi"] = m.apply_model(inp["z"], inp["t"], conds).clone()
        # ---- fine-tune model with only_mid_control
        f = ldm("cldm.cldm_ctrlora_finetune.ControlFinetuneLDM", "cldm.cldm_ctrlora_finetune.ControlNetFinetune", {})
        f.model.diffusion_model.load_state_dict(sd_un, strict=True)
        f.control_model.load_state_dict(sd_a, strict=True)
        f.only_mid_control = True
        cond = dict(c_crossattn=[inp["ctx"]], c_concat=[inp["hint_z"]])
        res["eps_only_mid"] = f.apply_model(inp["z"], inp["t"], cond).clone()
        # ---- pre-train model: two task banks, the task named in cond selects the bank
        pt = ldm("cldm.cldm_ctrlora_pretrain.ControlPretrainLDM", "cldm.cldm_ctrlora_pretrain.ControlNetPretrain",
                 dict(lora_rank=cfg.lora_rank, tasks=["hed", "canny"]), drop=("ft_with_lora", "norm_trainable", "lora_rank"))
        pt.model.diffusion_model.load_state_dict(sd_un, strict=True)
        for task, sd in (("hed", sd_a), ("canny", sd_b5)):   # tree (shared) weights from seed 4, bank LoRAs per task
            pt.control_model.switch_lora(task)
            sel = {k: v for k, v in sd.items() if "lora_layer" in k} if task == "canny" else sd
            pt.control_model.load_state_dict(sel, strict=False)
        for task in ("hed", "canny"):
            c = dict(cond); c["task"] = task
            res[f"eps_pretrain_{task}"] = pt.apply_model(inp["z"], inp["t"], c).clone()
        # ---- ragged shape through the fine-tune model: odd batch, non-square latent 24 x 16 (token counts 384 /
        #      96 / 24 / 6 per level: none a multiple of the 64-row attention / GEMM tiles)
        g = torch.Generator().manual_seed(77)
        rag = dict(z=torch.randn(3, 4, 24, 16, generator=g), hint_z=torch.randn(3, 4, 24, 16, generator=g) * 0.9,
                   ctx=torch.randn(3, 77, cfg.context_dim, generator=g), noise=torch.randn(3, 4, 24, 16, generator=g),
                   t=torch.tensor([0, 999, 417]))
        f.only_mid_control = False
        x_noisy = f.q_sample(rag["z"], rag["t"], rag["noise"])
        cond = dict(c_crossattn=[rag["ctx"]], c_concat=[rag["hint_z"]])
        res["ragged"] = dict(inputs=rag, x_noisy=x_noisy.clone(), eps=f.apply_model(x_noisy, rag["t"], cond).clone())
    out["variants"] = res
    print("[golden-next] variants: |eps_multi|=%.4f |eps_only_mid|=%.4f |eps_hed|=%.4f |eps_canny|=%.4f" % tuple(
        float(res[k].norm()) for k in ("eps_multi", "eps_only_mid", "eps_pretrain_hed", "eps_pretrain_canny")))


if __name__ == "__main__":
    assert os.path.isdir(REF), "run in the build container (needs /root/reference)"
    from make_golden import install_stubs, use_reference_packages
    install_stubs()
    os.chdir("/tmp")
    use_reference_packages()
    out = {}
    gen_ckpt_golden(out)
    gen_sampler_golden(out)
    gen_variant_golden(out)
    torch.save(out, f"{HERE}/next_rows.pt")
    print("[golden-next] wrote", f"{HERE}/next_rows.pt", os.path.getsize(f"{HERE}/next_rows.pt"), "bytes")
