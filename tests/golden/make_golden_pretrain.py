"""Golden vectors for Base-ControlNet PRE-TRAINING -- SURVEY.md 8(f3) -- from the UNMODIFIED reference.

    python tests/golden/make_golden_pretrain.py                 # writes tests/golden/pretrain.pt  (build container only)
    python tests/golden/make_golden_pretrain.py --lr 1e-4       # writes tests/golden/pretrain_lr1e-4.pt: the same three steps
                                                                # at a rate where a bf16 run stays on the fp32 trajectory

The reference's ControlPretrainLDM (tiny width, two tasks) runs THREE optimizer steps with the task sequence
hed, canny, hed through p_losses -> backward -> configure_optimizers().step(): AdamW over every control_model parameter
(cldm/cldm_ctrlora_pretrain.py:174-182).  Stored per step: the loss, and for EVERY control_model parameter a digest
(64 sampled entries + norm) of its gradient after the backward pass; plus digests of selected parameters after the three
steps.  `zero_grad(set_to_none=False)` reproduces the torch 1.13 / Lightning 1.5 behaviour the reference pins
(requirements.txt): a bank that has been trained once keeps being updated, with a zero gradient, in steps of other tasks.
Weights are key-addressed draws (oracle.arch), inputs tests/golden/make_golden.py:inputs_for.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

TASKS = ["hed", "canny"]
SEQ = ["hed", "canny", "hed"]
SEED, LR = 9, 1e-3


NS = 64


def sample_idx(numel, n=NS):
    return torch.linspace(0, numel - 1, min(n, numel)).long()


def digest(t, n=NS):
    """(l2 norm, n evenly spaced entries zero-padded to n) -- the index set is a function of the size (sample_idx)."""
    f = t.detach().float().flatten()
    v = torch.zeros(n)
    idx = sample_idx(f.numel(), n)
    v[:idx.numel()] = f[idx]
    return float(f.double().norm()), v


def pack(digests):
    """{name: digest or None} -> compact tensors."""
    names = sorted(digests)
    has = torch.tensor([digests[n] is not None for n in names])
    l2 = torch.tensor([digests[n][0] if digests[n] is not None else 0.0 for n in names], dtype=torch.float64)
    vals = torch.stack([digests[n][1] if digests[n] is not None else torch.zeros(NS) for n in names])
    return dict(names=names, has=has, l2=l2, vals=vals)


def unpack(p):
    return {n: ((float(p["l2"][i]), p["vals"][i]) if bool(p["has"][i]) else None) for i, n in enumerate(p["names"])}


def bank_state(cfg, task_seed):
    """LoRA factors of one task bank under ControlNetFinetune key names."""
    from oracle import arch
    return {k: v for k, v in arch.make_state(arch.controlnet_shapes(cfg), task_seed).items() if "lora_layer" in k}


def step_inputs(cfg, i):
    from make_golden import inputs_for
    return inputs_for(cfg, 2, 16, 100 + i)


if __name__ == "__main__":
    OUT = "pretrain.pt"
    if "--lr" in sys.argv:
        LR = float(sys.argv[sys.argv.index("--lr") + 1])
        OUT = f"pretrain_lr{sys.argv[sys.argv.index('--lr') + 1]}.pt"
    import make_golden as mg
    assert os.path.isdir(mg.REF)
    mg.install_stubs()
    os.chdir("/tmp")
    mg.use_reference_packages()
    from ldm.util import instantiate_from_config
    from oracle import arch
    cfg = arch.TINY
    cn = mg.ref_kwargs(cfg, True)
    for d in ("ft_with_lora", "norm_trainable", "lora_rank"):
        cn.pop(d)
    cn.update(lora_rank=cfg.lora_rank, tasks=TASKS)
    conf = dict(target="cldm.cldm_ctrlora_pretrain.ControlPretrainLDM", params=dict(
        linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000,
        first_stage_key="jpg", cond_stage_key="txt", control_key="hint", image_size=64, channels=4,
        cond_stage_trainable=False, conditioning_key="crossattn", monitor="val/loss_simple_ema",
        scale_factor=0.18215, use_ema=False, only_mid_control=False,
        control_stage_config=dict(target="cldm.cldm_ctrlora_pretrain.ControlNetPretrain", params=cn),
        unet_config=dict(target="cldm.cldm.ControlledUnetModel", params=mg.ref_kwargs(cfg, False)),
        first_stage_config=dict(target="torch.nn.Identity"), cond_stage_config=dict(target="torch.nn.Identity")))
    torch.manual_seed(0)
    m = instantiate_from_config(conf)
    m.encode_first_stage = lambda x: x
    m.get_first_stage_encoding = lambda z: z
    m.train()
    m.learning_rate = LR
    m.model.diffusion_model.load_state_dict(arch.make_state(arch.unet_shapes(cfg), SEED), strict=True)
    base = arch.make_state(arch.controlnet_shapes(cfg), SEED)
    cm = m.control_model
    cm.switch_lora("hed")
    cm.load_state_dict(base, strict=False)                       # tree + the hed bank
    cm.switch_lora("canny")
    cm.load_state_dict(bank_state(cfg, SEED + 1), strict=False)  # the canny bank
    os.makedirs("./tmp", exist_ok=True)
    opt = m.configure_optimizers()
    names = {id(p): n for n, p in cm.named_parameters()}
    out = dict(meta=dict(tasks=TASKS, seq=SEQ, seed=SEED, lr=LR, B=2, H=16), steps=[])
    before = {n: p.detach().clone() for n, p in cm.named_parameters()}
    for i, task in enumerate(SEQ):
        inp = step_inputs(cfg, i)
        cond = dict(c_crossattn=[inp["ctx"]], c_concat=[inp["hint_z"]], task=task)
        opt.zero_grad(set_to_none=False)
        loss, _ = m.p_losses(inp["z"], cond, inp["t"], noise=inp["noise"])
        loss.backward()
        # names AFTER the switch: the active bank's tensors appear under '<linear>.lora_layer.*', the others under loras_dict.*
        grads = {}
        for t in TASKS:
            for j, lora in enumerate(cm.loras_dict[t]):
                for part in ("down", "up"):
                    g = getattr(lora, part).weight.grad
                    grads[f"loras_dict.{t}.{j}.{part}.weight"] = None if g is None else digest(g)
        for n, p in cm.named_parameters():
            if "lora_layer" in n or n.startswith("loras_dict."):
                continue
            grads[n] = None if p.grad is None else digest(p.grad)
        opt.step()
        out["steps"].append(dict(task=task, loss=float(loss), grads=pack(grads)))
        print(f"[golden] step {i} task {task}: loss {float(loss):.6f}, {sum(g is not None for g in grads.values())} grads")
    after = {}
    for t in TASKS:
        for j, lora in enumerate(cm.loras_dict[t]):
            if j in (0, 5, 20):
                for part in ("down", "up"):
                    after[f"loras_dict.{t}.{j}.{part}.weight"] = digest(getattr(lora, part).weight)
    for n, p in cm.named_parameters():
        if "lora_layer" in n or n.startswith("loras_dict."):
            continue
        if any(s in n for s in ("input_blocks.0.0.", "input_blocks.1.0.in_layers", "input_blocks.1.0.emb_layers", "input_blocks.3.0.op",
                                "input_blocks.4.1.transformer_blocks.0.attn1.to_q", "input_blocks.4.1.proj_in", "middle_block.1.transformer_blocks.0.ff",
                                "time_embed.0", "zero_convs.3.", "middle_block.2.out_layers", "input_blocks.2.0.skip", "input_blocks.4.0.skip")):
            after[n] = digest(p)
    out["after"] = pack(after)
    torch.save(out, os.path.join(HERE, OUT))
    print(f"[golden] {OUT} written:", len(after), "parameter digests")
