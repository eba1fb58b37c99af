"""Host-side pieces of the reference's training entry point, CPU only: the paired-image dataset,
checkpoint initialisation rules of scripts/train_ctrlora_finetune.py, the Lightning-free loop (gradient accumulation,
step counting, checkpoint callback schedule, resume) and the image-grid helper."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import yaml

from tests.util import ROOT


def _script(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------ dataset

def test_custom_dataset_layout_ranges_and_prompt_dropout(tmp_path):
    from PIL import Image
    from datasets.custom_dataset import CustomDataset
    (tmp_path / "source").mkdir(); (tmp_path / "target").mkdir()
    rng = np.random.RandomState(0)
    imgs = {}
    for i in range(3):
        for d in ("source", "target"):
            if d == "target" and i == 2:
                continue                                   # third record has no target file: skipped
            a = rng.randint(0, 256, (8, 12, 3), dtype=np.uint8)
            Image.fromarray(a).save(tmp_path / d / f"{i:04d}.png")
            imgs[(d, i)] = a
    with open(tmp_path / "prompt.json", "w") as f:
        for i in range(3):
            f.write(json.dumps({"source": f"source/{i:04d}.png", "target": f"target/{i:04d}.png", "prompt": f"p{i}"}) + "\n")
    ds = CustomDataset(str(tmp_path), drop_rate=0.0)
    assert len(ds) == 2
    it = ds[1]
    assert it["txt"] == "p1" and it["jpg"].dtype == np.float32 and it["hint"].shape == (8, 12, 3)
    assert np.allclose(it["hint"], imgs[("source", 1)].astype(np.float32) / 255.0)
    assert np.allclose(it["jpg"], imgs[("target", 1)].astype(np.float32) / 127.5 - 1.0)
    assert it["jpg"].min() >= -1.0 and it["jpg"].max() <= 1.0
    np.random.seed(0)
    dropped = sum(CustomDataset(str(tmp_path), drop_rate=0.5)[0]["txt"] == "" for _ in range(200))
    assert 70 < dropped < 130
    with pytest.raises(FileNotFoundError):
        CustomDataset(str(tmp_path / "source"))


# ------------------------------------------------------------------ checkpoint initialisation

def _tiny_ldm(seed):
    from ldm.util import instantiate_from_config
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "ctrlora_finetune_sd15_rank32.yaml")))["model"]
    p = cfg["params"]
    for k in ("control_stage_config", "unet_config"):
        q = dict(p[k]["params"]); q.update(model_channels=64, context_dim=96); p[k]["params"] = q
    p["first_stage_config"] = {"target": "torch.nn.Identity"}
    p["cond_stage_config"] = {"target": "torch.nn.Identity"}
    torch.manual_seed(seed)
    return instantiate_from_config(cfg)


def test_init_weights_takes_sd_and_base_controlnet_but_keeps_fresh_loras(tmp_path):
    tool = _script("train_ctrlora_finetune")
    model, donor = _tiny_ldm(0), _tiny_ldm(1)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    noise = lambda v: torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone()
    sd_ckpt = {k: noise(v) for k, v in donor.state_dict().items() if not k.startswith("control_model.")}
    sd_ckpt["model_ema.decay"] = torch.zeros(1)                            # not in the model: reported missing
    cn_ckpt = {k: noise(v) for k, v in donor.state_dict().items() if k.startswith("control_model.")}
    cn_ckpt["control_model.input_hint_block.0.weight"] = torch.zeros(1)      # base ControlNets carry it; CtrLoRA drops it
    cn_ckpt["lr_scheduler"] = torch.zeros(1)                               # not a control_model key: ignored entirely
    (cp_sd, miss_sd), (cp_cn, miss_cn) = tool.init_weights(model, sd_ckpt, cn_ckpt, report_dir=str(tmp_path))
    after = model.state_dict()
    assert miss_sd == ["model_ema.decay"] and miss_cn == ["control_model.input_hint_block.0.weight"]
    assert all(torch.equal(after[k], sd_ckpt[k]) for k in cp_sd) and len(cp_sd) == len(sd_ckpt) - 1
    lora = [k for k in cn_ckpt if "lora" in k and k in after]
    assert len(lora) == 164 and not set(lora) & set(cp_cn)
    assert all(torch.equal(after[k], before[k]) for k in lora)             # fresh LoRA layers survive
    assert all(torch.equal(after[k], cn_ckpt[k]) for k in cp_cn)
    assert len(cp_cn) == len([k for k in cn_ckpt if k.startswith("control_model.") and k in after]) - 164
    for name in ("finetune_missing_keys_sd", "finetune_copied_keys_sd", "finetune_missing_keys_cn", "finetune_copied_keys_cn"):
        assert os.path.isfile(tmp_path / (name + ".txt"))
    assert open(tmp_path / "finetune_copied_keys_cn.txt").read().split("\n") == cp_cn


def test_train_script_cli_matches_reference_flags():
    tool = _script("train_ctrlora_finetune")
    a = tool.get_parser().parse_args(["--dataroot", "d", "--config", "c.yaml", "--sd_ckpt", "s", "--cn_ckpt", "n"])
    assert (a.lr, a.bs, a.max_steps, a.gradacc, a.precision, a.drop_rate) == (1e-5, 1, 100000, 1, 32, 0.3)
    assert (a.img_logger_freq, a.ckpt_logger_freq, a.subset, a.multigen20m, a.save_memory) == (1000, 1000, 0, False, False)
    with pytest.raises(SystemExit):
        tool.get_parser().parse_args(["--dataroot", "d", "--config", "c", "--sd_ckpt", "s", "--cn_ckpt", "n", "--task", "nope"])


# ------------------------------------------------------------------ the loop

class _Toy(nn.Module):
    """LightningModule-shaped toy: one weight vector, loss = mean((x.w - y)^2)."""

    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.zeros(3))
        self.seen = []

    def training_step(self, batch, batch_idx):
        x, y = batch
        self.seen.append(int(batch_idx))
        return ((x @ self.w - y) ** 2).mean()

    def configure_optimizers(self):
        return torch.optim.SGD(self.parameters(), lr=0.1)


def _loader(n_batches=4, bs=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n_batches * bs, 3, generator=g); y = x @ torch.tensor([1.0, -2.0, 0.5])
    return torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=bs, shuffle=False)


def test_trainer_gradient_accumulation_steps_checkpoints_and_resume(tmp_path):
    from cldm.logger import CheckpointEveryNSteps
    from ctrlora_amd.trainer import Trainer
    loader = _loader()
    model = _Toy()
    tr = Trainer(max_steps=5, accumulate_grad_batches=2, precision=32, default_root_dir=str(tmp_path / "run"),
                 callbacks=[CheckpointEveryNSteps(save_step_frequency=2)], device="cpu", log_every_n_steps=1)
    tr.fit(model, loader)
    assert tr.global_step == 5 and len(model.seen) == 10                   # 2 micro-batches per optimizer step
    assert model.seen == [0, 1, 2, 3, 0, 1, 2, 3, 0, 1] and tr.current_epoch == 3
    # reference: hand-rolled SGD with the micro-batch losses divided by the accumulation factor
    ref = torch.zeros(3, requires_grad=True)
    batches = [b for _ in range(3) for b in loader][:10]
    for s in range(5):
        gsum = torch.zeros(3)
        for x, y in batches[2 * s: 2 * s + 2]:
            gsum += torch.autograd.grad((((x @ ref - y) ** 2).mean()) / 2, ref)[0]
        ref = (ref - 0.1 * gsum).detach().requires_grad_(True)
    assert torch.allclose(model.w.detach(), ref.detach(), atol=1e-6)
    # CheckpointEveryNSteps rule: global_step == 0 or (global_step + 1) % 2 == 0, evaluated after every micro-batch
    files = sorted(os.listdir(tr.checkpoint_callback.dirpath))
    steps = sorted({int(f.split("global_step=")[1].split(".")[0]) for f in files})
    assert steps == [0, 1, 3, 5] and all(f.startswith("N-Step-Checkpoint_epoch=") for f in files)
    ck = torch.load(os.path.join(tr.checkpoint_callback.dirpath, [f for f in files if "global_step=3" in f][0]),
                    weights_only=False)
    assert set(ck) >= {"state_dict", "global_step", "epoch", "optimizer_states"} and ck["global_step"] == 3
    # resume: a new trainer continues the step count from the checkpoint
    m2 = _Toy()
    t2 = Trainer(max_steps=4, accumulate_grad_batches=2, default_root_dir=str(tmp_path / "run2"), device="cpu")
    t2.fit(m2, loader, ckpt_path=os.path.join(tr.checkpoint_callback.dirpath, [f for f in files if "global_step=3" in f][0]))
    assert t2.global_step == 4 and len(m2.seen) == 2


def test_image_grid_helper_matches_make_grid_layout():
    from cldm.logger import _grid
    imgs = torch.arange(5 * 1 * 2 * 3, dtype=torch.float32).reshape(5, 1, 2, 3) + 1
    g = _grid(imgs, nrow=4, pad=2)
    assert g.shape == (1, 2 * (2 + 2) + 2, 4 * (3 + 2) + 2)
    assert torch.equal(g[:, 2:4, 2:5], imgs[0]) and torch.equal(g[:, 2:4, 7:10], imgs[1])
    assert torch.equal(g[:, 6:8, 2:5], imgs[4]) and float(g[:, 6:8, 7:].abs().sum()) == 0.0
    assert float(g[:, :2].abs().sum()) == 0.0


# ------------------------------------------------------------------ scripts/sample.py

def test_sample_script_geometry_cli_and_output_layout(tmp_path):
    tool = _script("sample")
    # short side -> 512, both sides to multiples of 64 (annotator/util.py:28-38)
    assert tool.target_size(480, 640)[:2] == (512, 704) and tool.target_size(1024, 1024)[:2] == (512, 512)
    assert tool.target_size(300, 1000)[:2] == (512, 1728)
    assert tool.resize_image(np.zeros((480, 640, 3), np.uint8)).shape == (512, 704, 3)
    a = tool.get_parser().parse_args(["--dataroot", "d", "--config", "c", "--ckpt", "k", "--save_dir", str(tmp_path / "out")])
    assert (a.n_samples, a.ddim_steps, a.ddim_eta, a.strength, a.cfg, a.empty_prompt) == (10, 50, 0.0, 1.0, 7.5, False)

    calls = []

    class FakeModel:
        control_scales = None

        def get_learned_conditioning(self, prompts):
            return ("ctx", tuple(prompts))

        def decode_first_stage(self, z):
            return torch.zeros(z.shape[0], 3, z.shape[2] * 8, z.shape[3] * 8)

    class FakeSampler:
        def sample(self, S, B, shape, cond, verbose, eta, unconditional_guidance_scale, unconditional_conditioning):
            calls.append((S, B, shape, cond, eta, unconditional_guidance_scale, unconditional_conditioning))
            return torch.zeros(B, *shape), None

    items = [dict(jpg=np.zeros((100, 150, 3), np.float32), txt=" a cat ", hint=np.ones((100, 150, 3), np.float32) * 0.5),
             dict(jpg=np.zeros((64, 64, 3), np.float32), txt="dog", hint=np.zeros((64, 64, 3), np.float32))]
    a.task, a.strength, a.ddim_steps = "canny", 0.8, 7
    m = FakeModel()
    n = tool.sample_dataset(m, FakeSampler(), items, a, device="cpu")
    assert n == 2 and m.control_scales == [0.8] * 13
    S, B, shape, cond, eta, cfg, unc = calls[0]
    assert (S, B, shape, eta, cfg) == (7, 1, (4, 64, 96), 0.0, 7.5)            # 100 x 150 -> 512 x 768 -> latent 64 x 96
    assert cond["task"] == "canny" and cond["c_crossattn"] == [("ctx", (" a cat ",))] and unc["c_crossattn"] == [("ctx", ("",))]
    assert cond["c_concat"][0].shape == (1, 3, 512, 768) and cond["c_concat"][0] is unc["c_concat"][0]
    assert abs(float(cond["c_concat"][0].mean()) - 127 / 255) < 1e-3
    out = tmp_path / "out"
    for sub in ("sample", "control", "img"):
        assert sorted(os.listdir(out / sub)) == ["0.png", "1.png"]
    assert open(out / "prompt.txt").read() == "a cat\ndog\n"


def test_multigen20m_reader_and_collate(tmp_path):
    """datasets.multigen20m.MultiGen20M (reference datasets/multigen20m.py:20-142): json-lines index, conditions/ +
    images/ layout, SAME square crop on both images, 512 x 512 float outputs in the documented ranges, unreadable
    samples skipped forward, prompt dropout; datasets.dataset_collate.collate_fn drops failed samples."""
    import json
    import random
    from PIL import Image
    from datasets.dataset_collate import collate_fn
    from datasets.multigen20m import MultiGen20M
    root = tmp_path
    for d in ("json_files", "conditions", "images"):
        (root / d).mkdir()
    rng = np.random.RandomState(0)
    lines = []
    for i in range(4):
        w, h = (640, 384) if i % 2 == 0 else (300, 420)
        img = rng.randint(0, 255, (h, w, 3), dtype=np.uint8)
        img[:, : w // 2, 0] = 255                                    # left half marked in the red channel
        Image.fromarray(img).save(root / "images" / f"a{i}.png")
        Image.fromarray(img).save(root / "conditions" / f"c{i}.png")
        lines.append(json.dumps({"source": f"./a{i}.png", "prompt": f"p{i}", "control_canny": f"c{i}.png"}))
    lines.insert(1, json.dumps({"source": "./missing.png", "prompt": "x", "control_canny": "missing.png"}))
    (root / "json_files" / "aesthetics_plus_all_group_canny_all.json").write_text("\n".join(lines) + "\n")
    ds = MultiGen20M(str(root / "json_files" / "aesthetics_plus_all_group_canny_all.json"), str(root), "canny",
                     drop_rate=0.0, random_cropping=False)
    assert len(ds) == 5
    it = ds[0]
    assert it["task"] == "control_canny" and it["txt"] == "p0"
    assert it["jpg"].shape == (512, 512, 3) and it["hint"].shape == (512, 512, 3)
    assert it["jpg"].dtype == np.float32 and -1.0 <= it["jpg"].min() and it["jpg"].max() <= 1.0
    assert 0.0 <= it["hint"].min() and it["hint"].max() <= 1.0
    # centred crop of a 640 x 384 image keeps columns 128..512: the red marker covers the left (320-128)/384 = 1/2
    red = it["hint"][:, :, 0] > 0.99
    assert abs(red[:, :250].mean() - 1.0) < 0.02 and red[:, 270:].mean() < 0.1
    # same crop on the target: same marker position
    assert abs(((it["jpg"][:, :250, 0] + 1) / 2 > 0.99).mean() - 1.0) < 0.02
    # the unreadable sample falls through to the next one
    assert ds[1]["txt"] == "p1"
    random.seed(0)
    dropped = MultiGen20M(str(root / "json_files" / "aesthetics_plus_all_group_canny_all.json"), str(root), "canny", drop_rate=1.0)
    assert dropped[0]["txt"] == ""
    batch = collate_fn([ds[0], dict(jpg=None, hint=None, txt="", task="control_canny"), ds[2]])
    assert batch["jpg"].shape == (2, 512, 512, 3) and batch["txt"] == ["p0", "p1"] and batch["task"] == ["control_canny"] * 2
    assert collate_fn([dict(jpg=None, hint=None)]) is None
