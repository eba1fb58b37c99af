"""SURVEY.md 8(f) "next" rows that are host logic, checked on CPU against fixtures produced by the UNMODIFIED
reference (tests/golden/make_golden_next.py -> tests/golden/next_rows.pt):

  f2  LoRA / Base-ControlNet checkpoint extraction (scripts/tool_extract_weights.py), combination
      (scripts/tool_combine_weights.py) and the api.CtrLoRA.create_model load sequence into the switchable banks;
  f3  the multi-task batch scheduler (datasets/multi_task_scheduler.py) index streams, plain and 2-rank.
"""
import importlib.util
import os

import numpy as np
import pytest
import torch

from tests.golden.make_golden_next import checksum, key_tensor, tiny_control_params
from tests.util import GOLDEN, ROOT


@pytest.fixture(scope="module")
def gold():
    return torch.load(os.path.join(GOLDEN, "next_rows.pt"), weights_only=False)


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _build(target, extra):
    from ldm.util import instantiate_from_config
    return instantiate_from_config(dict(target=target, params=tiny_control_params(extra)))


def _close(a, b):
    return all(abs(x - y) <= 1e-9 * max(1.0, abs(y)) for x, y in zip(a, b))


# ------------------------------------------------------------------ f2: extraction / combination

def test_extract_lora_and_control_select_the_reference_key_sets(gold):
    tool = _load("scripts/tool_extract_weights.py", "tool_extract_weights")
    ft = _build("cldm.cldm_ctrlora_finetune.ControlNetFinetune", dict(ft_with_lora=True, lora_rank=32, norm_trainable=True))
    full = {"control_model." + k: v for k, v in ft.state_dict().items()}
    full.update({"model.diffusion_model.fake.weight": torch.zeros(1), "first_stage_model.norm.weight": torch.zeros(1),
                 "cond_stage_model.transformer.final_layer_norm.weight": torch.zeros(1), "logvar": torch.zeros(1)})
    assert sorted(tool.extract_lora(full)) == gold["finetune_extract_lora_keys"]
    assert sorted(tool.extract_control(full)) == gold["finetune_extract_control_keys"]
    assert len(gold["finetune_extract_lora_keys"]) == 246      # 164 LoRA + 26 zero-conv + 56 norm


def test_per_task_lora_files_from_a_pretrain_model_match_reference(gold):
    tool = _load("scripts/tool_extract_weights.py", "tool_extract_weights")
    pre = _build("cldm.cldm_ctrlora_pretrain.ControlNetPretrain", dict(lora_rank=32, tasks=["hed", "canny"]))
    pre.load_state_dict({k: key_tensor(k, v.shape, "pretrain") for k, v in pre.state_dict().items()}, strict=True)
    for task in pre.tasks:
        pre.switch_lora(task)
        sd = {"control_model." + k: v for k, v in pre.state_dict().items()}
        ex = tool.extract_lora(sd)
        ref = gold["pretrain_task_files"][task]
        assert sorted(ex) == sorted(ref)
        bad = [k for k in ref if not _close(checksum(ex[k]), ref[k])]
        assert not bad, f"task {task}: bank -> tree aliasing differs from the reference at {bad[:4]}"
    assert sorted(tool.extract_control({"control_model." + k: v for k, v in pre.state_dict().items()})) == \
        gold["pretrain_extract_control_keys"]
    # the two task files differ exactly in the LoRA entries
    a, b = gold["pretrain_task_files"]["hed"], gold["pretrain_task_files"]["canny"]
    assert {k for k in a if a[k] != b[k]} == {k for k in a if "lora_layer" in k}


def test_from_base_flow_on_a_full_pretrain_ldm(gold, tmp_path):
    """extract_task_loras = the --from_base branch: strict load of a checkpoint saved with a bank aliased in,
    then one LoRA file per task."""
    import yaml
    from ldm.util import instantiate_from_config
    tool = _load("scripts/tool_extract_weights.py", "tool_extract_weights")
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "ctrlora_pretrain_sd15_9tasks_rank128.yaml")))["model"]
    p = cfg["params"]
    p["control_stage_config"]["params"] = tiny_control_params(dict(lora_rank=32, tasks=["hed", "canny"]))
    un = dict(p["unet_config"]["params"]); un.update(model_channels=64, context_dim=96)
    p["unet_config"]["params"] = un
    p["first_stage_config"] = {"target": "torch.nn.Identity"}
    p["cond_stage_config"] = {"target": "torch.nn.Identity"}
    src = instantiate_from_config(cfg)
    cm = src.control_model
    cm.load_state_dict({k: key_tensor(k, v.shape, "pretrain") for k, v in cm.state_dict().items()}, strict=True)
    cm.switch_lora("canny")
    ckpt = {k: v.clone() for k, v in src.state_dict().items()}      # what Lightning would have saved
    dst = instantiate_from_config(cfg)
    files = tool.extract_task_loras(dst, ckpt)
    assert list(files) == ["hed", "canny"]
    for task, sd in files.items():
        ref = gold["pretrain_task_files"][task]
        assert sorted(sd) == sorted(ref)
        assert all(_close(checksum(sd[k]), ref[k]) for k in ref)


def test_combine_weights_precedence_and_logvar():
    tool = _load("scripts/tool_combine_weights.py", "tool_combine_weights")
    sd = {"model.a": torch.ones(1), "model_ema.a": torch.ones(1), "control_model.x": torch.zeros(1)}
    base = {"control_model.x": torch.full((1,), 2.0), "control_model.y": torch.full((1,), 3.0)}
    lora = {"control_model.y": torch.full((1,), 4.0)}
    out = tool.combine(sd, base, lora)
    assert sorted(out) == ["control_model.x", "control_model.y", "logvar", "model.a"]
    assert float(out["control_model.x"]) == 2.0 and float(out["control_model.y"]) == 4.0
    assert out["logvar"].shape == (1000,) and float(out["logvar"].abs().sum()) == 0.0


# ------------------------------------------------------------------ f2: api.CtrLoRA.create_model load sequence

def test_api_load_sequence_fills_the_switchable_banks_like_the_reference(gold):
    import api
    ft = _build("cldm.cldm_ctrlora_finetune.ControlNetFinetune", dict(ft_with_lora=True, lora_rank=32, norm_trainable=True))
    inf = _build("cldm.cldm_ctrlora_inference.ControlNetInference", dict(lora_rank=32, lora_num=2))
    inf.load_state_dict({k: key_tensor(k, v.shape, "inference-init") for k, v in inf.state_dict().items()}, strict=True)

    class Holder(torch.nn.Module):
        def __init__(self, cm):
            super().__init__()
            self.control_model = cm

    holder = Holder(inf)
    shapes = {k: v.shape for k, v in ft.state_dict().items()}
    base = {"control_model." + k: key_tensor(k, s, "basecn") for k, s in shapes.items()}
    loras = [{"control_model." + k: key_tensor(k, s, f"lora{i}") for k, s in shapes.items()} for i in range(2)]
    ctr = api.CtrLoRA(num_loras=2)
    assert sum(ctr.check_key(k) for k in loras[0]) == gold["api_num_lora_keys"] == 246
    ctr.load_weights(holder, cn_state_dict=base, lora_state_dicts=loras)
    state = inf.state_dict()
    ref = gold["api_final_state"]
    assert sorted(state) == sorted(ref) and len(ref) == 1062
    bad = [k for k in ref if not _close(checksum(state[k]), ref[k])]
    assert not bad, f"{len(bad)} entries differ from the reference load sequence, e.g. {bad[:4]}"
    # each bank holds its own LoRA file: bank_state(i) is that file re-keyed onto the plain ControlNet tree
    for i in range(2):
        bank = inf.bank_state(i)
        for k in ("zero_convs.0.0.weight", "input_blocks.1.1.norm.weight",
                  "input_blocks.1.1.transformer_blocks.0.attn1.to_q.lora_layer.up.weight"):
            assert torch.equal(bank[k], loras[i]["control_model." + k]), (i, k)
        assert torch.equal(bank["input_blocks.1.0.in_layers.2.weight"], base["control_model.input_blocks.1.0.in_layers.2.weight"])


def test_api_image_helpers():
    import api
    g = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert api.hwc3(g).shape == (3, 4, 3) and np.array_equal(api.hwc3(g)[:, :, 2], g)
    rgba = np.zeros((2, 2, 4), np.uint8); rgba[..., 0] = 200; rgba[..., 3] = 0
    assert np.array_equal(api.hwc3(rgba), np.full((2, 2, 3), 255, np.uint8))        # transparent -> white
    a, b = api.center_crop_to_common(np.zeros((10, 7, 3), np.uint8), np.zeros((6, 9, 3), np.uint8))
    assert a.shape == b.shape == (6, 7, 3)
    with pytest.raises(ValueError):
        api.CtrLoRA(num_loras=3)


# ------------------------------------------------------------------ f3: multi-task batch scheduler

class _DS(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


def test_batch_scheduler_streams_match_reference(gold, monkeypatch):
    import torch.distributed as dist
    from datasets.multi_task_scheduler import BatchSchedulerSampler
    g = gold["sampler"]
    ds = torch.utils.data.ConcatDataset([_DS(n) for n in g["sizes"]])
    bs = g["batch_size"]
    for shuffle in (False, True):
        torch.manual_seed(1234); np.random.seed(4321)
        s = BatchSchedulerSampler(ds, batch_size=bs, distributed=False, shuffle=shuffle)
        ref = g["streams"][f"plain_shuffle{int(shuffle)}"]
        assert len(s) == ref["len"] and list(iter(s)) == ref["idx"]
    monkeypatch.setattr(dist, "is_available", lambda: True)
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    for rank in (0, 1):
        monkeypatch.setattr(dist, "get_rank", lambda group=None, r=rank: r)
        for shuffle in (False, True):
            torch.manual_seed(1234); np.random.seed(4321 + rank)
            s = BatchSchedulerSampler(ds, batch_size=bs, distributed=True, shuffle=shuffle)
            ref = g["streams"][f"dist_rank{rank}_shuffle{int(shuffle)}"]
            assert len(s) == ref["len"] and list(iter(s)) == ref["idx"], (rank, shuffle)


def test_batch_scheduler_batches_are_single_task():
    from datasets.multi_task_scheduler import BatchSchedulerSampler
    sizes = [11, 4, 9, 6]
    ds = torch.utils.data.ConcatDataset([_DS(n) for n in sizes])
    s = BatchSchedulerSampler(ds, batch_size=3, distributed=False, shuffle=True)
    idx = list(iter(s))
    assert len(idx) == len(s)
    bounds = np.cumsum([0] + sizes)
    task_of = lambda i: int(np.searchsorted(bounds, i, side="right") - 1)
    for b in range(0, len(idx), 3):
        assert len({task_of(i) for i in idx[b:b + 3]}) == 1
    # the largest task is covered at least once per epoch
    assert {i for i in idx if task_of(i) == 0} == set(range(11))


# ------------------------------------------------------------------ f1 (minimum): hint encode leaves the DDIM loop

def test_hint_posterior_is_encoded_once_per_sampling_scope_and_resampled_every_call():
    """the reference VAE-encodes the condition image in EVERY apply_model call (2 x S per
    sampling run).  Inside ControlLDM.hint_cache() the encoder runs once per image, while each call still draws
    its own posterior sample (same distribution as the reference); outside a scope nothing is cached."""
    import yaml
    from ldm.util import instantiate_from_config
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "inference", "ctrlora_sd15_rank128_2loras.yaml")))["model"]
    p = cfg["params"]
    p["control_stage_config"]["params"] = tiny_control_params(dict(lora_rank=32, lora_num=2))
    un = dict(p["unet_config"]["params"]); un.update(model_channels=64, context_dim=96)
    p["unet_config"]["params"] = un
    p["first_stage_config"] = {"target": "torch.nn.Identity"}
    p["cond_stage_config"] = {"target": "torch.nn.Identity"}
    model = instantiate_from_config(cfg).eval()

    class Posterior:
        def __init__(self, mean):
            self.mean = mean

        def sample(self):
            return self.mean + 0.5 * torch.randn_like(self.mean)

    class CountingVAE(torch.nn.Module):
        calls = 0

        def encode(self, x):
            CountingVAE.calls += 1
            return Posterior(torch.nn.functional.avg_pool2d(x, 8)[:, :1].repeat(1, 4, 1, 1))

    model.first_stage_model = CountingVAE()
    img_a, img_b = torch.rand(2, 3, 64, 64), torch.rand(2, 3, 64, 64)
    cond_a, unc_a = {"c_concat": [img_a]}, {"c_concat": [img_a]}      # api.py: cond / un_cond share the image tensor
    cond_b = {"c_concat": [img_b]}
    lat4 = {"c_concat": [torch.randn(2, 4, 8, 8)]}
    # no scope: the reference behaviour, one encode per call
    model._hint_latent(cond_a); model._hint_latent(cond_a)
    assert CountingVAE.calls == 2
    CountingVAE.calls = 0
    with model.hint_cache():
        zs = [model._hint_latent(c) for c in (cond_a, unc_a, cond_b, cond_a, unc_a, cond_b)]
        assert CountingVAE.calls == 2                                   # one per distinct image
        assert zs[0].shape == (2, 4, 8, 8)
        assert not torch.equal(zs[0], zs[3])                            # fresh posterior sample every call
        mean_a = model.scale_factor * torch.nn.functional.avg_pool2d(img_a, 8)[:, :1].repeat(1, 4, 1, 1)
        assert float((torch.stack([model._hint_latent(cond_a) for _ in range(200)]).mean(0) - mean_a).abs().max()) < 0.05
        assert model._hint_latent(lat4) is not None and CountingVAE.calls == 2   # latents pass through
        with model.hint_cache():                                        # nested scopes share the cache
            model._hint_latent(cond_b)
            assert CountingVAE.calls == 2
        assert "_hint_cache" in model.__dict__
    assert "_hint_cache" not in model.__dict__
    model._hint_latent(cond_a)
    assert CountingVAE.calls == 3
