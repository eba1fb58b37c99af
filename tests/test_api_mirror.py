"""Drop-in boundary: the cldm / ldm mirror builds from configs/*.yaml with the
reference's class paths and produces exactly the reference's state-dict keys and shapes
(tests/golden/keys_*.json were dumped from the unmodified reference modules)."""
import json
import os

import pytest
import torch
import yaml

from tests.util import GOLDEN, ROOT


def _cfg(name):
    with open(os.path.join(ROOT, "configs", name)) as f:
        return yaml.safe_load(f)


def _tiny(params, rank=32):
    p = dict(params)
    p.update(model_channels=64, context_dim=96)
    if "lora_rank" in p:
        p["lora_rank"] = rank
    return p


def test_yaml_targets_resolve_and_keys_match_reference_tiny():
    from ldm.util import instantiate_from_config
    cfg = _cfg("ctrlora_finetune_sd15_rank128.yaml")["model"]["params"]
    cn_cfg = dict(cfg["control_stage_config"]); cn_cfg["params"] = _tiny(cn_cfg["params"])
    un_cfg = dict(cfg["unet_config"]); un_cfg["params"] = _tiny(un_cfg["params"])
    cn = instantiate_from_config(cn_cfg)
    un = instantiate_from_config(un_cfg)
    keys = json.load(open(os.path.join(GOLDEN, "keys_tiny.json")))
    assert {k: list(v.shape) for k, v in cn.state_dict().items()} == keys["controlnet"]
    assert {k: list(v.shape) for k, v in un.state_dict().items()} == keys["unet"]
    assert type(cn).__module__ == "cldm.cldm_ctrlora_finetune" and type(un).__module__ == "cldm.cldm"


def test_full_ldm_builds_from_yaml_with_identity_encoders():
    from ldm.util import instantiate_from_config
    cfg = _cfg("ctrlora_finetune_sd15_rank32.yaml")["model"]
    p = cfg["params"]
    p["control_stage_config"]["params"] = _tiny(p["control_stage_config"]["params"])
    p["unet_config"]["params"] = _tiny(p["unet_config"]["params"])
    p["first_stage_config"] = {"target": "torch.nn.Identity"}
    p["cond_stage_config"] = {"target": "torch.nn.Identity"}
    model = instantiate_from_config(cfg)
    assert model.control_scales == [1.0] * 13 and model.num_timesteps == 1000
    names = model.trainable_names()
    assert len(names) == 246
    g = torch.load(os.path.join(GOLDEN, "model_tiny.pt"), weights_only=False)
    assert sorted(names) == sorted(g["trainable_names"])
    # schedule buffers bit-exact with the reference's DDPM.register_schedule
    s = torch.load(os.path.join(GOLDEN, "schedule.pt"), weights_only=False)["ddpm"]
    for k, v in s.items():
        assert torch.equal(getattr(model, k), v), k


def test_vae_and_clip_targets_import():
    from ldm.util import get_obj_from_str
    cfg = _cfg("ctrlora_finetune_sd15_rank128.yaml")["model"]["params"]
    vae = get_obj_from_str(cfg["first_stage_config"]["target"])(**cfg["first_stage_config"]["params"])
    n = sum(p.numel() for p in vae.encoder.parameters())
    assert abs(n - 34.16e6) < 0.1e6          # VAE encoder 34.2 M params
    z = vae.encode(torch.zeros(1, 3, 64, 64)).mode()
    assert z.shape == (1, 4, 8, 8)
    assert get_obj_from_str(cfg["cond_stage_config"]["target"]).__name__ == "FrozenCLIPEmbedder"


@pytest.mark.parametrize("name", ["inference/ctrlora_sd15_rank128_2loras.yaml", "ctrlora_pretrain_sd15_9tasks_rank128.yaml"])
def test_inference_and_pretrain_trees(name):
    from ldm.util import instantiate_from_config
    cs = _cfg(name)["model"]["params"]["control_stage_config"]
    cs["params"] = _tiny(cs["params"])
    if "tasks" in cs["params"]:
        cs["params"]["tasks"] = ["hed", "canny"]
    cn = instantiate_from_config(cs)
    n0 = len(cn.state_dict())
    if "tasks" in cs["params"]:
        assert n0 == 652                     # 652 -> 816 keys after switch_lora (2 tasks)
        cn.switch_lora("canny")
        assert len(cn.state_dict()) == 816
        with pytest.raises(AssertionError):
            cn.switch_lora("nope")
    else:
        assert n0 == 816                     # lora_num = 2: 816 -> 1062 after switch_lora(0)
        cn.switch_lora(0)
        assert len(cn.state_dict()) == 1062
        sd = cn.bank_state(1)
        keys = json.load(open(os.path.join(GOLDEN, "keys_tiny.json")))["controlnet"]
        assert {k: list(v.shape) for k, v in sd.items()} == keys


def test_lora_fuse_unfuse_roundtrip_cpu():
    from cldm.lora import LoRACompatibleLinear, LoRALinearLayer
    g = torch.load(os.path.join(GOLDEN, "lora.pt"), weights_only=False)
    lin = LoRACompatibleLinear(96, 64, lora_layer=LoRALinearLayer(96, 64, rank=32))
    lin.load_state_dict(g["state"])
    w0 = lin.weight.data.clone()
    lin._fuse_lora()
    assert lin.lora_layer is None
    assert float((lin.weight.data - g["w_fused"]).abs().max()) < 1e-6
    lin._unfuse_lora()
    assert float((lin.weight.data - w0).abs().max()) < 1e-6


def test_checkpoint_readers_and_create_model(tmp_path):
    """cldm.model: Lightning-wrapped / bare pickles and safetensors files read to the same flat dict;
    create_model builds the class named in the YAML (anchors and merge keys of configs/*.yaml resolved)."""
    from safetensors.torch import save_file
    from cldm.model import create_model, get_state_dict, load_state_dict
    sd = {"control_model.zero_convs.0.0.weight": torch.arange(6.0).reshape(2, 3), "logvar": torch.zeros(3)}
    torch.save({"state_dict": sd, "epoch": 3}, tmp_path / "lightning.ckpt")
    torch.save(sd, tmp_path / "bare.ckpt")
    save_file(sd, str(tmp_path / "weights.safetensors"))
    for name in ("lightning.ckpt", "bare.ckpt", "weights.safetensors"):
        got = load_state_dict(str(tmp_path / name), location="cpu")
        assert sorted(got) == sorted(sd) and all(torch.equal(got[k], sd[k]) for k in sd), name
    assert get_state_dict({"state_dict": sd}) is sd and get_state_dict(sd) is sd
    cfg = _cfg("ctrlora_finetune_sd15_rank32.yaml")
    a, b = cfg["model"]["params"]["control_stage_config"]["params"], cfg["model"]["params"]["unet_config"]["params"]
    assert a is not b and a["model_channels"] == b["model_channels"] == 320 and a["lora_rank"] == 32 and "lora_rank" not in b
    # a narrow copy of the YAML builds through create_model
    cfg["model"]["params"]["control_stage_config"]["params"] = _tiny(a)
    cfg["model"]["params"]["unet_config"]["params"] = _tiny(b)
    cfg["model"]["params"]["first_stage_config"] = {"target": "torch.nn.Identity"}
    cfg["model"]["params"]["cond_stage_config"] = {"target": "torch.nn.Identity"}
    path = tmp_path / "tiny.yaml"
    path.write_text(yaml.safe_dump(cfg))
    model = create_model(str(path))
    assert type(model).__name__ == "ControlFinetuneLDM" and next(model.parameters()).device.type == "cpu"


def test_switchable_holders_run_standalone_and_inference_net_tracks_the_active_bank():
    """cldm/switchable.py:17-20,37-40,58-61: a holder runs its active bank's layer, or its own weights when none is set;
    ControlNetInference.switch_lora(i) records the bank that ControlNetInference.forward executes
    (cldm/cldm_ctrlora_inference.py:100-130)."""
    import torch.nn as nn
    from cldm.switchable import SwitchableConv2d, SwitchableGroupNorm, SwitchableLayerNorm
    x = torch.randn(2, 4, 5, 6)
    conv = SwitchableConv2d(4, 8, 1)
    assert torch.allclose(conv(x), nn.functional.conv2d(x, conv.weight, conv.bias))
    bank = nn.Conv2d(4, 8, 1)
    conv.set_conv_layer(bank)
    assert torch.equal(conv(x), bank(x))
    gn, gbank = SwitchableGroupNorm(2, 4), nn.GroupNorm(2, 4)
    gbank.weight.data.fill_(3.0)
    assert torch.allclose(gn(x), nn.functional.group_norm(x, 2, gn.weight, gn.bias, gn.eps))
    gn.set_norm_layer(gbank)
    assert torch.equal(gn(x), gbank(x))
    ln, lbank = SwitchableLayerNorm(6), nn.LayerNorm(6)
    lbank.bias.data.fill_(0.5)
    ln.set_norm_layer(lbank)
    assert torch.equal(ln(x), lbank(x))
    ln.weight.data.fill_(2.0)
    ln.copy_weights()
    assert torch.equal(lbank.weight.data, ln.weight.data)
    from cldm.cldm_ctrlora_inference import ControlNetInference
    a = _cfg("inference/ctrlora_sd15_rank128_2loras.yaml")["model"]["params"]["control_stage_config"]["params"]
    net = ControlNetInference(**{**_tiny(a), "lora_rank": 8, "lora_num": 2})
    assert net._active_bank is None
    net.switch_lora(1)
    assert net._active_bank == 1
    lin = net.get_submodule(net._linear_names[0])
    assert lin.lora_layer is net.loras_list[1][0]
    assert "forward" in ControlNetInference.__dict__      # the module itself is callable after switch_lora (GPU test: test_gpu_parity.py)
