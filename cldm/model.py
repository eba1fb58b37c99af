"""create_model / load_state_dict (API of the reference's cldm/model.py:1-28).  YAML is read with
PyYAML (the reference uses OmegaConf; the config trees are plain dict/list data)."""
import os

import torch
import yaml

from ldm.util import instantiate_from_config


def get_state_dict(d):
    return d.get("state_dict", d)


def load_state_dict(ckpt_path, location="cpu"):
    _, ext = os.path.splitext(ckpt_path)
    if ext.lower() == ".safetensors":
        import safetensors.torch
        sd = safetensors.torch.load_file(ckpt_path, device=location)
    else:
        sd = get_state_dict(torch.load(ckpt_path, map_location=torch.device(location), weights_only=False))
    sd = get_state_dict(sd)
    print(f"Loaded state_dict from [{ckpt_path}]")
    return sd


def create_model(config_path):
    with open(config_path) as f:
        config = yaml.safe_load(f)
    model = instantiate_from_config(config["model"]).cpu()
    print(f"Loaded model config from [{config_path}]")
    return model
