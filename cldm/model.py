"""Model construction and checkpoint reading with the call signatures of the reference's `cldm.model`
(`create_model(config_path)`, `load_state_dict(ckpt_path, location)`, `get_state_dict(d)`; cldm/model.py:1-28).

Configs are parsed with PyYAML (the trees are plain mappings; anchors / merge keys of configs/*.yaml resolve at load);
checkpoints may be Lightning pickles (`{"state_dict": ...}`, possibly nested once more), bare state dicts, or
safetensors files.
"""
from pathlib import Path

import torch
import yaml

from ldm.util import instantiate_from_config


def get_state_dict(d):
    """Unwrap one `{"state_dict": ...}` level if present."""
    return d["state_dict"] if "state_dict" in d else d


def _read_tensors(path: Path, location: str):
    if path.suffix.lower() == ".safetensors":
        from safetensors.torch import load_file
        return load_file(str(path), device=location)
    blob = torch.load(str(path), map_location=torch.device(location), weights_only=False)
    return get_state_dict(blob)


def load_state_dict(ckpt_path, location="cpu"):
    tensors = get_state_dict(_read_tensors(Path(ckpt_path), location))
    print(f"[cldm.model] {len(tensors)} tensors read from {ckpt_path}")
    return tensors


def create_model(config_path):
    """cldm/model.py:23-28.  The parameters come out with torch's default distributions, drawn the fast way
    (ctrlora_amd/fastinit.py: a checkpoint overwrites them in every script anyway, and eight ranks constructing the model with
    per-layer kaiming_uniform_ calls spend minutes of shared host cores).  CTRLORA_FAST_INIT=0 keeps torch's own per-layer
    draws (as bench.build_model does)."""
    import os
    from ctrlora_amd.fastinit import fill_default_init, skip_default_init
    with open(config_path) as fh:
        tree = yaml.safe_load(fh)
    if os.environ.get("CTRLORA_FAST_INIT", "1") == "0":
        net = instantiate_from_config(tree["model"])
    else:
        with skip_default_init():
            net = instantiate_from_config(tree["model"])
        fill_default_init(net, seed=int(torch.initial_seed() % (1 << 30)))
    print(f"[cldm.model] built {type(net).__name__} from {config_path}")
    return net.cpu()
