"""Config-driven construction helpers (API of the reference's ldm/util.py:39-87)."""
import importlib
from inspect import isfunction


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if isfunction(d) else d


def count_params(model, verbose=False):
    total = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {total * 1.e-6:.2f} M params.")
    return total


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    mod = importlib.import_module(module)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config):
    """{'target': 'pkg.mod.Class', 'params': {...}} -> Class(**params)   (ldm/util.py:72-79)"""
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def log_txt_as_img(wh, xc, size=10):
    """Render a list of captions as images, (B, 3, H, W) in [-1, 1] (ldm/util.py:12-33; PIL's built-in font instead of the
    reference's bundled TTF)."""
    import numpy as np
    import torch
    from PIL import Image, ImageDraw, ImageFont
    out = []
    for cap in xc:
        txt = Image.new("RGB", wh, color="white")
        draw = ImageDraw.Draw(txt)
        nc = int(40 * (wh[0] / 256))
        lines = "\n".join(str(cap)[i:i + nc] for i in range(0, len(str(cap)), nc))
        draw.text((0, 0), lines, fill="black", font=ImageFont.load_default())
        out.append(np.array(txt).transpose(2, 0, 1) / 127.5 - 1.0)
    return torch.tensor(np.stack(out), dtype=torch.float32)
