"""Model construction without torch's per-layer random draws.

`nn.Linear` / `nn.Conv2d` constructors call `kaiming_uniform_` on their own parameters: for the SD1.5 UNet + ControlNet (1.3 G
parameters) that is ~25 s of single-threaded host time per build -- per RANK of a multi-GPU job, per test of the GPU suite, and
wasted whenever a checkpoint overwrites the values (cldm.model.create_model -> load_state_dict, as every script of the reference
does: scripts/train_ctrlora_finetune.py:96-104).  `skip_default_init()` makes the constructors mark their parameters "not drawn yet" (NaN: a memset) instead,
`fill_default_init()` then gives every parameter that is STILL marked the SAME distribution torch would have (weight and bias ~ U(-1 / sqrt(fan_in),
1 / sqrt(fan_in)): kaiming_uniform_ with a = sqrt 5) as a window of ONE seeded block of uniform numbers -- a different window
per parameter, scaled by its own bound: memory-copy speed.  Explicit initialisations that follow construction (LoRALinearLayer:
normal / zeros, cldm/lora.py:67-68; zero_module) overwrite the mark and are therefore left exactly as the model code made them."""
import zlib

import torch
import torch.nn as nn


class skip_default_init:
    def __enter__(self):
        self.saved = (nn.Linear.reset_parameters, nn.modules.conv._ConvNd.reset_parameters)
        def mark(m):
            with torch.no_grad():
                m.weight.fill_(float("nan"))
                if m.bias is not None:
                    m.bias.fill_(float("nan"))
        nn.Linear.reset_parameters = mark
        nn.modules.conv._ConvNd.reset_parameters = mark
        return self

    def __exit__(self, *a):
        nn.Linear.reset_parameters, nn.modules.conv._ConvNd.reset_parameters = self.saved


def fill_default_init(model: nn.Module, seed: int = 0):
    block = torch.rand(1 << 26, generator=torch.Generator().manual_seed(seed + 12345)).mul_(2).sub_(1)
    with torch.no_grad():
        for name, m in model.named_modules():
            if not isinstance(m, (nn.Linear, nn.modules.conv._ConvNd)):
                continue
            bound = float(m.weight[0].numel()) ** -0.5
            for pn in ("weight", "bias"):
                p = getattr(m, pn, None)
                if p is None or p.numel() == 0 or not bool(torch.isnan(p.reshape(-1)[0])):
                    continue          # explicitly initialised after construction (or drawn by a constructor that was not patched)
                n = p.numel()
                flat = p.reshape(-1)
                if n < block.numel():
                    off = zlib.crc32(f"{name}.{pn}".encode()) % (block.numel() - n)
                    flat.copy_(block[off:off + n])
                else:                 # larger than the block: tile it
                    for lo in range(0, n, block.numel()):
                        hi = min(n, lo + block.numel())
                        flat[lo:hi].copy_(block[:hi - lo])
                flat.mul_(bound)
        # Anything that READ a marked parameter while the model was being constructed (weight_norm's g, an EMA clone, one layer's
        # weights copied into another) has baked the mark in and is not a Linear / Conv weight or bias of its own: refuse loudly
        # instead of training on NaNs.  One isnan().any() per tensor.
        bad = [n_ for n_, t_ in list(model.named_parameters()) + list(model.named_buffers())
               if t_.is_floating_point() and t_.numel() and bool(torch.isnan(t_).any())]
        if bad:
            raise RuntimeError("fastinit: tensors still carry the 'not drawn yet' mark after fill_default_init (something read a "
                               f"parameter during construction): {bad[:5]}{' ...' if len(bad) > 5 else ''}; set CTRLORA_FAST_INIT=0")
