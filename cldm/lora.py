"""LoRA injection points (API of the reference's cldm/lora.py: LoRALinearLayer :26-80,
LoRACompatibleLinear :225-291).  Inside a ControlNet the rank-r branch is folded into the main
MFMA GEMM by the engine; used stand-alone on a GPU tensor these modules call the same HIP kernel."""
from typing import Optional

import torch
from torch import nn


class LoRALinearLayer(nn.Module):
    def __init__(self, in_features, out_features, rank=4, network_alpha=None, device=None, dtype=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False, device=device, dtype=dtype)
        self.up = nn.Linear(rank, out_features, bias=False, device=device, dtype=dtype)
        self.network_alpha = network_alpha
        self.rank, self.out_features, self.in_features = rank, out_features, in_features
        nn.init.normal_(self.down.weight, std=1 / rank)      # lora.py:67-68
        nn.init.zeros_(self.up.weight)

    def forward(self, hidden_states):
        from ctrlora_amd.standalone import lora_delta
        return lora_delta(hidden_states, self.down.weight, self.up.weight,
                          None if self.network_alpha is None else self.network_alpha / self.rank)


class LoRACompatibleLinear(nn.Linear):
    """y = x W^T + b + scale * up(down(x))    (lora.py:285-291)"""

    def __init__(self, *args, lora_layer: Optional[LoRALinearLayer] = None, **kwargs):
        super().__init__(*args, **kwargs)
        self.lora_layer = lora_layer

    def set_lora_layer(self, lora_layer: Optional[LoRALinearLayer]):
        self.lora_layer = lora_layer

    def _fuse_lora(self, lora_scale: float = 1.0, safe_fusing: bool = False):
        # W += scale * up @ down in fp32 (lora.py:237-267)
        if self.lora_layer is None:
            return
        dtype, device = self.weight.data.dtype, self.weight.data.device
        w_up = self.lora_layer.up.weight.data.float()
        w_down = self.lora_layer.down.weight.data.float()
        if self.lora_layer.network_alpha is not None:
            w_up = w_up * self.lora_layer.network_alpha / self.lora_layer.rank
        fused = self.weight.data.float() + lora_scale * (w_up @ w_down)
        if safe_fusing and torch.isnan(fused).any().item():
            raise ValueError(f"This LoRA weight seems to be broken. Encountered NaN values when trying to fuse "
                             f"LoRA weights for {self}. LoRA weights will not be fused.")
        self.weight.data = fused.to(device=device, dtype=dtype)
        self.lora_layer = None
        self.w_up, self.w_down, self._lora_scale = w_up.cpu(), w_down.cpu(), lora_scale

    def _unfuse_lora(self):
        if getattr(self, "w_up", None) is None or getattr(self, "w_down", None) is None:
            return
        fused = self.weight.data
        w_up = self.w_up.to(fused.device).float()
        w_down = self.w_down.to(fused.device).float()
        self.weight.data = (fused.float() - self._lora_scale * (w_up @ w_down)).to(fused.dtype)
        self.w_up = self.w_down = None

    def forward(self, hidden_states, scale: float = 1.0):
        from ctrlora_amd.standalone import lora_linear
        lora = self.lora_layer
        return lora_linear(hidden_states, self.weight, self.bias,
                           None if lora is None else lora.down.weight, None if lora is None else lora.up.weight,
                           scale if lora is None or lora.network_alpha is None
                           else scale * lora.network_alpha / lora.rank)
