"""Switchable zero-conv / norm holders used by multi-LoRA inference (API of cldm/switchable.py).
`set_*_layer` re-points the active bank; `copy_weights` stores freshly loaded weights into it."""
import torch.nn as nn


class _Switchable:
    _slot = None

    def _active(self):
        return getattr(self, self._slot)

    def copy_weights(self):
        tgt = self._active()
        if tgt is not None:
            tgt.weight.data.copy_(self.weight.data)
            if getattr(self, "bias", None) is not None:
                tgt.bias.data.copy_(self.bias.data)

    def forward(self, x):
        """Stand-alone use (cldm/switchable.py:17-20,37-40,58-61): the active bank's layer when one is set, else this
        holder's own weights -- plain torch modules outside the engine (the enclosing ControlNetInference executes on
        the HIP executors and never calls this)."""
        tgt = self._active()
        if tgt is not None:
            return tgt(x)
        return super().forward(x)


class SwitchableGroupNorm(_Switchable, nn.GroupNorm):
    _slot = "norm_layer"

    def __init__(self, *args, norm_layer=None, **kwargs):
        nn.GroupNorm.__init__(self, *args, **kwargs)
        self.norm_layer = norm_layer

    def set_norm_layer(self, norm_layer):
        self.norm_layer = norm_layer


class SwitchableLayerNorm(_Switchable, nn.LayerNorm):
    _slot = "norm_layer"

    def __init__(self, *args, norm_layer=None, **kwargs):
        nn.LayerNorm.__init__(self, *args, **kwargs)
        self.norm_layer = norm_layer

    def set_norm_layer(self, norm_layer):
        self.norm_layer = norm_layer


class SwitchableConv2d(_Switchable, nn.Conv2d):
    _slot = "conv_layer"

    def __init__(self, *args, conv_layer=None, **kwargs):
        nn.Conv2d.__init__(self, *args, **kwargs)
        self.conv_layer = conv_layer

    def set_conv_layer(self, conv_layer):
        self.conv_layer = conv_layer
