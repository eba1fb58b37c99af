"""Multi-LoRA inference classes (API of cldm/cldm_ctrlora_inference.py): `lora_num` banks of
{LoRA, zero convs, norm layers}; apply_model sums the banks' residuals with `lora_weights`."""
import copy

import torch
import torch.nn as nn

from cldm.cldm import ControlLDM, ControlNet
from cldm.cldm_ctrlora_finetune import swap_linears
from cldm.ddim_hacked import DDIMSampler
from cldm.lora import LoRALinearLayer
from cldm.switchable import SwitchableConv2d, SwitchableGroupNorm, SwitchableLayerNorm


def _replace(root, name, new):
    *path, leaf = name.split(".")
    parent = root
    for p in path:
        parent = parent.get_submodule(p)
    parent._modules[leaf] = new


class ControlNetInference(ControlNet):
    def __init__(self, lora_rank=128, lora_num=1, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.lora_rank, self.lora_num = lora_rank, lora_num
        del self.input_hint_block
        named = list(self.named_modules())
        linears = [(n, m) for n, m in named if isinstance(m, nn.Linear)]
        zeros = [(n, m) for n, m in named if ("zero_convs" in n or "middle_block_out" in n) and isinstance(m, nn.Conv2d)]
        norms = [(n, m) for n, m in named if "norm" in n and isinstance(m, (nn.GroupNorm, nn.LayerNorm))]
        self._linear_names = [n for n, _ in linears]
        self._zero_names = [n for n, _ in zeros]
        self._norm_names = [n for n, _ in norms]
        self.loras_list = nn.ModuleList([nn.ModuleList(
            [LoRALinearLayer(m.in_features, m.out_features, rank=lora_rank) for _, m in linears]) for _ in range(lora_num)])
        self.zero_convs_list = nn.ModuleList([nn.ModuleList([copy.deepcopy(m) for _, m in zeros]) for _ in range(lora_num)])
        self.norms_list = nn.ModuleList([nn.ModuleList([copy.deepcopy(m) for _, m in norms]) for _ in range(lora_num)])
        swap_linears(self, lambda m: None, skip=("loras_list", "zero_convs_list", "norms_list"))
        for n, m in zeros:
            _replace(self, n, SwitchableConv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding,
                                               m.dilation, m.groups, m.bias is not None))
        for n, m in norms:
            _replace(self, n, SwitchableGroupNorm(m.num_groups, m.num_channels) if isinstance(m, nn.GroupNorm)
                     else SwitchableLayerNorm(m.normalized_shape, m.eps, m.elementwise_affine))
        self._bank_exec = {}
        self._active_bank = None     # index of the last switch_lora() (None: the holders' own weights, no LoRA)

    def switch_lora(self, index: int):
        self._active_bank = index
        for n, lora in zip(self._linear_names, self.loras_list[index]):
            self.get_submodule(n).set_lora_layer(lora)
        for n, z in zip(self._zero_names, self.zero_convs_list[index]):
            self.get_submodule(n).set_conv_layer(z)
        for n, nm in zip(self._norm_names, self.norms_list[index]):
            self.get_submodule(n).set_norm_layer(nm)

    def copy_weights_to_switchable(self):
        """After switch_lora(i) + load_state_dict(): move the freshly loaded zero-conv / norm weights into
        bank i (:132-139).  Invalidates that bank's packed engine copy."""
        for _, m in self.named_modules():
            if isinstance(m, (SwitchableConv2d, SwitchableGroupNorm, SwitchableLayerNorm)):
                m.copy_weights()
        self._bank_exec.clear()

    def invalidate_engine(self):
        self._bank_exec.clear()

    def _on_state_loaded(self):
        self._bank_exec.clear()      # inference banks carry no optimizer state: rebuilt lazily from bank_state()

    def bank_state(self, index: int):
        """State dict of bank `index` under ControlNetFinetune key names."""
        skip = ("loras_list.", "zero_convs_list.", "norms_list.", ".lora_layer.", ".conv_layer.", ".norm_layer.")
        sd = {k: v for k, v in self.state_dict().items() if not any(s in k for s in skip)}
        for n, lora in zip(self._linear_names, self.loras_list[index]):
            sd[f"{n}.lora_layer.down.weight"] = lora.down.weight.data
            sd[f"{n}.lora_layer.up.weight"] = lora.up.weight.data
        for n, z in zip(self._zero_names, self.zero_convs_list[index]):
            sd[f"{n}.weight"], sd[f"{n}.bias"] = z.weight.data, z.bias.data
        for n, nm in zip(self._norm_names, self.norms_list[index]):
            sd[f"{n}.weight"], sd[f"{n}.bias"] = nm.weight.data, nm.bias.data
        return sd

    def bank_executor(self, index: int):
        ex = self._bank_exec.get(index)
        if ex is None:
            from ctrlora_amd.engine import ControlNetE
            ex = ControlNetE(self.bank_state(index), self.net_cfg(), self._engine_dtype(), self._device(), need_bwd=False)
            self._bank_exec[index] = ex
        return ex

    def _unswitched_state(self):
        """Before any switch_lora(): the reference's modules then run on their OWN zero-conv / norm weights with no LoRA
        attached (cldm/switchable.py:17-20,37-40,58-61; cldm/lora.py:286-287)."""
        skip = ("loras_list.", "zero_convs_list.", "norms_list.", ".lora_layer.", ".conv_layer.", ".norm_layer.")
        return {k: v for k, v in self.state_dict().items() if not any(s in k for s in skip)}

    def forward(self, hint, timesteps, context, **kwargs):
        """:100-114 -- the 13 zero-conv outputs of the ACTIVE bank (the one of the last switch_lora(i)) for a 4-channel
        latent hint, NCHW fp32.  Runs that bank's executor, as ControlInferenceLDM.apply_model does per bank."""
        from ctrlora_amd.engine import ControlNetE, CtrLoRAEngine
        if self._active_bank is None:
            ex = self._bank_exec.get(None)
            if ex is None:
                ex = ControlNetE(self._unswitched_state(), self.net_cfg(), self._engine_dtype(), self._device(), need_bwd=False)
                self._bank_exec[None] = ex
        else:
            ex = self.bank_executor(self._active_bank)
        eng = CtrLoRAEngine.__new__(CtrLoRAEngine)
        eng.cfg, eng.dtype, eng.device, eng.unet, eng.controls = ex.cfg, ex.dtype, ex.device, None, [ex]
        return eng.control_outputs(hint, timesteps, context, 0)


class ControlInferenceLDM(ControlLDM):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.lora_weights = [1.0 / self.control_model.lora_num] * self.control_model.lora_num

    @torch.no_grad()
    def sample_log(self, cond, batch_size, ddim, ddim_steps, **kwargs):
        b, c, h, w = cond["c_concat"][0].shape
        shape = (self.channels, h // 8, w // 8) if c != self.channels else (self.channels, h, w)
        return DDIMSampler(self).sample(ddim_steps, batch_size, shape, cond, verbose=False, **kwargs)

    def _control_executors(self):
        return [self.control_model.bank_executor(i) for i in range(self.control_model.lora_num)]

    def _executor_owners(self):
        return []


    @torch.no_grad()
    def apply_model(self, x_noisy, t, conds, *args, **kwargs):
        """:156-178 -- one ControlNet pass per LoRA bank, weighted sum of the residual lists, UNet."""
        if isinstance(conds, dict):
            conds = [conds]
        assert isinstance(conds, (list, tuple))
        assert len(conds) == self.control_model.lora_num
        assert len(self.lora_weights) == self.control_model.lora_num
        cond_txt = torch.cat(conds[0]["c_crossattn"], 1)
        hints = [self._hint_latent(c) for c in conds]
        return self._run(x_noisy, t, cond_txt, hints, weights=list(self.lora_weights))
