"""CtrLoRA fine-tuning classes (API of the reference's cldm/cldm_ctrlora_finetune.py)."""
import os

import torch
import torch.nn as nn

from cldm.cldm import ControlLDM, ControlNet
from cldm.ddim_hacked import DDIMSampler
from cldm.lora import LoRACompatibleLinear, LoRALinearLayer


def swap_linears(root: nn.Module, make_lora, skip=()):
    """Replace every nn.Linear below `root` by a LoRACompatibleLinear carrying the same weights
    (cldm_ctrlora_finetune.py:21-38).  Returns the new modules in named_modules order."""
    out = []
    for name, m in list(root.named_modules()):
        if not isinstance(m, nn.Linear) or isinstance(m, LoRACompatibleLinear) or any(s in name for s in skip):
            continue
        new = LoRACompatibleLinear(m.in_features, m.out_features, lora_layer=make_lora(m))
        new.weight.data.copy_(m.weight.data)
        if m.bias is not None:
            new.bias.data.copy_(m.bias.data)
        else:
            new.bias = None
        *path, leaf = name.split(".")
        parent = root
        for p in path:
            parent = parent.get_submodule(p)
        parent._modules[leaf] = new
        out.append(new)
    return out


class ControlNetFinetune(ControlNet):
    def __init__(self, ft_with_lora=True, lora_rank=128, norm_trainable=True, zero_trainable=True, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.ft_with_lora, self.lora_rank = ft_with_lora, lora_rank
        self.norm_trainable, self.zero_trainable = norm_trainable, zero_trainable
        del self.input_hint_block                                     # :19
        if ft_with_lora:
            swap_linears(self, lambda m: LoRALinearLayer(m.in_features, m.out_features, rank=lora_rank))
        # ft_with_lora=False (configs/ctrlora_finetune_sd15_full.yaml): plain linears, EVERY ControlNet parameter trains
        # (:100-103) -- the engine then forms all weight gradients, as in pre-training
        self.train_all_weights = not ft_with_lora

    def forward(self, hint, timesteps, context, **kwargs):
        """13 zero-conv outputs for a 4-channel latent hint (:40-54)."""
        return self._latent_forward(hint, timesteps, context)


class ControlFinetuneLDM(ControlLDM):

    @torch.no_grad()
    def sample_log(self, cond, batch_size, ddim, ddim_steps, **kwargs):
        b, c, h, w = cond["c_concat"][0].shape
        shape = (self.channels, h // 8, w // 8) if c != self.channels else (self.channels, h, w)
        return DDIMSampler(self).sample(ddim_steps, batch_size, shape, cond, verbose=False, **kwargs)

    def apply_model(self, x_noisy, t, cond, *args, **kwargs):
        """:67-82 -- ControlNet on the hint latent, residuals * control_scales, frozen UNet."""
        assert isinstance(cond, dict)
        cond_txt = torch.cat(cond["c_crossattn"], 1)
        if cond["c_concat"] is None:
            return self._run(x_noisy, t, cond_txt, None)
        return self._run(x_noisy, t, cond_txt, [self._hint_latent(cond)])

    @torch.no_grad()
    def engine_train_step(self, x_start, cond, t, noise):
        """p_losses (ddpm.py:885-920) + backward of the loss WITHOUT torch.autograd: q_sample, apply_model with
        recording, the p_losses reduction with d loss / d eps, the hand-written backward -- every launch is one of
        this library's kernels, so the whole thing is capturable as a hipGraph with no ATen nodes
        (ctrlora_amd.train.GraphedTrainStep).  Returns the device 3-vector {loss_simple, loss_vlb, loss}; gradients
        land in the flat buffer the optimizer's parameters view."""
        from ctrlora_amd import hip
        if self.loss_type != "l2" or self.original_elbo_weight != 0.:
            raise NotImplementedError("engine_train_step: l2 loss without the elbo term (every CtrLoRA config)")
        x_noisy = self.q_sample(x_start=x_start, t=t, noise=noise)
        cc = cond["c_crossattn"]
        cond_txt = cc[0] if len(cc) == 1 else torch.cat(cc, 1)
        hints = [self._hint_latent(cond)]
        self._sync_trainables()
        eng = self.engine()
        eps = eng.forward(x_noisy, t, cond_txt, hints, control_scales=list(self.control_scales), record=True,
                          only_mid_control=self.only_mid_control)
        out = torch.empty(3, dtype=torch.float32, device=eps.device)
        scratch = torch.empty(16 * eps.shape[0], dtype=torch.float32, device=eps.device)
        d_eps = torch.empty_like(eps)
        hip.p_losses_mse(eps, noise.float().contiguous(), d_eps, t.long().contiguous(), self.lvlb_weights, out, scratch,
                         1.0, float(self.l_simple_weight), 0.0)
        eng.backward(d_eps)
        if self.dp is not None:
            self.dp.on_backward_done()
        return out

    def trainable_names(self):
        """Name filter of :84-108 (LoRA layers; zero convs incl. middle_block_out; `norm` layers)."""
        cm = self.control_model
        names = []
        for n, _ in cm.named_parameters():
            assert "input_hint" not in n
            if not cm.ft_with_lora:
                assert "lora_layer" not in n
                names.append(n)
            elif "lora_layer" in n:
                names.append(n)
            elif ("zero_convs" in n or "middle_block_out" in n) and cm.zero_trainable:
                names.append(n)
            elif "norm" in n and cm.norm_trainable:
                names.append(n)
        return names

    def configure_optimizers(self):
        from ctrlora_amd.train import FusedAdamW
        cm = self.control_model
        if cm.ft_with_lora and not (cm.zero_trainable and cm.norm_trainable):
            raise NotImplementedError("the engine trains LoRA + zero convs + norm layers together "
                                      "(every shipped config sets both flags)")
        names = self.trainable_names()
        os.makedirs("./tmp", exist_ok=True)
        with open("./tmp/finetune_trainable_params.txt", "w") as f:
            f.write("\n".join(names) + "\n")
        ex = cm.executor()
        bound = dict(zip([t.name for t in ex.tr.items], cm.__dict__["_bound"]))
        assert set(names) == set(bound), "engine trainable set differs from the reference's name filter"
        params = [bound[n] for n in names]
        print(f"Optimizable params: {sum(p.numel() for p in params) / 1e6:.1f}M")
        world = 1 if self.dp is None else self.dp.world_size
        opt = FusedAdamW(params, [ex], lr=self.learning_rate, grad_scale=1.0 / world)
        if self.dp is not None:
            opt.pre_step_hook = self.dp.wait
        return opt
