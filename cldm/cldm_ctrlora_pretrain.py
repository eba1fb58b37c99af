"""Multi-task Base-ControlNet pre-training classes (API of cldm/cldm_ctrlora_pretrain.py).

Module tree, `loras_dict` banks, `switch_lora(task)` and the forward / sampling path are provided.
Pre-TRAINING optimises every ControlNet weight (:174-182), i.e. needs weight gradients for all convs
and linears; that is SURVEY.md 8(f3) "next" work and raises NotImplementedError for now.
"""
import torch
import torch.nn as nn

from cldm.cldm import ControlLDM, ControlNet
from cldm.cldm_ctrlora_finetune import swap_linears
from cldm.ddim_hacked import DDIMSampler
from cldm.lora import LoRACompatibleLinear, LoRALinearLayer


class ControlNetPretrain(ControlNet):
    def __init__(self, lora_rank, tasks, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.lora_rank, self.tasks, self.n_tasks = lora_rank, list(tasks), len(tasks)
        del self.input_hint_block
        linears = [m for _, m in self.named_modules() if isinstance(m, nn.Linear)]
        self.loras_dict = nn.ModuleDict({
            task: nn.ModuleList([LoRALinearLayer(m.in_features, m.out_features, rank=lora_rank) for m in linears])
            for task in self.tasks})
        self._lora_linears = swap_linears(self, lambda m: None, skip=("loras_dict",))
        self._task = None

    def switch_lora(self, task: str):
        assert task in self.tasks
        for lin, lora in zip(self._lora_linears, self.loras_dict[task]):
            lin.set_lora_layer(lora)
        if task != self._task:
            self._task = task
            self.invalidate_engine()

    def _executor_state(self):
        return {k: v for k, v in self.state_dict().items() if not k.startswith("loras_dict.")}

    def _on_state_loaded(self):
        self.invalidate_engine()

    def forward(self, hint, timesteps, context, **kwargs):
        return self._latent_forward(hint, timesteps, context)


class ControlPretrainLDM(ControlLDM):

    @torch.no_grad()
    def sample_log(self, cond, batch_size, ddim, ddim_steps, **kwargs):
        b, c, h, w = cond["c_concat"][0].shape
        shape = (self.channels, h // 8, w // 8) if c != self.channels else (self.channels, h, w)
        return DDIMSampler(self).sample(ddim_steps, batch_size, shape, cond, verbose=False, **kwargs)

    @torch.no_grad()
    def get_input(self, batch, k, bs=None, *args, **kwargs):
        x, c_dict = super().get_input(batch, k, bs, *args, **kwargs)
        c_dict.update({"task": batch["task"][0][8:]})          # strips 'control_' (:91)
        return x, c_dict

    def apply_model(self, x_noisy, t, cond, *args, **kwargs):
        assert isinstance(cond, dict)
        cond_txt = torch.cat(cond["c_crossattn"], 1)
        if cond["c_concat"] is None:
            return self._run(x_noisy, t, cond_txt, None)
        self.control_model.switch_lora(cond["task"])
        self.__dict__.pop("_engine", None)
        if torch.is_grad_enabled() and self.training:
            raise NotImplementedError("Base-ControlNet pre-training (all ControlNet weights trainable) is not built "
                                      "yet; LoRA fine-tuning and sampling are (see DESIGN.md, scope)")
        return self._run(x_noisy, t, cond_txt, [self._hint_latent(cond)])

    def configure_optimizers(self):
        raise NotImplementedError("Base-ControlNet pre-training is SURVEY.md 8(f3) 'next' work")
