"""Multi-task Base-ControlNet pre-training classes (API of cldm/cldm_ctrlora_pretrain.py).

Module tree, `loras_dict` banks, `switch_lora(task)`, the forward / sampling path and the TRAINING path: pre-training
optimises every ControlNet weight (:174-182), so the engine forms the weight gradient of every conv (one tap at a time,
gathered inside the weight-gradient kernel's addressing), linear, bias and norm next to the task's LoRA factors; the
base weights live in one flat fp32 buffer, each task's LoRA bank in its own (ctrlora_amd.train.PretrainAdamW).
"""
import os

import torch
import torch.nn as nn

from cldm.cldm import ControlLDM, ControlNet
from cldm.cldm_ctrlora_finetune import swap_linears
from cldm.ddim_hacked import DDIMSampler
from cldm.lora import LoRALinearLayer


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
        ids = {id(m): n for n, m in self.named_modules()}
        self._lora_names = [ids[id(m)] for m in self._lora_linears]
        self._task = None

    def switch_lora(self, task: str):
        """:68-76 -- point every LoRACompatibleLinear at the task's bank.  The executor is kept (an optimizer may hold
        its flat buffers): the engine swaps the active LoRA bank and re-packs only the LoRA factors."""
        assert task in self.tasks
        for lin, lora in zip(self._lora_linears, self.loras_dict[task]):
            lin.set_lora_layer(lora)
        if task != self._task:
            self._task = task
            ex = self.__dict__.get("_exec")
            if ex is not None:
                ex.switch_bank(self.bank(task))

    def _executor_state(self):
        return {k: v for k, v in self.state_dict().items() if not k.startswith("loras_dict.")}

    def _on_state_loaded(self):
        ex = self.__dict__.get("_exec")
        if ex is not None:       # every tensor of the executor is a bound trainable: the load wrote through to the masters
            ex.repack()

    # ---- engine: base weights trainable (one flat buffer), one flat LoRA bank per task
    def executor(self):
        ex = self.__dict__.get("_exec")
        if ex is None:
            from ctrlora_amd.engine import ControlNetE
            from ctrlora_amd.train import bind_trainables
            if self._task is None:
                self.switch_lora(self.tasks[0])
            ex = ControlNetE(self._executor_state(), self.net_cfg(), self._engine_dtype(), self._device(), train_all=True)
            self.__dict__["_exec"] = ex
            self.__dict__["_banks"] = {self._task: ex.tr_lora}
            self.__dict__["_bound"] = bind_trainables(self, ex)
            self._bind_bank(self._task)
        return ex

    def _bank_params(self, task):
        out = {}
        for name, lora in zip(self._lora_names, self.loras_dict[task]):
            out[f"{name}.lora_layer.down.weight"] = lora.down.weight
            out[f"{name}.lora_layer.up.weight"] = lora.up.weight
        return out

    def _bind_bank(self, task):
        from ctrlora_amd.train import bind_bank
        bind_bank(self._bank_params(task), self.__dict__["_banks"][task])

    def bank(self, task):
        """The task's LoRA bank as a flat trainable set (created from the module's loras_dict[task] on first use)."""
        self.executor()
        banks = self.__dict__["_banks"]
        ts = banks.get(task)
        if ts is None:
            from ctrlora_amd.engine.packing import TrainableSet
            proto = next(iter(banks.values()))
            ts = TrainableSet()
            for t in proto.items:
                ts.declare(t.name, t.shape)
            ts.materialize({k: v.detach() for k, v in self._bank_params(task).items()}, self._device())
            banks[task] = ts
            self._bind_bank(task)
        return ts

    def invalidate_engine(self):
        if self.__dict__.get("_exec") is not None:
            # the flat buffers back the nn.Parameters: give them their own storage again before dropping the executor
            with torch.no_grad():
                for p in self.parameters():
                    p.data = p.data.clone()
                    p.grad = None
        for k in ("_exec", "_banks", "_bound", "_bound_version"):
            self.__dict__.pop(k, None)

    def forward(self, hint, timesteps, context, **kwargs):
        return self._latent_forward(hint, timesteps, context)


class _PretrainDP:
    """Gradient exchange of multi-task pre-training (ctrlora_amd.parallel.BankedGradAllReduce) behind the hooks the
    training glue calls: the shared buffer and the banks live anywhere this optimizer step are summed over ranks.

    Replica consistency (what DDP gives the reference for free): every rank must apply the SAME update to the SAME banks
    in the SAME step.  So (a) the set handed to the exchange is every task this rank trained since its last optimizer
    step (gradient accumulation may visit several), not only the current one; (b) right after the exchange -- before
    optimizer.step() -- every bank that was live on ANY rank is marked used on THIS rank's optimizer, which makes it part
    of this step's update (with the summed gradient) and of the following zero_grad, and starts its Adam step counter on
    every rank at once."""

    def __init__(self, cm, opt=None):
        from ctrlora_amd.parallel import BankedGradAllReduce
        ex = cm.executor()
        self.cm = cm
        self.opt = opt                   # PretrainAdamW (configure_optimizers sets it): mark_used(task) on exchange
        self.inner = BankedGradAllReduce([ex.tr.flat_grad], {t: cm.bank(t).flat_grad for t in cm.tasks})
        if os.environ.get("CTRLORA_PRETRAIN_OVERLAP", "1") != "0":
            self.inner.attach(ex)        # buckets of the base-ControlNet gradients leave from the backward's stage hooks
        self.world_size = self.inner.world_size
        self._enabled = True             # False on non-final gradient-accumulation micro-steps
        self._mask_sent = False          # this optimizer step's used-bank mask is already on the wire
        self.live = []                   # banks exchanged by the last optimizer step (on every rank)
        self.used = []                   # tasks this rank back-propagated through since its last optimizer step

    def _get_enabled(self):
        return self._enabled

    def _set_enabled(self, v):
        self._enabled = bool(v)
        self.inner.enabled = bool(v)     # (the stage hook must not launch collectives on accumulation micro-steps)

    enabled = property(_get_enabled, _set_enabled)

    def note_used(self, task):
        if task not in self.used:
            self.used.append(task)
        if self.enabled and self.inner.world_size > 1:
            # final micro-step of the optimizer step, called at the START of its forward (apply_model /
            # engine_train_step): the used-bank mask travels while the step computes (BankedGradAllReduce.prefetch_mask).
            # Idempotent for an unchanged set; a set that changed since an un-consumed prefetch (a forward without its backward
            # pass) supersedes it there -- on every rank alike
            self.inner.prefetch_mask(self.used)
            self._mask_sent = True

    def on_backward_done(self):
        self.note_used(self.cm._task)
        if not self.enabled:
            return
        self.live = self.inner.exchange(self.used)
        self.used = []
        self._mask_sent = False
        if self.opt is not None:
            for t in self.live:
                self.opt.mark_used(t)

    def wait(self):
        pass


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
        """:95-111 -- the task named in the batch selects the LoRA bank; training back-propagates into EVERY ControlNet
        parameter (base weights + that bank)."""
        assert isinstance(cond, dict)
        cc = cond["c_crossattn"]
        cond_txt = cc[0] if len(cc) == 1 else torch.cat(cc, 1)
        if cond["c_concat"] is None:
            return self._run(x_noisy, t, cond_txt, None)
        self.control_model.switch_lora(cond["task"])
        if torch.is_grad_enabled() and self.training:
            opt = self.__dict__.get("_opt")
            if opt is not None:
                opt.mark_used(cond["task"])
            if isinstance(self.dp, _PretrainDP):     # banks other ranks train join the update inside on_backward_done
                self.dp.note_used(cond["task"])
        return self._run(x_noisy, t, cond_txt, [self._hint_latent(cond)])

    @torch.no_grad()
    def engine_train_step(self, x_start, cond, t, noise):
        """p_losses + backward without torch.autograd (ControlFinetuneLDM.engine_train_step), after selecting the task's
        LoRA bank: capturable as a hipGraph (ctrlora_amd.train.GraphedPretrainStep keeps one graph per task)."""
        from cldm.cldm_ctrlora_finetune import ControlFinetuneLDM
        task = cond["task"]
        self.control_model.switch_lora(task)
        opt = self.__dict__.get("_opt")
        if opt is not None:
            opt.mark_used(task)
        if isinstance(self.dp, _PretrainDP):
            self.dp.note_used(task)
        return ControlFinetuneLDM.engine_train_step(self, x_start, cond, t, noise)

    def init_data_parallel(self):
        """Install the bank-sparse gradient exchange (call after torch.distributed is initialised)."""
        self.dp = _PretrainDP(self.control_model)
        return self.dp

    def configure_optimizers(self):
        """:174-182 -- AdamW over list(control_model.parameters()) (base ControlNet + every task's LoRA bank)."""
        import os
        from ctrlora_amd.train import PretrainAdamW
        cm = self.control_model
        ex = cm.executor()
        banks = {t: cm.bank(t) for t in cm.tasks}
        params = list(cm.parameters())
        print(f"Optimizable params: {sum(p.numel() for p in params) / 1e6:.1f}M")
        os.makedirs("./tmp", exist_ok=True)
        with open("./tmp/pretrain_trainable_params.txt", "w") as f:
            for n, _ in cm.named_parameters():
                f.write(n + "\n")
        world = 1 if self.dp is None else self.dp.world_size
        opt = PretrainAdamW(params, ex, banks, lr=self.learning_rate, grad_scale=1.0 / world)
        self.__dict__["_opt"] = opt
        if isinstance(self.dp, _PretrainDP):
            self.dp.opt = opt
        return opt
