"""A Lightning-free training loop with the pieces of `pl.Trainer` the reference's train scripts use
(scripts/train_ctrlora_finetune.py:122-129: strategy='ddp', accumulate_grad_batches, max_steps, precision,
callbacks, default_root_dir).

One process per GPU: launch with `torchrun --nproc-per-node N script.py ...` (RANK / LOCAL_RANK / WORLD_SIZE from
the environment, backend "nccl" = RCCL); the only collective is the all-reduce of the flat LoRA gradient buffer
(`ctrlora_amd.parallel.GradAllReduce`), suppressed on non-final gradient-accumulation micro-steps.  The model is
any module with the LightningModule-style hooks the reference's LDM classes expose: `training_step(batch,
batch_idx) -> loss`, `configure_optimizers()`, optionally `set_engine_dtype`, `dp`, `log_images`.

Semantics kept from Lightning 1.5: the loss of each micro-batch is divided by `accumulate_grad_batches`;
`global_step` counts OPTIMIZER steps; callbacks get `on_train_batch_end(trainer, module, outputs, batch, batch_idx)`
after every micro-batch and `on_batch_end(trainer, module)` (what `CheckpointEveryNSteps` hooks); checkpoints are
`{"state_dict", "global_step", "epoch", "optimizer_states"}` and resumable with `fit(..., ckpt_path=)`.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch


class _CheckpointDir:
    """`trainer.checkpoint_callback.dirpath` / `.filename` as the reference's callbacks read them."""

    def __init__(self, dirpath):
        self.dirpath, self.filename = dirpath, "last.ckpt"


class Trainer:
    def __init__(self, max_steps: int = 100000, accumulate_grad_batches: int = 1, precision=32,
                 callbacks: Sequence = (), default_root_dir: str = "runs/default", device: Optional[str] = None,
                 strategy: str = "ddp", accelerator: str = "gpu", devices=-1, log_every_n_steps: int = 50):
        self.max_steps = int(max_steps)
        self.accumulate_grad_batches = max(1, int(accumulate_grad_batches))
        self.precision = precision
        self.callbacks = list(callbacks)
        self.default_root_dir = default_root_dir
        self.log_every_n_steps = log_every_n_steps
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.global_rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = torch.device(device) if device is not None else torch.device("cuda", self.local_rank)
        self.log_dir = os.path.join(default_root_dir, "lightning_logs", "version_0")
        self.checkpoint_callback = _CheckpointDir(os.path.join(self.log_dir, "checkpoints"))
        self.global_step = 0
        self.current_epoch = 0
        self.model = None
        self.optimizer = None
        self.logged = []

    # ------------------------------------------------------------------ helpers
    @property
    def is_global_zero(self):
        return self.global_rank == 0

    def engine_dtype(self):
        return torch.float32 if str(self.precision) in ("32", "32-true") else torch.bfloat16

    def _call(self, hook, *args):
        for cb in self.callbacks:
            fn = getattr(cb, hook, None)
            if callable(fn):
                fn(self, *args)

    def _init_distributed(self, model):
        if self.world_size <= 1:
            return
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = "nccl" if self.device.type == "cuda" else "gloo"
            dist.init_process_group(backend, **({"device_id": self.device} if backend == "nccl" else {}))
        # what Lightning's DDP wrapper does at wrap time: every rank starts from rank 0's parameters and buffers
        # (the LoRA down-projections are drawn from an unseeded N(0, 1/r) in every process)
        self.broadcast_module_state(model)
        if callable(getattr(model, "init_data_parallel", None)) and self.device.type == "cuda":
            model.init_data_parallel()           # pre-training: bank-sparse exchange (base buffer + live LoRA banks)
        elif hasattr(model, "control_model") and hasattr(model.control_model, "executor") and self.device.type == "cuda":
            from ctrlora_amd.parallel import GradAllReduce
            model.dp = GradAllReduce([model.control_model.executor()])

    @staticmethod
    def broadcast_module_state(model, src: int = 0):
        """In-place broadcast of all parameters and buffers from rank `src` (no-op without a process group)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                if t.numel():
                    dist.broadcast(t.data, src)

    def save_checkpoint(self, path):
        if not self.is_global_zero:
            return
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        opt_state = self.optimizer.state_dict() if hasattr(self.optimizer, "state_dict") else None
        torch.save({"state_dict": self.model.state_dict(), "global_step": self.global_step,
                    "epoch": self.current_epoch, "optimizer_states": [opt_state]}, path)

    def _restore_model(self, ckpt_path):
        """Model part of a resume: BEFORE the executors / optimizer are built (they snapshot the weights)."""
        ck = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ck["state_dict"], strict=True)
        self.global_step, self.current_epoch = int(ck.get("global_step", 0)), int(ck.get("epoch", 0))
        return ck

    def _restore_optimizer(self, ck):
        st = (ck.get("optimizer_states") or [None])[0]
        if st is not None and hasattr(self.optimizer, "load_state_dict"):
            self.optimizer.load_state_dict(st)

    # ------------------------------------------------------------------ the loop
    def fit(self, model, train_dataloader, ckpt_path: Optional[str] = None):
        self.model = model
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        model.to(self.device).train()
        if hasattr(model, "set_engine_dtype"):
            model.set_engine_dtype(self.engine_dtype())
        ck = self._restore_model(ckpt_path) if ckpt_path else None
        self._init_distributed(model)
        opt = model.configure_optimizers()
        self.optimizer = opt[0] if isinstance(opt, (list, tuple)) else opt
        if ck is not None:
            self._restore_optimizer(ck)
        acc = self.accumulate_grad_batches
        dp = getattr(model, "dp", None)
        self.optimizer.zero_grad()
        micro = 0
        while self.global_step < self.max_steps:
            sampler = getattr(train_dataloader, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(self.current_epoch)
            seen = 0
            for batch_idx, batch in enumerate(train_dataloader):
                seen += 1
                last = (micro + 1) % acc == 0
                if dp is not None:
                    dp.enabled = last                      # exchange gradients on the final micro-step only
                loss = model.training_step(batch, batch_idx)
                (loss / acc).backward()
                micro += 1
                if last:
                    self.optimizer.step()
                    self.optimizer.zero_grad()
                    self.global_step += 1
                    model.global_step = self.global_step
                    if self.is_global_zero and self.global_step % self.log_every_n_steps == 0:
                        self.logged.append((self.global_step, float(loss.detach())))
                        print(f"[trainer] step {self.global_step} epoch {self.current_epoch} loss {float(loss.detach()):.5f}")
                self._call("on_train_batch_end", model, {"loss": loss.detach()}, batch, batch_idx)
                self._call("on_batch_end", model)
                if self.global_step >= self.max_steps:
                    break
            if seen == 0:
                raise RuntimeError("empty dataloader")
            self.current_epoch += 1
            model.current_epoch = self.current_epoch
        self._call("on_train_end", model)
        return self
