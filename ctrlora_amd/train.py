"""Training glue between torch.autograd / torch.optim and the HIP engine.

  * ApplyModelFn   -- one autograd node for the whole ControlNet+UNet pass: forward records,
                      backward runs the hand-written backward and fills the flat fp32 gradient buffers
                      (the trainable nn.Parameters' .grad are views of those buffers).
  * MSELossFn      -- fused mean((eps - target)^2) + its gradient (ddpm.py:902-918 with logvar = 0).
  * FusedAdamW     -- torch.optim.AdamW semantics (cldm_ctrlora_finetune.py:105) as ONE kernel over the
                      flat master/grad buffers, followed by the re-pack of the trainables.
  * bind_trainables -- re-points the ControlNet's trainable nn.Parameters at the flat buffers.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import hip


def bind_trainables(module: torch.nn.Module, executor) -> List[torch.nn.Parameter]:
    """After this, optimizer updates on the module's trainable Parameters update the engine's fp32
    masters in place, and the engine's backward fills their .grad."""
    params = dict(module.named_parameters())
    out = []
    for t in executor.tr.items:
        p = params[t.name]
        p.data = t.master
        p.grad = t.grad
        p.requires_grad_(True)
        out.append(p)
    return out


def trainables_version(params: Sequence[torch.nn.Parameter]) -> int:
    return sum(p._version for p in params)


class ApplyModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, engine, x_noisy, t, context, hints, scales, weights, only_mid, on_backward):
        ctx.engine, ctx.on_backward = engine, on_backward
        return engine.forward(x_noisy, t, context, hints, control_scales=scales, lora_weights=weights, record=True,
                              only_mid_control=only_mid)

    @staticmethod
    def backward(ctx, d_eps):
        ctx.engine.backward(d_eps)
        if ctx.on_backward is not None:
            ctx.on_backward()
        return (None,) * 10


class MSELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eps, target):
        eps = eps.float().contiguous()
        target = target.float().contiguous()
        loss = torch.zeros((), dtype=torch.float32, device=eps.device)
        d_eps = torch.empty_like(eps)
        hip.mse_loss(eps, target, d_eps, loss)
        ctx.save_for_backward(d_eps)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_eps,) = ctx.saved_tensors
        return d_eps * g, None


class FusedAdamW(torch.optim.Optimizer):
    """AdamW over the engine's flat buffers.  `params` are the bound nn.Parameters (kept in param_groups
    for scheduler / checkpoint compatibility); the arithmetic is one HIP kernel per ControlNet bank."""

    def __init__(self, params, executors, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 grad_scale: float = 1.0):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.executors = list(executors)
        self.grad_scale = grad_scale
        self._step = 0
        self._m = [torch.zeros_like(ex.tr.flat) for ex in self.executors]
        self._v = [torch.zeros_like(ex.tr.flat) for ex in self.executors]
        self.pre_step_hook = None     # e.g. DP: wait for the gradient all-reduce

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self.pre_step_hook is not None:
            self.pre_step_hook()
        self._step += 1
        g = self.param_groups[0]
        for ex, m, v in zip(self.executors, self._m, self._v):
            hip.adamw(ex.tr.flat, ex.tr.flat_grad, m, v, g["lr"], self._step, g["betas"][0], g["betas"][1], g["eps"],
                      g["weight_decay"], self.grad_scale)
            ex.repack()
        return loss

    def zero_grad(self, set_to_none: bool = False):
        # .grad tensors are views of flat_grad: zero in place, they must stay attached
        for ex in self.executors:
            ex.tr.flat_grad.zero_()

    def state_dict(self):
        return dict(step=self._step, m=[m.clone() for m in self._m], v=[v.clone() for v in self._v],
                    param_groups=[{k: v for k, v in g.items() if k != "params"} for g in self.param_groups])

    def load_state_dict(self, sd):
        self._step = sd["step"]
        for dst, src in zip(self._m, sd["m"]):
            dst.copy_(src)
        for dst, src in zip(self._v, sd["v"]):
            dst.copy_(src)
