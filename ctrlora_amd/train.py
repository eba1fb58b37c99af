"""Training glue between torch.autograd / torch.optim and the HIP engine.

  * ApplyModelFn   -- one autograd node for the whole ControlNet+UNet pass: forward records,
                      backward runs the hand-written backward and fills the flat fp32 gradient buffers
                      (the trainable nn.Parameters' .grad are views of those buffers).
  * PLossFn        -- p_losses' reduction {loss_simple, loss_vlb, loss} + d loss / d eps, one deterministic kernel pair
                      (ddpm.py:902-918 with logvar = 0).
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


class PLossFn(torch.autograd.Function):
    """LatentDiffusion.p_losses' reduction (ddpm.py:902-918 with logvar = 0) as ONE deterministic HIP reduction:
    returns the 3-vector {loss_simple, loss_vlb, loss = w_simple * loss_simple + w_elbo * loss_vlb} and keeps
    d(w_simple * loss_simple) / d eps for the backward (the elbo term is not differentiated; its weight is 0 in
    every CtrLoRA config and the caller checks that)."""

    @staticmethod
    def forward(ctx, eps, target, t, lvlb, w_simple, w_elbo):
        eps = eps.float().contiguous()
        target = target.float().contiguous()
        out = torch.empty(3, dtype=torch.float32, device=eps.device)
        scratch = torch.empty(16 * eps.shape[0], dtype=torch.float32, device=eps.device)
        d_eps = torch.empty_like(eps)
        hip.p_losses_mse(eps, target, d_eps, t.long().contiguous(), lvlb, out, scratch, 1.0, float(w_simple), float(w_elbo))
        ctx.save_for_backward(d_eps)
        return out

    @staticmethod
    def backward(ctx, g):
        (d_eps,) = ctx.saved_tensors
        return d_eps * g[2], None, None, None, None, None


class FusedAdamW(torch.optim.Optimizer):
    """AdamW over the engine's flat buffers.  `params` are the bound nn.Parameters (kept in param_groups
    for scheduler / checkpoint compatibility); the arithmetic is one HIP kernel per ControlNet bank.

    Step count and hyper-parameters live in DEVICE memory (cl_adamw_dev), so that `step()` can be captured
    in a hipGraph and replayed: a learning-rate schedule only has to refresh the 6-float `hyper` tensor
    (`sync_hyper()`, done automatically outside capture)."""

    def __init__(self, params, executors, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 grad_scale: float = 1.0):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.executors = list(executors)
        self.grad_scale = grad_scale
        dev = self.executors[0].tr.flat.device
        self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._hyper = torch.zeros(6, dtype=torch.float32, device=dev)
        self._hyper_host = None
        self._m = [torch.zeros_like(ex.tr.flat) for ex in self.executors]
        self._v = [torch.zeros_like(ex.tr.flat) for ex in self.executors]
        self.pre_step_hook = None     # e.g. DP: wait for the gradient all-reduce
        self.sync_hyper()

    @property
    def _step(self) -> int:
        return int(self._step_dev.item())

    def sync_hyper(self):
        """Push {lr, betas, eps, weight_decay, grad_scale} to the device if they changed (host -> device copy:
        call it OUTSIDE graph capture / between replays)."""
        g = self.param_groups[0]
        cur = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
               float(self.grad_scale))
        if cur != self._hyper_host:
            self._hyper.copy_(torch.tensor(cur, dtype=torch.float32))
            self._hyper_host = cur

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self.pre_step_hook is not None:
            self.pre_step_hook()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        hip.tick(self._step_dev)          # one tick per optimizer step, however many banks follow
        for ex, m, v in zip(self.executors, self._m, self._v):
            hip.adamw_dev(ex.tr.flat, ex.tr.flat_grad, m, v, self._hyper, self._step_dev)
            ex.repack()
        return loss

    def zero_grad(self, set_to_none: bool = False):
        # .grad tensors are views of flat_grad: zero in place, they must stay attached
        for ex in self.executors:
            hip.zero_(ex.tr.flat_grad)

    def state_dict(self):
        return dict(step=self._step, m=[m.clone() for m in self._m], v=[v.clone() for v in self._v],
                    param_groups=[{k: v for k, v in g.items() if k != "params"} for g in self.param_groups])

    def load_state_dict(self, sd):
        self._step_dev.fill_(int(sd["step"]))
        for dst, src in zip(self._m, sd["m"]):
            dst.copy_(src)
        for dst, src in zip(self._v, sd["v"]):
            dst.copy_(src)


class GraphedTrainStep:
    """One optimizer step of LoRA fine-tuning as hipGraph replays.

    The eager step is ~2800 kernel launches driven from Python through ctypes; at < 50 ms per step the host
    becomes the bottleneck.  Everything in the step has static shapes and no host dependence (the optimizer's
    step counter and hyper-parameters are device-resident), so it is captured once and replayed:

        graph A : zero_grad -> p_losses forward -> hand-written backward      (all ranks, no collectives)
        eager   : all-reduce of the flat LoRA gradient buffer over RCCL        (world_size > 1 only)
        graph B : fused AdamW + re-pack of the trainables

    With one rank, A and B are a single graph.  Inputs are copied into static buffers before each replay.
    `model` is a ControlFinetuneLDM-like module (p_losses / dp / control_model), `opt` its FusedAdamW.
    """

    def __init__(self, model, opt, z, cond_txt, hint, t, noise, warmup: int = 2, split_graphs=None):
        import torch.distributed as dist
        self.model, self.opt = model, opt
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.dist = dist
        # two graphs with the all-reduce in between whenever there is more than one rank (or when forced, to
        # exercise that structure on a single GPU)
        split = (self.world > 1) if split_graphs is None else bool(split_graphs)
        self.s_z, self.s_ctx, self.s_hint = z.clone(), cond_txt.clone(), hint.clone()
        self.s_t, self.s_noise = t.clone(), noise.clone()
        self.loss = None
        self.loss3 = None
        dp = model.dp
        if dp is not None:
            dp.enabled = False          # no collectives inside the captured region

        direct = getattr(model, "engine_train_step", None)

        def fwd_bwd():
            opt.zero_grad()
            cond = {"c_crossattn": [self.s_ctx], "c_concat": [self.s_hint]}
            if direct is not None:
                # p_losses + backward without autograd: the captured region holds hand-written kernels and memset
                # nodes only (no ATen launch); out = {loss_simple, loss_vlb, loss}
                self.loss3 = direct(self.s_z, cond, self.s_t, self.s_noise)
                return self.loss3[2]
            loss, _ = model.p_losses(self.s_z, cond, self.s_t, noise=self.s_noise)
            loss.backward()
            return loss.detach()

        def reduce_grads():
            if split and dist.is_initialized():
                for ex in opt.executors:
                    dist.all_reduce(ex.tr.flat_grad, op=dist.ReduceOp.SUM)

        hook, opt.pre_step_hook = opt.pre_step_hook, None
        self._hook = hook
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fwd_bwd(); reduce_grads(); opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.g_a = torch.cuda.CUDAGraph()
        if not split:
            with torch.cuda.graph(self.g_a):
                self.loss = fwd_bwd()
                opt.step()
            self.g_b = None
        else:
            with torch.cuda.graph(self.g_a):
                self.loss = fwd_bwd()
            self.g_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_b, pool=self.g_a.pool()):
                opt.step()
        self._reduce = reduce_grads
        self.warmup_steps = warmup

    def __call__(self, z, cond_txt, hint, t, noise):
        self.s_z.copy_(z); self.s_ctx.copy_(cond_txt); self.s_hint.copy_(hint)
        self.s_t.copy_(t); self.s_noise.copy_(noise)
        self.opt.sync_hyper()
        self.g_a.replay()
        if self.g_b is not None:
            self._reduce()
            self.g_b.replay()
        return self.loss
