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

from typing import List, Sequence

import torch

from . import hip


def bind_trainables(module: torch.nn.Module, executor) -> List[torch.nn.Parameter]:
    """After this, optimizer updates on the module's trainable Parameters update the engine's fp32
    masters in place, and the engine's backward fills their .grad."""
    params = dict(module.named_parameters())
    out = []
    for t in executor.tr.items:
        p = params[t.name]
        p.data = t.param_view(t.master)        # conv weights: a permuted view of the [O][9][I] master storage
        p.grad = t.param_view(t.grad)
        p.requires_grad_(True)
        out.append(p)
    return out


def bind_bank(params_by_name, bank) -> List[torch.nn.Parameter]:
    """Point one LoRA bank's nn.Parameters (name -> Parameter, names as in the bank's TrainableSet) at its flat buffers."""
    out = []
    for t in bank.items:
        p = params_by_name[t.name]
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
        """Moments BY PARAMETER NAME: the order of the flat buffers is an implementation detail of a build / dtype (the bf16
        engine keeps the grouped emb_layers factors at the tail, nets.py), a checkpoint must not depend on it."""
        def by_name(ex, buf):
            return {t.name: buf[t.offset:t.offset + t.master.numel()].clone() for t in ex.tr.items}
        return dict(step=self._step, format="by_name", m=[by_name(ex, m) for ex, m in zip(self.executors, self._m)],
                    v=[by_name(ex, v) for ex, v in zip(self.executors, self._v)],
                    param_groups=[{k: v for k, v in g.items() if k != "params"} for g in self.param_groups])

    def load_state_dict(self, sd):
        self._step_dev.fill_(int(sd["step"]))
        assert len(sd["m"]) == len(self._m), "optimizer state was saved for a different number of ControlNet banks"
        for key, bufs in (("m", self._m), ("v", self._v)):
            for ex, dst, src in zip(self.executors, bufs, sd[key]):
                if isinstance(src, dict):
                    missing = [t.name for t in ex.tr.items if t.name not in src]
                    if missing:
                        raise KeyError(f"optimizer state lacks {len(missing)} tensors, e.g. {missing[:3]}")
                    for t in ex.tr.items:
                        dst[t.offset:t.offset + t.master.numel()].copy_(src[t.name].reshape(-1))
                else:       # rounds 1-3: ONE flat tensor in the flat buffer's order of the build that wrote it (no emb hoist)
                    _load_legacy_flat(ex, dst, src)
        _restore_param_groups(self, sd)


def _load_legacy_flat(ex, dst, src):
    """A flat moment tensor saved by rounds 1-3 -> this build's layout.  Those builds laid the trainables out in
    `ex.legacy_item_names` order (backward-completion order without the emb_layers hoist), every tensor padded to 64 floats
    (packing.TrainableSet.materialize); the tensors are found by walking that order and copied to today's offsets."""
    from .engine.packing import rup
    names = getattr(ex, "legacy_item_names", None) or [t.name for t in ex.tr.items]
    by_name = {t.name: t for t in ex.tr.items}
    if sorted(names) != sorted(by_name):
        raise KeyError("the legacy optimizer state does not describe this model's trainable set")
    off = 0
    spans = []
    for n in names:
        k = by_name[n].master.numel()
        spans.append((by_name[n], off, k))
        off += rup(k, 64)
    if off != src.numel():
        raise ValueError(f"legacy optimizer state holds {src.numel()} floats, the legacy layout of this model has {off}")
    src = src.reshape(-1)
    for t, o, k in spans:
        dst[t.offset:t.offset + k].copy_(src[o:o + k])


def _restore_param_groups(opt, sd):
    """Resume: the saved hyper-parameters (a scheduler may have moved lr) replace the constructor's, then go to the device."""
    for g, saved in zip(opt.param_groups, sd.get("param_groups") or []):
        for k, v in saved.items():
            if k != "params":
                g[k] = v
    opt.sync_hyper()


class PretrainAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW over ALL ControlNet parameters as the reference's multi-task pre-training uses it
    (cldm/cldm_ctrlora_pretrain.py:174-182, torch 1.13 / Lightning 1.5 semantics), on the engine's flat buffers:

      * the shared (base) parameters are updated every step;
      * a task's LoRA bank joins the update the first time it receives a gradient (before that its .grad is None and
        torch skips it); from then on it is updated EVERY step -- with a zero gradient when another task ran, because
        `optimizer.zero_grad()` of that torch version zero-fills instead of setting None: weight decay and the decaying
        first moment keep moving it.  Each bank therefore has its own step counter (bias correction).

    `mark_used(task)` is called by the model when a task's bank took part in a backward pass."""

    def __init__(self, params, executor, banks, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.executor, self.banks = executor, dict(banks)
        self.executors = [executor]
        self.grad_scale = grad_scale
        dev = executor.tr.flat.device
        self._hyper = torch.zeros(6, dtype=torch.float32, device=dev)
        self._hyper_host = None
        mk = lambda ts: dict(m=torch.zeros_like(ts.flat), v=torch.zeros_like(ts.flat),
                             step=torch.zeros(1, dtype=torch.int32, device=dev))
        self._base = mk(executor.tr)
        self._bank_state = {k: mk(ts) for k, ts in self.banks.items()}
        self.active = []                 # tasks whose bank has received a gradient at least once, in order of first use
        self.pre_step_hook = None
        self.sync_hyper()

    def sync_hyper(self):
        g = self.param_groups[0]
        cur = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
               float(self.grad_scale))
        if cur != self._hyper_host:
            self._hyper.copy_(torch.tensor(cur, dtype=torch.float32))
            self._hyper_host = cur

    def mark_used(self, task):
        if task not in self.active:
            self.active.append(task)

    @property
    def _step(self) -> int:
        return int(self._base["step"].item())

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self.pre_step_hook is not None:
            self.pre_step_hook()
        if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        ex = self.executor
        hip.tick(self._base["step"])
        hip.adamw_dev(ex.tr.flat, ex.tr.flat_grad, self._base["m"], self._base["v"], self._hyper, self._base["step"])
        for task in self.active:
            ts, st = self.banks[task], self._bank_state[task]
            hip.tick(st["step"])
            hip.adamw_dev(ts.flat, ts.flat_grad, st["m"], st["v"], self._hyper, st["step"])
        ex.repack()
        return loss

    def zero_grad(self, set_to_none: bool = False):
        hip.zero_(self.executor.tr.flat_grad)
        for task in self.active:
            hip.zero_(self.banks[task].flat_grad)

    def state_dict(self):
        """Moments BY PARAMETER NAME (format 2), like FusedAdamW: the flat buffers' order is a detail of a build."""
        def pack(ts, st):
            by = lambda buf: {t.name: buf[t.offset:t.offset + t.master.numel()].clone() for t in ts.items}
            return dict(m=by(st["m"]), v=by(st["v"]), step=int(st["step"].item()))
        return dict(format="by_name", version=2, base=pack(self.executor.tr, self._base),
                    banks={k: pack(self.banks[k], v) for k, v in self._bank_state.items()}, active=list(self.active),
                    param_groups=[{k: v for k, v in g.items() if k != "params"} for g in self.param_groups])

    def load_state_dict(self, sd):
        def unpack(ts, st, src):
            for key in ("m", "v"):
                if isinstance(src[key], dict):
                    missing = [t.name for t in ts.items if t.name not in src[key]]
                    if missing:
                        raise KeyError(f"optimizer state lacks {len(missing)} tensors, e.g. {missing[:3]}")
                    for t in ts.items:
                        st[key][t.offset:t.offset + t.master.numel()].copy_(src[key][t.name].reshape(-1))
                else:       # format 1 (rounds 2-4): raw flat buffers; only valid for an unchanged layout -- check what can be checked
                    if src[key].numel() != st[key].numel():
                        raise ValueError(f"flat optimizer state of {src[key].numel()} floats does not fit this build's {st[key].numel()}")
                    st[key].copy_(src[key])
            st["step"].fill_(int(src["step"]))
        if set(sd["banks"]) != set(self._bank_state):
            raise KeyError(f"optimizer state holds banks {sorted(sd['banks'])}, the model has {sorted(self._bank_state)}")
        unpack(self.executor.tr, self._base, sd["base"])
        for k, src in sd["banks"].items():
            unpack(self.banks[k], self._bank_state[k], src)
        self.active = list(sd["active"])
        _restore_param_groups(self, sd)


class GraphedTrainStep:
    """One optimizer step of LoRA fine-tuning as hipGraph replays.

    The eager step is ~1900 kernel launches driven from Python through ctypes; at ~40 ms per step the host
    becomes the bottleneck.  Everything in the step has static shapes and no host dependence (the optimizer's
    step counter and hyper-parameters are device-resident), so it is captured once and replayed.

    One rank:      ONE graph = zero_grad -> p_losses forward -> hand-written backward -> fused AdamW + re-pack.
    Data parallel: the backward is cut at ControlNet stage boundaries into SEGMENT graphs -- a new segment starts
                   whenever the gradient slice finalised so far (the flat buffer is laid out in backward-completion
                   order) has reached `bucket_bytes`:

        graph S0 : zero_grad, forward, UNet-decoder backward, ControlNet middle + deepest stages
        RCCL     : async all-reduce of flat_grad[bucket 0]            } runs on RCCL's stream while
        graph S1 : the next ControlNet stages                          } graph S1 is executing
        RCCL     : async all-reduce of flat_grad[bucket 1]  ...
        graph SN : remaining stages; then every rank waits for its collectives
        graph B  : fused AdamW (grad_scale = 1 / world) + re-pack

    so the LoRA-only gradient exchange overlaps the remaining backward as in the eager path, with collectives kept
    OUT of the captured regions.  `split_graphs`: None = segmented iff world > 1; "segmented" / True force that
    structure on one rank (tests); "two" = the old A | all-reduce | B form; False = one graph.
    `model` is a ControlFinetuneLDM-like module (engine_train_step or p_losses / dp / control_model), `opt` its
    FusedAdamW.
    """

    def __init__(self, model, opt, z, cond_txt, hint, t, noise, warmup: int = 2, split_graphs=None,
                 bucket_bytes: int = 32 << 20, reduce_fn=None, capture_error_mode=None):
        import torch.distributed as dist
        self.model, self.opt = model, opt
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # With a process group alive, RCCL's watchdog thread polls events while this thread captures: only calls made
        # by the capturing thread may invalidate the capture ("thread_local"); single-process keeps the strict default.
        self.capture_error_mode = capture_error_mode or ("thread_local" if self.world > 1 else "global")
        self.dist = dist
        if split_graphs is None:
            mode = "segmented" if self.world > 1 else "one"
        elif split_graphs in ("segmented", True):
            mode = "segmented"
        elif split_graphs == "two":
            mode = "two"
        else:
            mode = "one"
        self.mode = mode
        self.s_z, self.s_ctx, self.s_hint = z.clone(), cond_txt.clone(), hint.clone()
        self.s_t, self.s_noise = t.clone(), noise.clone()
        self.loss = None
        self.loss3 = None
        dp = model.dp
        if dp is not None:
            dp.enabled = False          # no collectives inside the captured regions: this class issues them itself
        # reduce_fn(tensor_slice) -> work handle with .wait() or None; default: RCCL SUM all-reduce, asynchronous
        if reduce_fn is None:
            from .parallel import all_reduce_slice, payload_dtype_from_env
            payload = payload_dtype_from_env()       # CTRLORA_DP_PAYLOAD=bf16: half the bytes per bucket

            def reduce_fn(buf):
                if dist.is_initialized() and self.world > 1:
                    return all_reduce_slice(buf, None, payload)
                return None
        self._reduce_fn = reduce_fn

        direct = getattr(model, "engine_train_step", None)

        def fwd_bwd():
            opt.zero_grad()
            cond = {"c_crossattn": [self.s_ctx], "c_concat": [self.s_hint]}
            if direct is not None:
                # p_losses + backward without autograd: the captured region holds hand-written kernel nodes only
                # (no ATen launch, no memset node); out = {loss_simple, loss_vlb, loss}
                self.loss3 = direct(self.s_z, cond, self.s_t, self.s_noise)
                return self.loss3[2]
            loss, _ = model.p_losses(self.s_z, cond, self.s_t, noise=self.s_noise)
            loss.backward()
            return loss.detach()

        def reduce_all():
            works = [self._reduce_fn(ex.tr.flat_grad) for ex in opt.executors]
            for w in works:
                if w is not None:
                    w.wait()

        hook, opt.pre_step_hook = opt.pre_step_hook, None
        self._hook = hook
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fwd_bwd()
                if mode != "one":
                    reduce_all()
                opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.segments = []               # [(graph, executor index or None, lo, hi)]
        self.g_b = None
        if mode == "one":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=self.capture_error_mode):
                self.loss = fwd_bwd()
                opt.step()
            self.segments.append((g, None, 0, 0))
        elif mode == "two":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=self.capture_error_mode):
                self.loss = fwd_bwd()
            self.segments.append((g, "all", 0, 0))
            self.g_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_b, pool=g.pool(), capture_error_mode=self.capture_error_mode):
                opt.step()
        else:
            self._capture_segments(fwd_bwd, max(1, bucket_bytes // 4))
            self.g_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_b, pool=self._pool, capture_error_mode=self.capture_error_mode):
                opt.step()
        self.warmup_steps = warmup

    def _capture_segments(self, fwd_bwd, bucket_elems):
        """Capture forward + backward as consecutive graphs that end where a gradient bucket is complete.  The
        executors' stage-completion hook (ControlNetE._done -> on_stage_done(start, end), called after the stage's last
        kernel was enqueued) closes the running capture and opens the next one in the same memory pool."""
        opt = self.opt
        execs = list(opt.executors)
        saved_hooks = [ex.on_stage_done for ex in execs]
        self._pool = torch.cuda.graph_pool_handle()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        state = {"g": None, "lo": {id(ex): 0 for ex in execs}, "hi": {id(ex): 0 for ex in execs}}

        def begin():
            state["g"] = torch.cuda.CUDAGraph()
            state["g"].capture_begin(pool=self._pool, capture_error_mode=self.capture_error_mode)

        def end(tag, lo, hi):
            state["g"].capture_end()
            self.segments.append((state["g"], tag, lo, hi))
            state["g"] = None

        def make_hook(i, ex):
            def on_stage(start, stop):
                k = id(ex)
                if start != state["hi"][k]:
                    return                                  # out-of-order report: left to the final flush
                state["hi"][k] = stop
                if stop - state["lo"][k] >= bucket_elems and stop < ex.tr.numel:
                    end(i, state["lo"][k], stop)
                    state["lo"][k] = stop
                    begin()
            return on_stage

        for i, ex in enumerate(execs):
            ex.on_stage_done = make_hook(i, ex)
        try:
            torch.cuda.synchronize()
            with torch.cuda.stream(stream):
                begin()
                self.loss = fwd_bwd()
                # whatever has not been handed out yet (the tail of every bank) goes with the last segment
                g_last = state["g"]
                g_last.capture_end()
                tails = [(i, state["lo"][id(ex)], ex.tr.numel) for i, ex in enumerate(execs) if state["lo"][id(ex)] < ex.tr.numel]
                self.segments.append((g_last, "tails", tails, 0))
            torch.cuda.current_stream().wait_stream(stream)
            torch.cuda.synchronize()
        except BaseException:
            # leave no stream in capture mode behind: the caller falls back to eager launches on this device
            g_open, state["g"] = state["g"], None
            if g_open is not None:
                try:
                    with torch.cuda.stream(stream):
                        g_open.capture_end()
                except Exception:
                    pass
            self.segments = []
            raise
        finally:
            for ex, h in zip(execs, saved_hooks):
                ex.on_stage_done = h

    def __call__(self, z, cond_txt, hint, t, noise):
        self.s_z.copy_(z); self.s_ctx.copy_(cond_txt); self.s_hint.copy_(hint)
        self.s_t.copy_(t); self.s_noise.copy_(noise)
        self.opt.sync_hyper()
        replay_with_exchange(self.segments, self.g_b, self.opt.executors, self._reduce_fn)
        return self.loss


def replay_with_exchange(segments, g_opt, execs, reduce_fn):
    """One optimizer step from captured pieces (anything with .replay()): every segment is enqueued, and right after
    a segment that completes a gradient bucket the reduction of that slice of the flat buffer is issued
    (`reduce_fn(slice)` -> handle with .wait() or None) -- it runs while the NEXT segment executes; all handles are waited for
    before the optimizer piece.  segments: [(graph, tag, lo, hi)], tag None = nothing to exchange, "all" = every executor's
    whole buffer, "tails" = lo is [(executor index, start, stop)], an int = that executor's [lo, hi) slice."""
    from .engine.nets import WEIGHTS_GENERATION
    WEIGHTS_GENERATION[0] += 1         # the optimizer piece re-packs the trainables: caches of the old copies are stale
    works = []
    for g, tag, lo, hi in segments:
        g.replay()
        if tag is None:
            continue
        if tag == "all":
            works += [reduce_fn(ex.tr.flat_grad) for ex in execs]
        elif tag == "tails":
            works += [reduce_fn(execs[i].tr.flat_grad[a:b]) for i, a, b in lo]
        else:
            works.append(reduce_fn(execs[tag].tr.flat_grad[lo:hi]))   # overlaps the next segment's replay
    for w in works:
        if w is not None:
            w.wait()
    if g_opt is not None:
        g_opt.replay()


class GraphedPretrainStep:
    """One optimizer step of multi-task Base-ControlNet pre-training (cldm/cldm_ctrlora_pretrain.py:95-111,174-182) as a
    hipGraph replay: ONE GRAPH PER TASK, all in one memory pool.  A task's graph is self-contained:

        re-pack of that task's LoRA bank into the shared packed copies (the bank switch)  ->  zero_grad  ->  p_losses
        forward + hand-written backward (base weights + that bank)  ->  PretrainAdamW (base + every active bank)  ->  re-pack

    so replaying graphs in any task order is the eager sequence.  PretrainAdamW updates every bank that has EVER received a
    gradient, so the set of active banks is part of what a graph captured: while banks are still joining (a task's first
    step) the step runs eagerly, and graphs are (re-)captured once the active set is what they would bake in.  After a
    replay the host-side bank pointers are moved without another re-pack, so eager code that follows sees a consistent
    executor.  Single process (the data-parallel exchange of pre-training is eager: `_PretrainDP`)."""

    def __init__(self, model, opt, z, cond_txt, hint, t, noise):
        self.model, self.opt = model, opt
        self.s_z, self.s_ctx, self.s_hint = z.clone(), cond_txt.clone(), hint.clone()
        self.s_t, self.s_noise = t.clone(), noise.clone()
        self.graphs = {}             # task -> (graph, loss 3-vector)
        self._active_at_capture = None
        self._pool = None
        self.eager_steps = 0

    def _step(self, task):
        cm = self.model.control_model
        cm.switch_lora(task)
        cm.executor().switch_bank(cm.bank(task), force=True)      # the graph's first kernel: this bank -> packed copies
        self.opt.zero_grad()
        cond = {"c_crossattn": [self.s_ctx], "c_concat": [self.s_hint], "task": task}
        loss3 = self.model.engine_train_step(self.s_z, cond, self.s_t, self.s_noise)
        self.opt.step()
        return loss3

    def __call__(self, task, z, cond_txt, hint, t, noise):
        self.s_z.copy_(z); self.s_ctx.copy_(cond_txt); self.s_hint.copy_(hint)
        self.s_t.copy_(t); self.s_noise.copy_(noise)
        opt, cm = self.opt, self.model.control_model
        if task not in opt.active:                      # the bank joins the optimizer in this step: eager
            self.eager_steps += 1
            return self._step(task)[2]
        if self._active_at_capture != tuple(opt.active):   # the active set changed: every captured optimizer piece is stale
            self.graphs.clear()
            self._active_at_capture = tuple(opt.active)
        ent = self.graphs.get(task)
        if ent is None:
            opt.sync_hyper()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                done = self._step(task).clone()         # warm-up of this task's allocation pattern: it IS this call's step
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.eager_steps += 1
            with torch.cuda.graph(g, pool=self._pool):
                loss3 = self._step(task)                # capture does not execute
            if self._pool is None:
                self._pool = g.pool()
            self.graphs[task] = (g, loss3)
            return done[2]
        opt.sync_hyper()
        from .engine.nets import WEIGHTS_GENERATION
        WEIGHTS_GENERATION[0] += 1
        ent[0].replay()
        cm._task = task
        for lin, lora in zip(cm._lora_linears, cm.loras_dict[task]):
            lin.set_lora_layer(lora)
        cm.executor().switch_bank(cm.bank(task), repack=False)
        return ent[1][2]
