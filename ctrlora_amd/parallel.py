"""Data-parallel LoRA fine-tuning: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm)
over xGMI, all-reduce on the optimizer's gradient subset ONLY.

The reference trains through Lightning DDP, which all-reduces every parameter that received a gradient
(~3.5 GB fp32 per step incl. 0.5 G dead UNet-decoder weight gradients).  Here the trainable
gradients live in one flat fp32 buffer per ControlNet (148 MB for rank 128) laid out in backward-completion
order, so the exchange is a handful of large contiguous all-reduces launched while the remaining backward is
still running (the ControlNet backward calls `on_stage_done(start, end)` after each encoder stage).
Averaging is folded into the optimizer (grad_scale = 1 / world_size).
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class _CastWork:
    """Handle of a reduced-precision exchange: the collective ran on a cast copy; wait() writes the sum back."""

    def __init__(self, work, dst, tmp):
        self.work, self.dst, self.tmp = work, dst, tmp

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.tmp)


def payload_dtype_from_env():
    import os
    v = os.environ.get("CTRLORA_DP_PAYLOAD", "f32").lower()
    return torch.bfloat16 if v in ("bf16", "bfloat16") else None


def all_reduce_slice(buf: torch.Tensor, group=None, payload_dtype=None):
    """Asynchronous SUM all-reduce of one slice of a flat fp32 gradient buffer; returns a handle with .wait().
    payload_dtype=torch.bfloat16 halves the bytes on the wire (the sum is then formed in bf16 by the collective: a
    relative error of ~2^-8 per addend, acceptable for LoRA gradients only when the links, not the backward, bound the
    step -- fp32 is the default: 148 MB per step hide under the ControlNet backward on xGMI)."""
    if payload_dtype is None or payload_dtype == buf.dtype:
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
    tmp = buf.to(payload_dtype)
    return _CastWork(dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group, async_op=True), buf, tmp)


class GradAllReduce:
    def __init__(self, executors, group=None, bucket_bytes: int = 32 << 20, overlap: bool = True, payload_dtype=None):
        self.group = group
        self.payload_dtype = payload_dtype if payload_dtype is not None else payload_dtype_from_env()
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.overlap = overlap
        self.enabled = True          # set False on non-final gradient-accumulation micro-steps
        self.executors = list(executors)
        self._pending: List = []
        self._lo = {id(ex): 0 for ex in self.executors}
        self._hi = {id(ex): 0 for ex in self.executors}
        self.launched_bytes = 0
        self.launches = 0
        for ex in self.executors:
            ex.on_stage_done = (lambda s, e, ex=ex: self._stage_done(ex, s, e))

    def _launch(self, ex, lo, hi):
        if hi <= lo or self.world_size == 1:
            return
        buf = ex.tr.flat_grad[lo:hi]
        self._pending.append(all_reduce_slice(buf, self.group, self.payload_dtype))
        self.launched_bytes += (hi - lo) * (4 if self.payload_dtype is None else torch.empty(0, dtype=self.payload_dtype).element_size())
        self.launches += 1

    def _stage_done(self, ex, start, end):
        if not self.enabled:
            return
        k = id(ex)
        # stages complete in increasing offset order; anything else falls back to the final flush
        if start == self._hi[k]:
            self._hi[k] = end
            if self.overlap and self._hi[k] - self._lo[k] >= self.bucket_elems:
                self._launch(ex, self._lo[k], self._hi[k])
                self._lo[k] = self._hi[k]

    def on_backward_done(self):
        """Called at the end of the engine backward: flush what has not been launched yet."""
        if not self.enabled:
            return
        for ex in self.executors:
            k = id(ex)
            self._launch(ex, self._lo[k], ex.tr.numel)
            self._lo[k] = self._hi[k] = 0

    def wait(self):
        """Before the optimizer step: the current stream waits for every outstanding all-reduce."""
        for w in self._pending:
            w.wait()
        self._pending.clear()


class BankedGradAllReduce:
    """Gradient exchange for multi-task pre-training under data parallelism.

    The reference's `BatchSchedulerSampler` (datasets/multi_task_scheduler.py:59) draws the task order from an
    unseeded per-rank numpy RNG, so in one step different ranks may train different tasks: each rank touches the
    shared (base-ControlNet) gradients and ONE LoRA bank.  Lightning DDP then averages every parameter over the
    world size, ranks that did not use a bank contributing zeros.  Restated here without moving dead bytes:

      * `shared`: flat gradient buffers every rank produces (all-reduced every step);
      * `banks[task]`: one flat gradient buffer per task bank; a small MAX all-reduce of the per-rank "used"
        mask tells every rank which banks are live anywhere this step, and only those are all-reduced (a rank
        that did not use a live bank contributes zeros).  With T tasks and N ranks at most min(T, N) banks move
        instead of T.

    The result in every buffer is the SUM over ranks; averaging (1 / world_size, as DDP does) is left to the
    optimizer's `grad_scale`, like `GradAllReduce`.  Returns the list of tasks whose banks were exchanged.
    """

    def __init__(self, shared, banks, group=None, bucket_bytes: int = 32 << 20, payload_dtype=None):
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.shared = list(shared)
        self.tasks = list(banks.keys())
        self.banks = dict(banks)
        self.bucket_elems = max(1, bucket_bytes // 4)
        # The LoRA banks may travel in bf16 (CTRLORA_DP_PAYLOAD=bf16, as in fine-tuning); the SHARED base-ControlNet gradients
        # stay fp32 unless CTRLORA_DP_PAYLOAD_SHARED=bf16 opts them in as well (they went out in fp32 unconditionally
        # before the overlapped form existed)
        self.payload_dtype = payload_dtype if payload_dtype is not None else payload_dtype_from_env()
        import os
        self.shared_payload_dtype = payload_dtype if payload_dtype is not None else (
            torch.bfloat16 if os.environ.get("CTRLORA_DP_PAYLOAD_SHARED", "f32").lower() in ("bf16", "bfloat16") else None)
        # overlapped form (attach): buckets of the backward-ordered shared buffer launched from the executor's
        # stage-completion hook, as GradAllReduce does for fine-tuning
        self._ex = None
        self._lo = self._hi = 0
        self._pending: List = []
        self.enabled = True
        self.launches = 0                 # bucket all-reduces issued by the hook since the last exchange()
        self.last_launches = 0            # ... during the step the last exchange() closed
        self.launches_before_last_stage = 0
        self.exposed_tail_elems = 0       # elements of the shared buffer that were reduced only in exchange()
        # "which banks are live on ANY rank" (prefetch_mask): exchanged while the step computes, read without a stream sync
        self._mask_pre = None
        self._mask_stream = None
        self._mask_host = None
        self.mask_prefetch_hits = 0

    def attach(self, executor):
        """Overlap the exchange of the base-ControlNet gradients (360 M floats at SD1.5 width: ~1.4 GB fp32 -- the reference's
        DDP buckets and overlaps them, scripts/train_ctrlora_pretrain.py:117-121) with the backward pass: `executor` keeps
        them in ONE flat buffer laid out in backward-completion order and reports every finished stage through
        `on_stage_done(start, end)`; a bucket of >= bucket_bytes is all-reduced asynchronously as soon as it is final and
        runs under the remaining stages.  exchange() then only has the mask, the tail and the (small) banks left."""
        assert len(self.shared) == 1 and self.shared[0].data_ptr() == executor.tr.flat_grad.data_ptr()
        self._ex = executor
        self._last_stage_start = executor.backward_stage_order()[-1][0] if hasattr(executor, "backward_stage_order") else None
        executor.on_stage_done = self._stage_done
        return self

    def reset(self):
        """Drop the hook's per-step state: wait for (and forget) every bucket already in flight, rewind the cursor.  Called at
        the start of every backward pass (an aborted backward, or `enabled` flipped off after buckets went out, must not leave
        a stale cursor: part of the buffer would be reduced twice and part skipped in the next step).
        CONTRACT: the spans [0, cursor) that were already in flight HAVE been summed across ranks in place; after a
        reset that interrupts a step the gradient buffer must be zeroed (optimizer.zero_grad()) before the next backward pass,
        exactly as after an optimizer step -- the regular accumulation flow always does (exchange() drains the buckets before
        `enabled` goes False again).  (A used-bank mask prefetched for an abandoned step is superseded by the next
        prefetch_mask(), see there.)"""
        for w in self._pending:
            w.wait()
        self._pending = []
        self.interrupted = bool(self._lo or self._hi)      # diagnostics: the last reset cut a step short
        self._lo = self._hi = 0
        self.launches = 0
        self.launches_before_last_stage = 0

    def __setattr__(self, name, value):
        if name == "enabled" and not value and getattr(self, "_pending", None):
            object.__setattr__(self, name, value)
            self.reset()
            return
        object.__setattr__(self, name, value)

    def _stage_done(self, start, end):
        if not self.enabled or self.world_size == 1 or self._ex is None:
            return
        if start == 0 and (self._lo or self._hi or self._pending):
            self.reset()                    # first stage of a new backward (offset 0) with leftovers of an unfinished one
        if start == self._last_stage_start:
            self.launches_before_last_stage = self.launches
        if start != self._hi:
            return                          # out-of-order report: left to exchange()
        self._hi = end
        if self._hi - self._lo >= self.bucket_elems:
            self._pending.append(all_reduce_slice(self._ex.tr.flat_grad[self._lo:self._hi], self.group, self.shared_payload_dtype))
            self.launches += 1
            self._lo = self._hi

    @torch.no_grad()
    def prefetch_mask(self, used_tasks):
        """Start the MAX all-reduce of the used-bank mask BEFORE the forward pass of the final micro-step -- the tasks a rank
        trains in an optimizer step are known when its last batch is drawn (datasets/multi_task_scheduler.py:45-59) -- and
        land the result in pinned host memory from a side stream.  exchange() then reads it after one event wait that has
        long been satisfied, instead of a `mask.tolist()` that blocks the host until the whole backward has drained and
        leaves the GPU idle while the collectives are launched.  A COLLECTIVE: every rank calls it at the same point."""
        if self.world_size == 1:
            return
        used = frozenset(used_tasks)
        assert used <= set(self.tasks), f"unknown task(s) {sorted(used - set(self.tasks))}"
        if self._mask_pre is not None:
            if self._mask_pre[0] == used:
                return
            # A prefetch that no exchange() consumed -- a grad-enabled forward that was not followed by a backward pass (logging,
            # a sanity pass, a skipped step) -- is SUPERSEDED: its collective is waited for and its result dropped, then the new
            # set goes out.  Like every collective this must happen on all ranks alike: every prefetch_mask() is paired with
            # exactly one exchange() or one superseding prefetch_mask() on every rank.
            stale = self._mask_pre
            self._mask_pre = None
            if stale[3]:
                stale[2].synchronize()
            else:
                stale[2].wait()
            self.mask_prefetch_superseded = getattr(self, "mask_prefetch_superseded", 0) + 1
        ref = self.shared[0] if self.shared else next(iter(self.banks.values()))
        mask = torch.tensor([1 if t in used else 0 for t in self.tasks], dtype=torch.int32, device=ref.device)
        work = dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        if mask.is_cuda:
            if self._mask_stream is None:
                self._mask_stream = torch.cuda.Stream(device=mask.device)
                self._mask_host = torch.empty(len(self.tasks), dtype=torch.int32).pin_memory()
            with torch.cuda.stream(self._mask_stream):
                work.wait()                                  # stream-level: the side stream waits for the collective
                self._mask_host.copy_(mask, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._mask_stream)
            self._mask_pre = (used, mask, ev, True)
        else:
            self._mask_pre = (used, mask, work, False)

    def _live_mask(self, used):
        pre, self._mask_pre = self._mask_pre, None
        if pre is not None:
            if pre[0] != frozenset(used):
                raise RuntimeError("exchange() got a different used-task set than prefetch_mask(); ranks would diverge")
            self.mask_prefetch_hits += 1
            if pre[3]:
                pre[2].synchronize()
                return self._mask_host.tolist()
            pre[2].wait()
            return pre[1].tolist()
        ref = self.shared[0] if self.shared else next(iter(self.banks.values()))
        mask = torch.tensor([1 if t in used else 0 for t in self.tasks], dtype=torch.int32, device=ref.device)
        dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
        return mask.tolist()

    @torch.no_grad()
    def exchange(self, used_tasks) -> List[str]:
        used = set(used_tasks)
        assert used <= set(self.tasks), f"unknown task(s) {sorted(used - set(self.tasks))}"
        if self.world_size == 1:
            return [t for t in self.tasks if t in used]
        live = [t for t, m in zip(self.tasks, self._live_mask(used)) if m]
        work = list(self._pending)
        self._pending = []
        if self._ex is not None:            # the part of the shared buffer the hook has not launched yet
            n = self._ex.tr.flat_grad.numel()
            self.exposed_tail_elems = n - self._lo
            if self._lo < n:
                work.append(all_reduce_slice(self._ex.tr.flat_grad[self._lo:n], self.group, self.shared_payload_dtype))
            self._lo = self._hi = 0
        else:
            work += [dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for buf in self.shared]
        for t in live:
            if t not in used:
                self.banks[t].zero_()       # this rank did not train the bank: zero contribution
            work.append(dist.all_reduce(self.banks[t], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in work:
            w.wait()
        self.last_launches, self.launches = self.launches, 0
        return live
