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
        self.payload_dtype = payload_dtype if payload_dtype is not None else payload_dtype_from_env()
        # overlapped form (attach): buckets of the backward-ordered shared buffer launched from the executor's
        # stage-completion hook, as GradAllReduce does for fine-tuning
        self._ex = None
        self._lo = self._hi = 0
        self._pending: List = []
        self.enabled = True
        self.launches = 0                 # bucket all-reduces issued by the hook since the last exchange()
        self.launches_before_last_stage = 0
        self.exposed_tail_elems = 0       # elements of the shared buffer that were reduced only in exchange()

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

    def _stage_done(self, start, end):
        if not self.enabled or self.world_size == 1 or self._ex is None:
            return
        if start == self._last_stage_start:
            self.launches_before_last_stage = self.launches
        if start != self._hi:
            return                          # out-of-order report: left to exchange()
        self._hi = end
        if self._hi - self._lo >= self.bucket_elems:
            self._pending.append(all_reduce_slice(self._ex.tr.flat_grad[self._lo:self._hi], self.group, self.payload_dtype))
            self.launches += 1
            self._lo = self._hi

    @torch.no_grad()
    def exchange(self, used_tasks) -> List[str]:
        used = set(used_tasks)
        assert used <= set(self.tasks), f"unknown task(s) {sorted(used - set(self.tasks))}"
        if self.world_size == 1:
            return [t for t in self.tasks if t in used]
        ref = self.shared[0] if self.shared else next(iter(self.banks.values()))
        mask = torch.tensor([1 if t in used else 0 for t in self.tasks], dtype=torch.int32, device=ref.device)
        dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
        live = [t for t, m in zip(self.tasks, mask.tolist()) if m]
        work = list(self._pending)
        self._pending = []
        if self._ex is not None:            # the part of the shared buffer the hook has not launched yet
            n = self._ex.tr.flat_grad.numel()
            self.exposed_tail_elems = n - self._lo
            if self._lo < n:
                work.append(all_reduce_slice(self._ex.tr.flat_grad[self._lo:n], self.group, self.payload_dtype))
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
