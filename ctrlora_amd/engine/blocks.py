"""Forward / backward of the CtrLoRA building blocks on the HIP kernels.

Activations are token-major ("NHWC") 2-D tensors [B*H*W, C] in the engine dtype.  Every
block has an explicit `fwd` (optionally recording what its `bwd` needs) and a hand-written
`bwd` that produces data gradients and accumulates gradients of the *trainable* tensors only
(LoRA A/B, zero convs, `norm` layers) -- no dW is ever formed for a frozen weight and nothing
is recomputed (the algorithmic minimum, vs. the reference's
checkpoint()-recompute + dead weight-gradients).

Reference modules restated (behaviour, not code):
  ResBlock._forward                ldm/modules/diffusionmodules/openaimodel.py:254-274
  Downsample / Upsample            openaimodel.py:108-118,157-159
  SpatialTransformer.forward       ldm/modules/attention.py:321-340
  BasicTransformerBlock._forward   attention.py:271-275
  CrossAttention.forward           attention.py:163-194
  GEGLU / FeedForward              attention.py:49-76
  LoRACompatibleLinear.forward     cldm/lora.py:285-291
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch

from .. import hip
from .packing import Conv3W, LinearW, NormW, rup

# A/B switch: 0 = the attention kernels get a plain q and scale the scores themselves (round-3 behaviour)
PRESCALE_Q = os.environ.get("CTRLORA_PRESCALE_Q", "1") != "0"


class Ctx:
    """Per-call execution context: dtype, device, whether to record for backward, scratch."""

    def __init__(self, dtype: torch.dtype, device, record: bool):
        self.dtype = dtype
        self.device = device
        self.record = record
        self._gn_ws: Optional[torch.Tensor] = None
        self._tcache: Dict[Tuple[int, int, int, int], torch.Tensor] = {}
        # queued weight-gradient problems (dy, x, dW, scale): flushed as ONE grouped launch per block
        self._wq: list = []
        self._bq: list = []       # queued bias gradients (dy, db, rows, scale)
        # Weight gradients feed nothing but the optimizer: with `wstream` set, a flushed group runs on that stream
        # (own split scratch) while the main stream goes on with the next stage's data gradients.  One group is in
        # flight at a time: (done event, operands kept alive, callback to run once the main stream has joined it).
        self.wstream: Optional["torch.cuda.Stream"] = None
        self._wpending = None

    def queue_wgrad(self, dy, x, dW, scale=1.0, conv=None):
        self._wq.append((dy, x, dW, scale, conv))  # keeps dy / x alive until the flush
        if len(self._wq) >= 24:
            self.flush_wgrad()

    def queue_bias_grad(self, dy, db, rows: int, scale: float = 1.0):
        """db += scale * column sums of dy: rides with the stage's weight-gradient group when that runs on the side stream."""
        if self.wstream is None:
            hip.colsum(dy, db, 1, rows, scale)
        else:
            self._bq.append((dy, db, rows, scale))

    def _launch_group(self):
        if self._wq:
            hip.weight_grad_tn_group(self._wq)
        for dy, db, rows, scale in self._bq:
            hip.colsum(dy, db, 1, rows, scale)

    def flush_wgrad(self, after=None):
        """Launch the queued group.  `after()` runs when the group's results are ordered before everything the main
        stream does next: at once without a side stream, at the NEXT flush / retire_wgrad() with one."""
        if self.wstream is None:
            self._launch_group()
            self._wq, self._bq = [], []
            if after is not None:
                after()
            return
        self.retire_wgrad()
        if self._wq or self._bq:
            main = torch.cuda.current_stream()
            self.wstream.wait_stream(main)
            with torch.cuda.stream(self.wstream):
                self._launch_group()
                done = torch.cuda.Event()
                done.record(self.wstream)
            self._wpending = (done, (self._wq, self._bq), after)
            self._wq, self._bq = [], []
        elif after is not None:
            after()

    def retire_wgrad(self):
        """Join the group in flight into the main stream, release its operands, run its callback."""
        if self._wpending is not None:
            done, _held, after = self._wpending
            self._wpending = None
            torch.cuda.current_stream().wait_event(done)
            if after is not None:
                after()

    def new(self, rows: int, cols: int, dtype=None) -> torch.Tensor:
        return torch.empty((rows, cols), dtype=dtype or self.dtype, device=self.device)

    def zeros(self, rows: int, cols: int, dtype=None) -> torch.Tensor:
        return hip.zero_(torch.empty((rows, cols), dtype=dtype or self.dtype, device=self.device))

    def gn_ws(self, B: int, HW: int, C: int) -> torch.Tensor:
        n = hip.groupnorm_ws(B, HW, C)
        if self._gn_ws is None or self._gn_ws.numel() < n:
            self._gn_ws = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=self.device)
        return self._gn_ws

    # transposed, zero-padded copy [C, Mp] of a [M, C] activation (weight-gradient operand);
    # cached for the lifetime of one block backward because q/k/v share their input.
    def transposed(self, x: torch.Tensor) -> torch.Tensor:
        key = (x.data_ptr(), x.shape[0], x.shape[1], x.stride(0))
        hit = self._tcache.get(key)
        if hit is None:
            M, Cc = x.shape
            Mp = rup(M, 32)
            t = torch.empty((Cc, Mp), dtype=x.dtype, device=x.device)
            hip.transpose(x, t, 1, M, Cc, Mp)
            hit = (x, t)     # keep the source alive so its address cannot be recycled while cached
            self._tcache[key] = hit
        return hit[1]

    def drop_transposes(self):
        self._tcache.clear()


# --------------------------------------------------------------------------- linear

def linear_fwd(ctx: Ctx, L: LinearW, x, out=None, residual=None, act=hip.ACT_NONE, alpha=1.0, beta=1.0,
               out_f32=False, alpha_n=0, ln=None):
    """y = (x W^T + b [+ (x A^T) B^T]) * alpha + beta * residual.  Returns (y, t = x A^T or None).
    alpha_n > 0: alpha multiplies output columns [0, alpha_n) only (the q part of a fused q | k | v).
    ln = (gamma, beta, eps, stats): x is UN-normalised and LayerNorm runs as the product's prologue (ln_prologue_ok)."""
    M = x.shape[0]
    t = None
    merged = L.Wm is not None and not ctx.record      # inference executor: W + B A already folded
    assert ln is None or not (L.r and not merged), "LayerNorm prologue: the product may not carry a live LoRA segment"
    if L.r and not merged:
        t = ctx.new(M, L.r)
        hip.gemm(x, L.A, t)
    if out is None:
        out = ctx.new(M, L.N, torch.float32 if out_f32 else None)
    hip.gemm(x, L.Wm if merged else L.W, out, a2=t, w2=L.B if t is not None else None, bias=L.bias, residual=residual,
             alpha=alpha, beta=beta if residual is not None else 0.0, act=act, out_f32=out_f32, alpha_n=alpha_n, ln=ln)
    return out, t


def ln_prologue_ok(ctx: Ctx, M: int, N: int, K: int, lora_live: bool, act=hip.ACT_NONE) -> bool:
    """May LayerNorm(x) . W^T run as ONE launch here (hip.xs_ln_ok: the x-stationary kernel normalises its rows in registers)?
    Only where nothing else reads the normalised tensor: a product with a live LoRA segment needs it for t = LN(x) A^T and, in
    training, for dA (cldm/lora.py:285-291) -- the frozen UNet and the LoRA-merged inference executors qualify."""
    return ctx.dtype == torch.bfloat16 and not lora_live and hip.xs_ln_ok(M, N, K, act)


def group_fwd(ctx: Ctx, grp, x, alpha=1.0, alpha_n=0, ln=None):
    """Every member of a packing.LoraGroup applied to x in two launches: (y [M, G N] (+ bias), t [M, G r] or None).
    alpha / alpha_n: as linear_fwd (the first member's output scaled in the product's epilogue)."""
    M = x.shape[0]
    y = ctx.new(M, grp.G * grp.N)
    if grp.Wm is not None and not ctx.record:              # inference executor: W + B A folded, one plain product
        hip.gemm(x, grp.Wm, y, bias=grp.bias, alpha=alpha, alpha_n=alpha_n, ln=ln)
        return y, None
    assert ln is None, "LayerNorm prologue: the grouped product carries live LoRA segments"
    t = ctx.new(M, grp.G * grp.r)
    hip.gemm(x, grp.A, t)
    hip.gemm(x, grp.W, y, a2=t, w2=grp.B, bias=grp.bias, a2_group_n=grp.N, alpha=alpha, alpha_n=alpha_n)
    return y, t


def linear_bwd_data(ctx: Ctx, L: LinearW, dy, out=None, accum=None):
    """dx = dy W + (dy B) A (+ accum).  Returns (dx, u = dy B or None)."""
    M = dy.shape[0]
    u = None
    if L.r:
        u = ctx.new(M, L.r)
        hip.gemm(dy, L.Bt, u)
    if out is None:
        out = ctx.new(M, L.K)
    hip.gemm(dy, L.Wt, out, a2=u, w2=L.At if L.r else None, residual=accum, beta=1.0 if accum is not None else 0.0)
    return out, u


def linear_bwd_lora(ctx: Ctx, L: LinearW, x, t, dy, u):
    """dB += dy^T t ;  dA += u^T x   (fp32, split-K atomics into the flat gradient buffer)."""
    if not L.r:
        return
    if ctx.dtype == torch.bfloat16:      # transpose-free kernel (LDS transpose reads), grouped per block
        ctx.queue_wgrad(dy, t, L.tB.grad)
        ctx.queue_wgrad(u, x, L.tA.grad)
        return
    hip.weight_grad(ctx.transposed(dy), ctx.transposed(t), L.tB.grad)
    hip.weight_grad(ctx.transposed(u), ctx.transposed(x), L.tA.grad)


def base_bwd_weight(ctx: Ctx, L: LinearW, x, dy):
    """Base-ControlNet pre-training: dW += dy^T x, db += colsum(dy) of a linear whose dense weight trains (no-op for
    frozen weights).  The zero convs keep their own call (dense_bwd_weight, with the control scale)."""
    if L.tW is None:
        return
    dense_bwd_weight(ctx, L, x, dy, 1, dy.shape[0], 1.0)


def conv3_bwd_weight(ctx: Ctx, cw: Conv3W, x, dy, B, Hin, Win, mode=hip.CONV_S1):
    """dW[o][tap][i] += sum_m dy[m, o] x[pixel(m, tap), i] for the nine taps, db += colsum(dy)  (trainable convs only).
    x: the conv's NHWC input [B*Hin*Win, Ip]; dy on the output grid."""
    if cw.tW is None:
        return
    stride = 2 if mode == hip.CONV_S2 else 1
    assert mode in (hip.CONV_S1, hip.CONV_S2)
    Ho, Wo = Hin // stride, Win // stride
    g = cw.tW.grad.view(cw.O, 9 * cw.Ip)
    if WGRAD_ROW3 and ctx.dtype == torch.bfloat16 and stride == 1 and (B * Hin * Win) % 32 == 0 and Win % 64 == 0:
        # the three taps of a kernel row as ONE problem (cl_wgrad_desc.tap = 16 + ky): dy is read three times instead of nine,
        # the shifted x tiles of a row share their pixels (csrc/wgrad.hip: wgrad_row3_kernel).  Taken at the 64x64 level only:
        # 244 -> 144 us per 320 -> 320 conv there, parity at the 32x32 / 16x16 levels (152 -> 159, 147 -> 143 us) and 50 -> 57 us
        # at the 8x8 level, where one 8-wave workgroup per CU has nothing over three 4-wave ones (tools/time_wgrad_row3.py,
        # profiles/r06_row3/)
        for ky in range(3):
            ctx.queue_wgrad(dy, x, g[:, 3 * ky * cw.Ip:(3 * ky + 1) * cw.Ip], 1.0, conv=(16 + ky, Hin, Win, Ho, Wo, 1, 1))
        hip.colsum(dy, cw.tb.grad.view(1, cw.O), 1, dy.shape[0], 1.0)
        return
    for t in range(9):
        gs = g[:, t * cw.Ip:(t + 1) * cw.Ip]
        if ctx.dtype == torch.bfloat16:
            ctx.queue_wgrad(dy, x, gs, 1.0, conv=(t, Hin, Win, Ho, Wo, stride, 1))
        else:     # fp32 parity mode: materialise the shifted operand, explicit transposes
            xs = ctx.new(B * Ho * Wo, cw.Ip)
            hip.conv_tap_gather(x, xs, B, Hin, Win, Ho, Wo, t, stride, 1)
            hip.weight_grad(ctx.transposed(dy), ctx.transposed(xs), gs)
    hip.colsum(dy, cw.tb.grad.view(1, cw.O), 1, dy.shape[0], 1.0)


def dense_bwd_weight(ctx: Ctx, L: LinearW, x, dy, B: int, HW: int, scale: float = 1.0):
    """Trainable dense 1x1 conv (zero convs): dW += scale * dy^T x ; db += scale * colsum(dy)."""
    if ctx.dtype == torch.bfloat16:
        ctx.queue_wgrad(dy, x, L.tW.grad.view(L.N, L.K), scale)
    else:
        hip.weight_grad(ctx.transposed(dy), ctx.transposed(x), L.tW.grad.view(L.N, L.K), scale)
    if L.tb is not None:
        ctx.queue_bias_grad(dy, L.tb.grad.view(1, L.N), B * HW, scale)


# --------------------------------------------------------------------------- norms

class GroupNormOp:
    def __init__(self, w: NormW, eps: float, silu: bool):
        self.w, self.eps, self.silu = w, eps, silu

    def fwd(self, ctx: Ctx, x, B: int, HW: int, out=None):
        C = x.shape[1]
        if out is None:
            out = ctx.new(x.shape[0], C)
        stats = torch.empty((B, 32, 2), dtype=torch.float32, device=ctx.device)
        hip.groupnorm_fwd(x, out, self.w.gamma, self.w.beta, B, HW, self.eps, self.silu, stats, ctx.gn_ws(B, HW, C))
        return out, stats

    def bwd(self, ctx: Ctx, x, dy, stats, B: int, HW: int, accum=None, out=None):
        C = x.shape[1]
        if out is None:
            out = ctx.new(x.shape[0], C)
        hip.groupnorm_bwd(x, dy, out, self.w.gamma, self.w.beta, stats, B, HW, self.silu, ctx.gn_ws(B, HW, C),
                          accum=accum, dgamma=self.w.ggamma, dbeta=self.w.gbeta)
        return out


class LayerNormOp:
    def __init__(self, w: NormW, eps: float = 1e-5):
        self.w, self.eps = w, eps

    def fwd(self, ctx: Ctx, x):
        out = ctx.new(*x.shape)
        stats = torch.empty((x.shape[0], 2), dtype=torch.float32, device=ctx.device) if ctx.record else None
        hip.layernorm_fwd(x, out, self.w.gamma, self.w.beta, self.eps, stats)
        return out, stats

    def prologue(self, ctx: Ctx, rows: int):
        """(gamma, beta, eps, stats) for a product that normalises its own input rows (hip.gemm(ln=...)); stats [rows, 2]
        (mean, rstd) only when a backward pass will want them."""
        stats = torch.empty((rows, 2), dtype=torch.float32, device=ctx.device) if ctx.record else None
        return (self.w.gamma, self.w.beta, self.eps, stats)

    def bwd(self, ctx: Ctx, x, dy, stats, accum=None):
        out = ctx.new(*x.shape)
        hip.layernorm_bwd(x, dy, out, self.w.gamma, stats, accum=accum, dgamma=self.w.ggamma, dbeta=self.w.gbeta)
        return out


# --------------------------------------------------------------------------- conv helpers

# Phase-decomposed Upsample conv / Downsample data gradient (hip.CONV_UP2P / CONV_T2P: 2.25x / 4x fewer MACs, same results):
# frozen convs whose channels are whole 128-byte lines and whose source grid fills 128-row tiles.  CTRLORA_CONV_PHASE=0: A/B switch.
CONV_PHASE = os.environ.get("CTRLORA_CONV_PHASE", "1") != "0"
# pre-training's 3x3 weight gradients: one problem per kernel ROW (three taps) instead of one per tap; =0: nine single-tap problems
WGRAD_ROW3 = os.environ.get("CTRLORA_WGRAD_ROW3", "1") != "0"


def _phase_ok(ctx: Ctx, cw: Conv3W, B, Hs, Ws, k1, n, t2=False):
    kq = 64 if ctx.dtype == torch.bfloat16 else 32
    ok = CONV_PHASE and cw.tW is None and (B * Hs * Ws) % 128 == 0 and k1 % kq == 0 and n >= 96
    if ok and t2:
        # the data-gradient form has phases 1 / 2 / 2 / 4 taps deep and therefore no K split: it needs the tile grid alone to
        # fill the chip (8 x 8 x 1280 at batch 8: 128 tiles, 95 us against the nine-tap mode's split-K 83 us -- profiles/r06_phase/)
        ok = (4 * B * Hs * Ws // 128) * ((n + 159) // 160) >= 192
    return bool(ok)


def conv3_fwd(ctx: Ctx, cw: Conv3W, x, B, Hin, Win, mode=hip.CONV_S1, out=None, rowbias=None, residual=None,
              out_f32=False):
    if mode == hip.CONV_UP2 and _phase_ok(ctx, cw, B, Hin, Win, cw.Ip, cw.Op):
        M = 4 * B * Hin * Win
        if out is None:
            out = ctx.new(M, cw.Op, torch.float32 if out_f32 else None)
        hip.gemm(x, cw.phase_weights("up2"), out, bias=cw.bias, rowbias=rowbias, rows_per_batch=4 * Hin * Win, residual=residual,
                 beta=1.0 if residual is not None else 0.0, mode=hip.CONV_UP2P, conv=(B, Hin, Win, 2 * Hin, 2 * Win), k1=cw.Ip,
                 out_f32=out_f32, N=cw.Op)
        return out
    if mode in (hip.CONV_S2, hip.CONV_S2A):
        Ho, Wo = Hin // 2, Win // 2
    elif mode == hip.CONV_UP2:
        Ho, Wo = 2 * Hin, 2 * Win
    else:
        Ho, Wo = Hin, Win
    M = B * Ho * Wo
    if out is None:
        out = ctx.new(M, cw.Op, torch.float32 if out_f32 else None)
    hip.gemm(x, cw.Wp, out, bias=cw.bias, rowbias=rowbias, rows_per_batch=Ho * Wo, residual=residual,
             beta=1.0 if residual is not None else 0.0, mode=mode, conv=(B, Hin, Win, Ho, Wo), k1=cw.Ip,
             out_f32=out_f32, N=cw.Op)
    return out


def conv3_bwd_data(ctx: Ctx, cw: Conv3W, dy, B, Hdy, Wdy, fwd_mode=hip.CONV_S1, out=None, accum=None):
    """Data gradient of a 3x3 conv given dy on the conv's OUTPUT grid (Hdy x Wdy)."""
    if fwd_mode == hip.CONV_S1:
        M = B * Hdy * Wdy
        if out is None:
            out = ctx.new(M, cw.Ip)
        hip.gemm(dy, cw.Wd, out, residual=accum, beta=1.0 if accum is not None else 0.0, mode=hip.CONV_S1,
                 conv=(B, Hdy, Wdy, Hdy, Wdy), k1=cw.Op, N=cw.Ip)
        return out
    if fwd_mode == hip.CONV_S2:      # dx lives on the 2x grid: conv over the zero-stuffed dy
        M = B * 4 * Hdy * Wdy
        if out is None:
            out = ctx.new(M, cw.Ip)
        if _phase_ok(ctx, cw, B, Hdy, Wdy, cw.Op, cw.Ip, t2=True):     # only the taps whose zero-stuffed input is non-zero
            hip.gemm(dy, cw.phase_weights("t2"), out, residual=accum, beta=1.0 if accum is not None else 0.0,
                     mode=hip.CONV_T2P, conv=(B, Hdy, Wdy, 2 * Hdy, 2 * Wdy), k1=cw.Op, N=cw.Ip)
            return out
        hip.gemm(dy, cw.Wd, out, residual=accum, beta=1.0 if accum is not None else 0.0, mode=hip.CONV_T2,
                 conv=(B, Hdy, Wdy, 2 * Hdy, 2 * Wdy), k1=cw.Op, N=cw.Ip)
        return out
    if fwd_mode == hip.CONV_UP2 and _phase_ok(ctx, cw, B, Hdy // 2, Wdy // 2, cw.Op, cw.Ip) and Hdy % 2 == 0 and Wdy % 2 == 0:
        # the gradient formed on the SOURCE grid: a 4x4 stride-2 window over dy with the coincident taps summed (16 K deep at
        # M / 4 rows instead of 9 K at M rows + a pooling pass)
        if out is None:
            out = ctx.new(B * Hdy * Wdy // 4, cw.Ip)
        hip.gemm(dy, cw.phase_weights("up2d"), out, residual=accum, beta=1.0 if accum is not None else 0.0, mode=hip.CONV_S2K4,
                 conv=(B, Hdy, Wdy, Hdy // 2, Wdy // 2), k1=cw.Op, N=cw.Ip)
        return out
    if fwd_mode == hip.CONV_UP2:     # dy on the upsampled grid -> stride-1 data gradient -> 2x2 sum pool
        M = B * Hdy * Wdy
        dup = ctx.new(M, cw.Ip)
        hip.gemm(dy, cw.Wd, dup, mode=hip.CONV_S1, conv=(B, Hdy, Wdy, Hdy, Wdy), k1=cw.Op, N=cw.Ip)
        if out is None:
            out = ctx.new(M // 4, cw.Ip)
        hip.pool2x2(dup, out, B, Hdy // 2, Wdy // 2, accumulate=False)
        return out
    raise ValueError(fwd_mode)


# --------------------------------------------------------------------------- ResBlock

class ResBlockE:
    def __init__(self, gn1: NormW, conv1: Conv3W, emb: LinearW, gn2: NormW, conv2: Conv3W,
                 skip: Optional[LinearW]):
        self.gn1 = GroupNormOp(gn1, 1e-5, True)
        self.gn2 = GroupNormOp(gn2, 1e-5, True)
        self.conv1, self.conv2, self.emb, self.skip = conv1, conv2, emb, skip
        self.cin, self.cout = conv1.I, conv1.O

    def fwd(self, ctx: Ctx, x, semb, B, H, W, out=None, e_pre=None):
        HW = H * W
        h1, st1 = self.gn1.fwd(ctx, x, B, HW)
        if e_pre is not None:                                             # formed with all the others at the start of the pass
            e_out, t_e = e_pre if isinstance(e_pre, tuple) else (e_pre, None)
        else:
            e_out, t_e = linear_fwd(ctx, self.emb, semb)                  # [B, cout]
        h2 = conv3_fwd(ctx, self.conv1, h1, B, H, W, rowbias=e_out)       # + bias + emb (openaimodel.py:272)
        keep = ctx.record and self.conv1.tW is not None                   # conv inputs: operands of the conv dW
        if not keep:
            del h1
        h3, st2 = self.gn2.fwd(ctx, h2, B, HW)
        if self.skip is not None:
            out, _ = linear_fwd(ctx, self.skip, x, out=out)
            conv3_fwd(ctx, self.conv2, h3, B, H, W, out=out, residual=out)  # skip(x) + h  (:274)
        else:
            out = conv3_fwd(ctx, self.conv2, h3, B, H, W, out=out, residual=x)
        saved = (x, st1, h2, st2, semb, t_e, (h1, h3) if keep else None) if ctx.record else None
        return out, saved

    def bwd(self, ctx: Ctx, dout, saved, B, H, W, dsemb=None, need_emb_grads=False, out=None, de_slot=None):
        x, st1, h2, st2, semb, t_e, conv_in = saved
        HW = H * W
        if conv_in is not None:
            conv3_bwd_weight(ctx, self.conv2, conv_in[1], dout, B, H, W)
        dh3 = conv3_bwd_data(ctx, self.conv2, dout, B, H, W)
        dh2 = self.gn2.bwd(ctx, h2, dh3, st2, B, HW)
        del dh3
        if need_emb_grads and de_slot is not None:
            # d emb_out[b, c] = sum_p dh2[b, p, c] into this block's slice; the linear's backward runs grouped, after the trunk
            hip.colsum(dh2, de_slot, B, HW)
        elif need_emb_grads:
            # d emb_out[b, c] = sum_p dh2[b, p, c]; back through the LoRA'd emb linear into d silu(emb)
            de32 = ctx.zeros(B, self.cout, torch.float32)
            hip.colsum(dh2, de32, B, HW)
            de = ctx.new(B, self.cout)
            hip.pack2d(de32, de)
            _, u = linear_bwd_data(ctx, self.emb, de, out=dsemb, accum=dsemb)
            linear_bwd_lora(ctx, self.emb, semb, t_e, de, u)
            base_bwd_weight(ctx, self.emb, semb, de)
        if conv_in is not None:
            conv3_bwd_weight(ctx, self.conv1, conv_in[0], dh2, B, H, W)
        dh1 = conv3_bwd_data(ctx, self.conv1, dh2, B, H, W)
        del dh2
        if self.skip is not None:
            base_bwd_weight(ctx, self.skip, x, dout)
            dskip, _ = linear_bwd_data(ctx, self.skip, dout)
        else:
            dskip = dout
        dx = self.gn1.bwd(ctx, x, dh1, st1, B, HW, accum=dskip, out=out)
        ctx.drop_transposes()
        return dx


# --------------------------------------------------------------------------- attention

class AttnE:
    """CrossAttention.  `fused` = [q|k|v] (self) or [k|v] (cross) weights concatenated along N,
    used when there is no LoRA (frozen UNet): one GEMM instead of three."""

    def __init__(self, to_q: LinearW, to_k: LinearW, to_v: LinearW, to_out: LinearW, heads: int, is_self: bool,
                 fused_qkv: Optional[LinearW] = None, fused_kv: Optional[LinearW] = None, need_kv_grad=True, group=None):
        self.q, self.k, self.v, self.o = to_q, to_k, to_v, to_out
        self.group = group       # packing.LoraGroup over (q, k, v) [self] or (k, v) [cross]: grouped LoRA launches
        self.heads, self.is_self = heads, is_self
        self.fused_qkv, self.fused_kv = fused_qkv, fused_kv
        self.need_kv_grad = need_kv_grad
        self.inner = to_q.N
        self.dh = self.inner // heads
        self.scale = float(self.dh) ** -0.5
        # Pre-scaled-Q contract of the bf16 attention kernels (CL_ATTN_Q_PRESCALED): the to_q projection's epilogue writes
        # q * d_head^-0.5 * log2(e) -- one rounding from its fp32 accumulators, like any stored q -- so the kernels take
        # log2-domain scores straight off the matrix product; dQ / dK / dV stay the gradients of the TRUE q, k, v, so the
        # projections' backward is unchanged.  q is internal to the block (attention.py:163-171: only the attention reads it).
        self.q_alpha = self.scale * 1.4426950408889634

    def _prescaled(self, ctx: Ctx) -> bool:
        return PRESCALE_Q and ctx.dtype == torch.bfloat16

    def _group_fwd(self, ctx: Ctx, x, alpha=1.0, alpha_n=0, ln=None):
        return group_fwd(ctx, self.group, x, alpha=alpha, alpha_n=alpha_n, ln=ln)

    def ln_fusable(self, ctx: Ctx, M: int) -> bool:
        """Can the LayerNorm in front of this attention be the prologue of the product that reads it (q | k | v of a
        self-attention, to_q of a cross-attention)?"""
        # "live": the normalised tensor is an operand of something else too -- t = LN(x) A^T of an unmerged LoRA, or the
        # weight gradient of a dense weight that trains (pre-training / ft_with_lora = False)
        live = lambda L: (bool(L.r) and not (L.Wm is not None and not ctx.record)) or (ctx.record and L.tW is not None)
        if self.is_self:
            if self.fused_qkv is not None:
                return ln_prologue_ok(ctx, M, 3 * self.inner, self.fused_qkv.K, live(self.fused_qkv))
            if self.group is not None:
                return ln_prologue_ok(ctx, M, 3 * self.inner, self.group.K, not (self.group.Wm is not None and not ctx.record)
                                      or any(ctx.record and L.tW is not None for L in self.group.members))
            return False          # three separate products would each normalise the rows again
        return ln_prologue_ok(ctx, M, self.inner, self.q.K, live(self.q))

    def _group_bwd(self, ctx: Ctx, dy, need_dx: bool, accum=None):
        """dy [M, G N] -> (dx [M, K] (+ accum) or None, u [M, G r])."""
        grp = self.group
        M = dy.shape[0]
        u = ctx.new(M, grp.G * grp.r)
        if grp.r % 64 == 0:
            hip.gemm(dy, grp.Bt, u, k1=grp.N, a1_group_n=grp.r)
        else:                                 # rank below the narrowest tile: one small product per member
            for g, L in enumerate(grp.members):
                hip.gemm(dy[:, g * grp.N:(g + 1) * grp.N], L.Bt, u[:, g * grp.r:(g + 1) * grp.r])
        dx = None
        if need_dx:
            dx = ctx.new(M, grp.K)
            hip.gemm(dy, grp.Wt, dx, a2=u, w2=grp.At, residual=accum, beta=1.0 if accum is not None else 0.0)
        return dx, u

    def project_context(self, ctx: Ctx, c):
        """K / V projections of the text context (identical for every denoising step)."""
        if self.fused_kv is not None:
            kv, _ = linear_fwd(ctx, self.fused_kv, c)
            return kv[:, :self.inner], kv[:, self.inner:], None, None
        if self.group is not None and not self.is_self:
            kv, t = self._group_fwd(ctx, c)
            r = self.group.r
            return (kv[:, :self.inner], kv[:, self.inner:], None if t is None else t[:, :r], None if t is None else t[:, r:])
        k, tk = linear_fwd(ctx, self.k, c)
        v, tv = linear_fwd(ctx, self.v, c)
        return k, v, tk, tv

    def fwd(self, ctx: Ctx, xn, c, B, N, Nkv, residual, kv_cache=None, ln=None):
        """ln = LayerNormOp.prologue(...): xn is the UN-normalised input and the product that reads it applies the norm
        (only where ln_fusable() said so)."""
        inner, H = self.inner, self.heads
        tq = tk = tv = None
        pre = self._prescaled(ctx)
        qa = self.q_alpha if pre else 1.0
        if self.is_self:
            if self.fused_qkv is not None:
                qkv, _ = linear_fwd(ctx, self.fused_qkv, xn, alpha=qa, alpha_n=inner if pre else 0, ln=ln)
                q, k, v = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
            elif self.group is not None:
                qkv, t = self._group_fwd(ctx, xn, alpha=qa, alpha_n=inner if pre else 0, ln=ln)
                q, k, v = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
                if t is not None:
                    r = self.group.r
                    tq, tk, tv = t[:, :r], t[:, r:2 * r], t[:, 2 * r:]
            else:
                assert ln is None
                q, tq = linear_fwd(ctx, self.q, xn, alpha=qa)
                k, tk = linear_fwd(ctx, self.k, xn)
                v, tv = linear_fwd(ctx, self.v, xn)
        else:
            q, tq = linear_fwd(ctx, self.q, xn, alpha=qa, ln=ln)
            if kv_cache is not None:
                k, v, tk, tv = kv_cache
            else:
                k, v, tk, tv = self.project_context(ctx, c)
        a = ctx.new(B * N, inner)
        lse = torch.empty((B, H, rup(N, 64)), dtype=torch.float32, device=ctx.device) if ctx.record else None
        if ctx.dtype == torch.bfloat16:       # transpose-free kernels (LDS transpose reads)
            hip.attention_fwd_v2(q, k, v, a, lse, B, H, N, Nkv, self.dh, self.scale, q_prescaled=pre)
        else:
            kpad = rup(Nkv, 64)
            vt = torch.empty((B, inner, kpad), dtype=ctx.dtype, device=ctx.device)
            hip.transpose(v, vt, B, Nkv, inner, kpad, ldi=v.stride(0))
            hip.attention_fwd(q, k, vt, a, lse, B, H, N, Nkv, self.dh, self.scale)
            del vt
        out, to_ = linear_fwd(ctx, self.o, a, residual=residual)
        saved = (xn, c, q, k, v, a, lse, tq, tk, tv, to_) if ctx.record else None
        return out, saved

    def bwd(self, ctx: Ctx, dout, saved, B, N, Nkv, accum_xn=None):
        """dout: gradient of the block output (residual path handled by the caller).
        Returns d(xn) (+ accum_xn)."""
        xn, c, q, k, v, a, lse, tq, tk, tv, to_ = saved
        inner, H = self.inner, self.heads
        da, uo = linear_bwd_data(ctx, self.o, dout)
        linear_bwd_lora(ctx, self.o, a, to_, dout, uo)
        base_bwd_weight(ctx, self.o, a, dout)
        delta = torch.empty_like(lse)
        want_kv = self.is_self or self.need_kv_grad
        grouped = self.group is not None and ctx.dtype == torch.bfloat16
        dkv = None
        if self.is_self and (self.fused_qkv is not None or grouped):
            dqkv = ctx.new(B * N, 3 * inner)
            dq, dk, dv = dqkv[:, :inner], dqkv[:, inner:2 * inner], dqkv[:, 2 * inner:]
        else:
            dq = ctx.new(B * N, inner)
            if want_kv and grouped:
                dkv = ctx.new(B * Nkv, 2 * inner)
                dk, dv = dkv[:, :inner], dkv[:, inner:]
            else:
                dk = ctx.new(B * Nkv, inner) if want_kv else None
                dv = ctx.new(B * Nkv, inner) if want_kv else None
        if ctx.dtype == torch.bfloat16:
            pre = self._prescaled(ctx)
            # (-lse, -delta) of every query row as bf16 triples: lets the d_head-40 kernels fold them into their products
            row_ws = torch.empty(lse.numel() * 8, dtype=torch.float32, device=ctx.device) if (pre and self.dh == 40) else None
            hip.attention_bwd_v2(q, k, v, a, da, lse, delta, dq, dk, dv, B, H, N, Nkv, self.dh, self.scale,
                                 q_prescaled=pre, row_ws=row_ws)
        else:
            npad, kpad = rup(N, 64), rup(Nkv, 64)
            qt = torch.empty((B, inner, npad), dtype=ctx.dtype, device=ctx.device)
            dot = torch.empty((B, inner, npad), dtype=ctx.dtype, device=ctx.device)
            kt = torch.empty((B, inner, kpad), dtype=ctx.dtype, device=ctx.device)
            hip.transpose(q, qt, B, N, inner, npad, ldi=q.stride(0))
            hip.transpose(da, dot, B, N, inner, npad, ldi=da.stride(0))
            hip.transpose(k, kt, B, Nkv, inner, kpad, ldi=k.stride(0))
            hip.attention_bwd(q, k, v, a, da, qt, dot, kt, lse, delta, dq, dk, dv, B, H, N, Nkv, self.dh, self.scale)
            del qt, dot, kt
        if self.is_self:
            if self.fused_qkv is not None:
                dxn, _ = linear_bwd_data(ctx, self.fused_qkv, dqkv, accum=accum_xn)
                return dxn
            if grouped:
                dxn, u = self._group_bwd(ctx, dqkv, True, accum=accum_xn)
                r = self.group.r
                for i, (L, d, t_) in enumerate(((self.q, dq, tq), (self.k, dk, tk), (self.v, dv, tv))):
                    linear_bwd_lora(ctx, L, xn, t_, d, u[:, i * r:(i + 1) * r])
                    base_bwd_weight(ctx, L, xn, d)
                return dxn
            dxn, uq = linear_bwd_data(ctx, self.q, dq, accum=accum_xn)
            linear_bwd_lora(ctx, self.q, xn, tq, dq, uq)
            _, uk = linear_bwd_data(ctx, self.k, dk, out=dxn, accum=dxn)
            linear_bwd_lora(ctx, self.k, xn, tk, dk, uk)
            _, uv = linear_bwd_data(ctx, self.v, dv, out=dxn, accum=dxn)
            linear_bwd_lora(ctx, self.v, xn, tv, dv, uv)
            for L, d in ((self.q, dq), (self.k, dk), (self.v, dv)):
                base_bwd_weight(ctx, L, xn, d)
            return dxn
        dxn, uq = linear_bwd_data(ctx, self.q, dq, accum=accum_xn)
        linear_bwd_lora(ctx, self.q, xn, tq, dq, uq)
        base_bwd_weight(ctx, self.q, xn, dq)
        if want_kv and self.k.r and dkv is not None:
            _, u = self._group_bwd(ctx, dkv, False)
            r = self.group.r
            linear_bwd_lora(ctx, self.k, c, tk, dk, u[:, :r])
            linear_bwd_lora(ctx, self.v, c, tv, dv, u[:, r:])
        elif want_kv and self.k.r:
            # context is an input (no data gradient needed), only the LoRA factors of to_k / to_v train
            uk = ctx.new(B * Nkv, self.k.r); hip.gemm(dk, self.k.Bt, uk)
            linear_bwd_lora(ctx, self.k, c, tk, dk, uk)
            uv = ctx.new(B * Nkv, self.v.r); hip.gemm(dv, self.v.Bt, uv)
            linear_bwd_lora(ctx, self.v, c, tv, dv, uv)
        if want_kv:
            base_bwd_weight(ctx, self.k, c, dk)
            base_bwd_weight(ctx, self.v, c, dv)
        return dxn


# --------------------------------------------------------------------------- SpatialTransformer

class SpatialTransformerE:
    def __init__(self, norm: NormW, proj_in: LinearW, ln1: NormW, attn1: AttnE, ln2: NormW, attn2: AttnE,
                 ln3: NormW, ff_proj: LinearW, ff_out: LinearW, proj_out: LinearW):
        self.norm = GroupNormOp(norm, 1e-6, False)
        self.proj_in, self.proj_out = proj_in, proj_out
        self.ln1, self.ln2, self.ln3 = LayerNormOp(ln1), LayerNormOp(ln2), LayerNormOp(ln3)
        self.attn1, self.attn2 = attn1, attn2
        self.ff_proj, self.ff_out = ff_proj, ff_out
        self.C = proj_in.N

    def fwd(self, ctx: Ctx, x, c, B, H, W, Nkv, out=None, kv_cache=None):
        N = H * W
        xn, st0 = self.norm.fwd(ctx, x, B, N)
        h0, _ = linear_fwd(ctx, self.proj_in, xn)                    # 1x1 conv == per-token linear
        if not (ctx.record and self.proj_in.tW is not None):         # operand of proj_in's dW when it trains
            xn = None
        # norm1 / norm2 / norm3 (attention.py:271-275) run as the PROLOGUE of the product that reads them wherever nothing else
        # needs the normalised tensor (ln_prologue_ok): then n_i below is the un-normalised input and s_i the statistics the
        # product wrote for the backward pass
        M = B * N
        if self.attn1.ln_fusable(ctx, M):
            l1 = self.ln1.prologue(ctx, M)
            n1, s1 = h0, l1[3]
        else:
            l1 = None
            n1, s1 = self.ln1.fwd(ctx, h0)
        h1, sv1 = self.attn1.fwd(ctx, n1, None, B, N, N, residual=h0, ln=l1)
        if self.attn2.ln_fusable(ctx, M):
            l2 = self.ln2.prologue(ctx, M)
            n2, s2 = h1, l2[3]
        else:
            l2 = None
            n2, s2 = self.ln2.fwd(ctx, h1)
        h2, sv2 = self.attn2.fwd(ctx, n2, c, B, N, Nkv, residual=h1, kv_cache=kv_cache, ln=l2)
        L = self.ff_proj
        ff_live = (bool(L.r) and not (L.Wm is not None and not ctx.record)) or (ctx.record and L.tW is not None)
        xs_geglu = (not ctx.record and ctx.dtype == torch.bfloat16 and L.N % 64 == 0
                    and hip.xs_geglu_ok(M, L.K, 0 if L.Wm is not None else L.r))
        tile_geglu = not xs_geglu and not ctx.record and L.geglu_ok()
        if not tile_geglu and ln_prologue_ok(ctx, M, L.N, L.K, ff_live, hip.ACT_GEGLU_SPLIT if xs_geglu else hip.ACT_NONE):
            l3 = self.ln3.prologue(ctx, M)
            n3, s3 = h2, l3[3]
        else:
            l3 = None
            n3, s3 = self.ln3.fwd(ctx, h2)
        if xs_geglu:
            # the same fusion on the x-stationary kernel (csrc/gemm_xs.hip): W's rows in their natural [value | gate] order
            tp = None
            if L.r and L.Wm is None:
                tp = ctx.new(B * N, L.r)
                hip.gemm(n3, L.A, tp)
            gg = ctx.new(B * N, 4 * self.C)
            hip.gemm(n3, L.W if L.Wm is None else L.Wm, gg, a2=tp, w2=L.B if tp is not None else None, bias=L.bias,
                     act=hip.ACT_GEGLU_SPLIT, N=L.N, ln=l3)
            p = None
        elif tile_geglu:
            # no backward will need the 8C-wide pre-activation: value * gelu(gate) is formed in the projection's
            # epilogue (half the output bytes, no separate GEGLU pass)
            Wg, bg, Bg = L.geglu_pack()
            tp = None
            if Bg is not None:
                tp = ctx.new(B * N, L.r)
                hip.gemm(n3, L.A, tp)
            gg = ctx.new(B * N, 4 * self.C)
            hip.gemm(n3, Wg, gg, a2=tp, w2=Bg, bias=bg, act=hip.ACT_GEGLU, N=L.N)
            p = None
        else:
            p, tp = linear_fwd(ctx, self.ff_proj, n3, ln=l3)         # [M, 8C]
            gg = ctx.new(B * N, 4 * self.C)
            hip.geglu_fwd(p, gg)
        h3, tf = linear_fwd(ctx, self.ff_out, gg, residual=h2)
        out, _ = linear_fwd(ctx, self.proj_out, h3, out=out, residual=x)
        saved = (x, st0, h0, s1, sv1, h1, s2, sv2, h2, s3, n3, p, tp, gg, tf, h3, xn) if ctx.record else None
        return out, saved

    def bwd(self, ctx: Ctx, dout, saved, B, H, W, Nkv, out=None):
        x, st0, h0, s1, sv1, h1, s2, sv2, h2, s3, n3, p, tp, gg, tf, h3, xn = saved
        N = H * W
        base_bwd_weight(ctx, self.proj_out, h3, dout)
        dh, _ = linear_bwd_data(ctx, self.proj_out, dout)           # d h3 ; (x_in residual: + dout at the end)
        # feed-forward
        dgg, uf = linear_bwd_data(ctx, self.ff_out, dh)
        linear_bwd_lora(ctx, self.ff_out, gg, tf, dh, uf)
        base_bwd_weight(ctx, self.ff_out, gg, dh)
        dp = ctx.new(B * N, 8 * self.C)
        hip.geglu_bwd(p, dgg, dp)
        del dgg
        dn3, up = linear_bwd_data(ctx, self.ff_proj, dp)
        linear_bwd_lora(ctx, self.ff_proj, n3, tp, dp, up)
        base_bwd_weight(ctx, self.ff_proj, n3, dp)
        del dp
        dh = self.ln3.bwd(ctx, h2, dn3, s3, accum=dh)                # d h2
        ctx.drop_transposes()
        # cross attention
        dn2 = self.attn2.bwd(ctx, dh, sv2, B, N, Nkv)
        dh = self.ln2.bwd(ctx, h1, dn2, s2, accum=dh)                # d h1
        ctx.drop_transposes()
        # self attention
        dn1 = self.attn1.bwd(ctx, dh, sv1, B, N, N)
        dh = self.ln1.bwd(ctx, h0, dn1, s1, accum=dh)                # d h0
        ctx.drop_transposes()
        if xn is not None:
            base_bwd_weight(ctx, self.proj_in, xn, dh)
        dxn, _ = linear_bwd_data(ctx, self.proj_in, dh)
        dx = self.norm.bwd(ctx, x, dxn, st0, B, N, accum=dout, out=out)
        return dx
