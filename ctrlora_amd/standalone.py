"""Stand-alone (outside a network) execution of LoRA linears on the HIP GEMM kernel, differentiable like the reference's
modules (cldm/lora.py:70-80, 285-291): every product of the forward AND of the backward is a cl_gemm launch.  Training at
speed runs in the engine (grouped / fused launches, packed operands kept across steps); this path packs its operands per call.
"""
import torch

from . import hip
from .engine.packing import rup


def _as_rows(x):
    lead = x.shape[:-1]
    return x.reshape(-1, x.shape[-1]), lead


def _prep(t, dtype, cols_pad=None):
    t = t.detach().to(dtype)
    if cols_pad is not None and t.shape[1] != cols_pad:
        t = torch.nn.functional.pad(t, (0, cols_pad - t.shape[1]))
    return t.contiguous()


def _kq(dtype):
    return 32 if dtype == torch.bfloat16 else 16


def _mm_nt(a, b, dtype, a2=None, b2=None, bias=None):
    """a[m, k] . b[n, k]^T (+ a2[m, k2] . b2[n, k2]^T + bias) on the GEMM kernel; K padded to the kernel's quantum and N
    to 8 with zeros; returns [m, n] in `dtype`."""
    m, k = a.shape
    n = b.shape[0]
    kp, np_ = rup(k, _kq(dtype)), rup(n, 8)
    ap = _prep(a, dtype, kp)
    bp = _prep(torch.nn.functional.pad(b.detach(), (0, 0, 0, np_ - n)), dtype, kp)
    out = torch.empty((m, np_), dtype=dtype, device=a.device)
    a2p = b2p = None
    if a2 is not None:
        k2p = rup(a2.shape[1], _kq(dtype))
        a2p = _prep(a2, dtype, k2p)
        b2p = _prep(torch.nn.functional.pad(b2.detach(), (0, 0, 0, np_ - n)), dtype, k2p)
    bp32 = None if bias is None else torch.nn.functional.pad(bias.detach().float(), (0, np_ - n)).contiguous()
    hip.gemm(ap, bp, out, a2=a2p, w2=b2p, bias=bp32)
    return out[:, :n]


class _LoraLinearFn(torch.autograd.Function):
    """y = x W^T + b + s (x A^T) B^T with the gradients of every input the reference's autograd would produce:
        u  = s dy B            dx = dy W + u A        dW = dy^T x       db = sum_rows dy
        dA = u^T x             dB = s dy^T (x A^T)"""

    @staticmethod
    def forward(ctx, x2, W, bias, A, B, scale, dtype):
        t = None
        if A is not None:
            t = _mm_nt(x2, A, dtype)                                      # [M, r]
            y = _mm_nt(x2, W, dtype, a2=t, b2=B.detach() * scale, bias=bias)
        else:
            y = _mm_nt(x2, W, dtype, bias=bias)
        ctx.save_for_backward(x2, W, A, B, t)
        ctx.scale, ctx.dtype, ctx.bias_dtype = scale, dtype, None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, W, A, B, t = ctx.saved_tensors
        s, dtype = ctx.scale, ctx.dtype
        need = ctx.needs_input_grad                                      # x2, W, bias, A, B
        dy = dy.contiguous()
        dx = dW = db = dA = dB = None
        u = None
        if A is not None and (need[0] or need[3]):
            u = _mm_nt(dy, (B.detach() * s).t(), dtype)                   # [M, r] = s dy B
        if need[0]:
            dx = (_mm_nt(dy, W.detach().t(), dtype, a2=u, b2=A.detach().t()) if u is not None
                  else _mm_nt(dy, W.detach().t(), dtype)).to(x2.dtype)
        dyT = dy.t() if (need[1] or (A is not None and need[4])) else None
        if need[1]:
            dW = _mm_nt(dyT, x2.detach().t(), dtype).to(W.dtype)          # [N, K]
        if ctx.bias_dtype is not None and need[2]:
            n8 = rup(dy.shape[1], 8)
            acc = torch.zeros((1, n8), dtype=torch.float32, device=dy.device)     # cl_colsum accumulates into its output
            hip.colsum(_prep(dy, dtype, n8), acc, 1, dy.shape[0])
            db = acc[0, :dy.shape[1]].to(ctx.bias_dtype)
        if A is not None and need[3]:
            dA = _mm_nt(u.t(), x2.detach().t(), dtype).to(A.dtype)        # [r, K]
        if A is not None and need[4]:
            dB = (_mm_nt(dyT, t.t(), dtype) * s).to(B.dtype)              # [N, r]
        return dx, dW, db, dA, dB, None, None


def lora_linear(x, W, bias, A, B, scale=1.0):
    """y = x W^T + b + scale * (x A^T) B^T on the GPU kernel; differentiable in x, W, b, A, B."""
    if not x.is_cuda:
        raise hip.HipError("LoRACompatibleLinear runs on the MI355X HIP kernels only (tensor is on CPU)")
    dtype = x.dtype if x.dtype in (torch.bfloat16, torch.float32) else torch.float32
    x2, lead = _as_rows(x)
    y = _LoraLinearFn.apply(x2, W, bias, A, B, float(scale), dtype)
    return y.reshape(*lead, W.shape[0]).to(x.dtype)


def lora_delta(x, A, B, scale=None):
    """up(down(x)) [* scale]  (LoRALinearLayer.forward, cldm/lora.py:70-80)."""
    zeros = torch.zeros((B.shape[0], A.shape[1]), dtype=A.dtype, device=A.device)
    return lora_linear(x, zeros, None, A, B, 1.0 if scale is None else scale)
