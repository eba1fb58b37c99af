"""Stand-alone (outside a network) execution of LoRA linears on the HIP GEMM kernel."""
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


@torch.no_grad()
def lora_linear(x, W, bias, A, B, scale=1.0):
    """y = x W^T + b + scale * (x A^T) B^T on the GPU kernel (inference use; training runs in the engine)."""
    if not x.is_cuda:
        raise hip.HipError("LoRACompatibleLinear runs on the MI355X HIP kernels only (tensor is on CPU)")
    dtype = x.dtype if x.dtype in (torch.bfloat16, torch.float32) else torch.float32
    x2, lead = _as_rows(x)
    kq = 32 if dtype == torch.bfloat16 else 16
    K, N = W.shape[1], W.shape[0]
    Kp, Np = rup(K, kq), rup(N, 8)
    xa = _prep(x2, dtype, Kp)
    Wp = _prep(torch.nn.functional.pad(W.detach(), (0, 0, 0, Np - N)), dtype, Kp)
    bp = None if bias is None else torch.nn.functional.pad(bias.detach().float(), (0, Np - N)).contiguous()
    out = torch.empty((xa.shape[0], Np), dtype=dtype, device=x.device)
    t = Bp = None
    if A is not None:
        r = A.shape[0]
        rp = rup(r, kq)
        Ap = _prep(torch.nn.functional.pad(A.detach(), (0, 0, 0, rp - r)), dtype, Kp)
        Bp = _prep(torch.nn.functional.pad(B.detach() * scale, (0, 0, 0, Np - N)), dtype, rp)
        t = torch.empty((xa.shape[0], rp), dtype=dtype, device=x.device)
        hip.gemm(xa, Ap, t)
    hip.gemm(xa, Wp, out, a2=t, w2=Bp, bias=bp)
    return out[:, :N].reshape(*lead, N).to(x.dtype)


@torch.no_grad()
def lora_delta(x, A, B, scale=None):
    """up(down(x)) [* scale]  (LoRALinearLayer.forward, cldm/lora.py:70-80)."""
    zeros = torch.zeros((B.shape[0], A.shape[1]), dtype=A.dtype, device=A.device)
    return lora_linear(x, zeros, None, A, B, 1.0 if scale is None else scale)
