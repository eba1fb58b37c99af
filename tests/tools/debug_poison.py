"""Debug aid: the small-M (M = batch) products of the emb_layers / time_embed backward with every operand carved out of a
poisoned arena (NaN or 1e30 on both sides of each operand).  A kernel that reads outside its operands -- and lets what it
read reach the result -- shows up as a mismatch against torch.  Not part of the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctrlora_amd import hip  # noqa: E402
from ctrlora_amd.engine.packing import rup  # noqa: E402

dev = "cuda"


class Arena:
    def __init__(self, poison, dtype):
        self.buf = torch.full((8 << 20,), poison, dtype=dtype, device=dev)
        self.off = 4096

    def take(self, rows, cols, fill=None):
        n = rows * cols
        t = self.buf[self.off:self.off + n].view(rows, cols)
        self.off += rup(n, 64) + 64 * 7            # a poisoned gap after every operand
        if fill is not None:
            t.copy_(fill)
        return t


def check(name, got, want, tol=1e-4):
    err = float((got.float() - want.float()).abs().max())
    ref = float(want.float().abs().max()) + 1e-12
    bad = (not torch.isfinite(got.float()).all()) or err > tol * ref
    print(f"{'BAD ' if bad else 'ok  '} {name}: err {err:.3e} ref {ref:.3e}", flush=True)
    return bad


nbad = 0
for dtype in (torch.float32, torch.bfloat16):
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    for poison in (float("nan"), 1e30):
        for (B, N, K, r) in ((2, 64, 256, 32), (2, 128, 256, 32), (2, 256, 256, 32), (2, 1280, 1280, 128), (4, 320, 1280, 128), (1, 640, 1280, 64)):
            torch.manual_seed(B * 1000 + N)
            ar = Arena(poison, dtype)
            f32 = Arena(poison, torch.float32)
            de = ar.take(B, N, torch.randn(B, N, device=dev))
            x = ar.take(B, K, torch.randn(B, K, device=dev))
            Wt = ar.take(N, K, torch.randn(N, K, device=dev) * 0.05)          # [N, K]: dy W
            Bt = ar.take(r, N, torch.randn(r, N, device=dev) * 0.05)          # u = dy Bt^T  (Bt [r, N])
            At = ar.take(K, r, torch.randn(K, r, device=dev) * 0.05)          # + u At^T (At [K, r])
            accum = ar.take(B, K, torch.randn(B, K, device=dev))
            tag = f"{str(dtype)[6:]} poison={poison} B{B} N{N} K{K} r{r}"
            # u = de B ; dx = de W + u A + accum
            u = ar.take(B, r)
            hip.gemm(de, Bt, u)
            nbad += check(tag + " u", u, de.float() @ Bt.float().t(), tol)
            dx = ar.take(B, K)
            WtT = ar.take(K, N, Wt.t().contiguous())
            hip.gemm(de, WtT, dx, a2=u, w2=At, residual=accum, beta=1.0)
            nbad += check(tag + " dx", dx, de.float() @ Wt.float() + u.float() @ At.float().t() + accum.float(), tol)
            if dtype == torch.float32:
                # weight gradients through explicit transposes (fp32 parity mode)
                Mp = rup(B, 32)
                t_e = ar.take(B, r, torch.randn(B, r, device=dev))
                deT = ar.take(N, Mp); hip.transpose(de, deT, 1, B, N, Mp)
                tT = ar.take(r, Mp); hip.transpose(t_e, tT, 1, B, r, Mp)
                uT = ar.take(r, Mp); hip.transpose(u, uT, 1, B, r, Mp)
                xT = ar.take(K, Mp); hip.transpose(x, xT, 1, B, K, Mp)
                nbad += check(tag + " deT", deT[:, :B], de.t())
                nbad += check(tag + " deT pad", deT[:, B:], torch.zeros_like(deT[:, B:]))
                gB = f32.take(N, r, torch.zeros(N, r, device=dev))
                gA = f32.take(r, K, torch.zeros(r, K, device=dev))
                hip.weight_grad(deT, tT, gB)
                hip.weight_grad(uT, xT, gA)
                nbad += check(tag + " dB", gB, de.t() @ t_e, tol)
                nbad += check(tag + " dA", gA, u.t() @ x, tol)
            # colsum of a [B*HW, N] gradient into [B, N] fp32, then pack
            HW = 16
            dh = ar.take(B * HW, N, torch.randn(B * HW, N, device=dev))
            s32 = f32.take(B, N, torch.zeros(B, N, device=dev))
            hip.colsum(dh, s32, B, HW)
            nbad += check(tag + " colsum", s32, dh.float().view(B, HW, N).sum(1), tol)
            pk = ar.take(B, N)
            hip.pack2d(s32, pk)
            nbad += check(tag + " pack", pk, s32, tol)
torch.cuda.synchronize()
print("bad checks:", nbad)
