"""Is attn_fwd40_kernel a pure function of its inputs?  (The replayed training step was not: tests/tools/debug_determinism.py.)
(1) synthetic inputs, contiguous and as column slices of a [M, 3 inner] buffer, with and without the spike that forces the second
pass, alone and next to a GEMM stream; (2) the q / k / v of every d_head-40 self-attention of one eager training forward at the
bench shape, each re-run REPS times.  Reports, per case, how many of the repeats differ bitwise from the first and by how much.
Debug aid, GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctrlora_amd import hip   # noqa: E402

REPS = 8
LOG2E = 1.4426950408889634


def repeat(tag, q, k, v, B, H, N, Nkv, dh, side=None):
    o = torch.empty(B * N, H * dh, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, (N + 63) // 64 * 64, dtype=torch.float32, device="cuda")
    outs = []
    for _ in range(REPS):
        o.fill_(float("nan")); lse.fill_(float("nan"))
        if side is not None:
            side()
        hip.attention_fwd_v2(q, k, v, o, lse, B, H, N, Nkv, dh, dh ** -0.5, q_prescaled=True)
        torch.cuda.synchronize()
        outs.append((o.clone(), lse.clone()))
    nd = sum(1 for a, b in outs[1:] if not (torch.equal(a, outs[0][0]) and torch.equal(b, outs[0][1])))
    worst = max(float((a.float() - outs[0][0].float()).abs().max()) for a, _ in outs[1:])
    nbad = max(int((a != outs[0][0]).sum()) for a, _ in outs[1:])
    rows = max(int((a != outs[0][0]).any(dim=1).sum()) for a, _ in outs[1:])
    finite = bool(torch.isfinite(outs[0][0].float()).all())
    print(f"[{tag}] repeats differing: {nd}/{REPS - 1}  max |diff| {worst:.3e}  elements {nbad}  rows {rows}  finite {finite}", flush=True)
    return nd


def synthetic():
    B, H, N, dh = 8, 8, 4096, 40
    inner = H * dh
    g = torch.Generator().manual_seed(5)
    mk = lambda s=1.0: (torch.randn(B * N, inner, generator=g) * s).cuda()
    q32, k32, v32 = mk(1.5), mk(1.5), mk()
    qs = (q32 * (dh ** -0.5 * LOG2E)).to(torch.bfloat16)
    k, v = k32.to(torch.bfloat16), v32.to(torch.bfloat16)
    repeat("synthetic contiguous", qs, k, v, B, H, N, N, dh)
    qkv = torch.cat([qs, k, v], 1).contiguous()
    repeat("synthetic sliced ld=3*inner", qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], B, H, N, N, dh)
    # scale the scores up: many rows leave the optimistic range -> second pass
    for s in (3.0, 6.0, 12.0):
        qb = (q32 * s * (dh ** -0.5 * LOG2E)).to(torch.bfloat16)
        repeat(f"synthetic q x {s}", qb, k, v, B, H, N, N, dh)
    a = torch.randn(8192, 4096, device="cuda").to(torch.bfloat16)
    w = torch.randn(4096, 4096, device="cuda").to(torch.bfloat16)
    out = torch.empty(8192, 4096, dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.Stream()

    def side():
        with torch.cuda.stream(st):
            for _ in range(3):
                hip.gemm(a, w, out)
    repeat("synthetic contiguous + GEMM on a side stream", qs, k, v, B, H, N, N, dh, side=side)
    torch.cuda.synchronize()


def from_model():
    import bench
    model = bench.build_model("ctrlora_finetune_sd15_rank128.yaml", 0).cuda().train()
    model.set_engine_dtype(torch.bfloat16)
    d = bench.synth(8, 64, model.control_model.context_dim, "cuda", 99, 1)
    z, ctx, hint, t, noise = d["z"][0], d["ctx"][0], d["hint"][0], d["t"][0], d["noise"][0]
    calls = []
    orig = hip.attention_fwd_v2

    def rec(q, k, v, o, lse, B, H, N, Nkv, dh, scale, q_prescaled=False):
        if q_prescaled and dh == 40 and N == 4096 and Nkv == 4096:
            calls.append((q, k, v, B, H, N, Nkv, dh))       # views of live buffers: the record keeps them alive
        return orig(q, k, v, o, lse, B, H, N, Nkv, dh, scale, q_prescaled=q_prescaled)

    hip.attention_fwd_v2 = rec
    import ctrlora_amd.engine.blocks as blocks
    eng = model.engine()
    eng.overlap_streams = False
    x_noisy = model.q_sample(z, t, noise)
    eng.forward(x_noisy, t, ctx, [hint], record=True)
    torch.cuda.synchronize()
    hip.attention_fwd_v2 = orig
    print(f"captured {len(calls)} d_head-40 self-attention calls", flush=True)
    for i, (q, k, v, B, H, N, Nkv, dh) in enumerate(calls):
        qf = q.float()
        print(f"   call {i}: |q'| max {float(qf.abs().max()):.2f}  |k| max {float(k.float().abs().max()):.2f}  ld {q.stride(0)}", flush=True)
        repeat(f"model call {i}", q, k, v, B, H, N, Nkv, dh)
        repeat(f"model call {i} contiguous copies", q.contiguous(), k.contiguous(), v.contiguous(), B, H, N, Nkv, dh)


if __name__ == "__main__":
    synthetic()
    from_model()
