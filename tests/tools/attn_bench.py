"""Attention kernels A/B on the GPU: correctness vs fp64 torch + HIP-event timing through the C ABI.

    python tests/tools/attn_bench.py [--bwd] [--variants 14,0p,21p] [--rounds 7] [--out gpurun_out/attn_bench.json]

A variant token is a `cl_debug_attention_variant` code (csrc/debug_hooks.h), optionally followed by `p`: the call then uses
the pre-scaled-Q contract (CL_ATTN_Q_PRESCALED): q' = bf16(q32 * d^-0.5 * log2 e) where the plain variants get q = bf16(q32)
-- the same fp32 values rounded ONCE either way, as the to_q projection's epilogue does -- and every variant is checked
against the fp64 reference evaluated on ITS OWN q.  `--spike` adds a case whose scores grow by ~80 nats in a late key tile:
the optimistic pre-scaled-Q forward has to notice and repeat the block with the running maximum.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctrlora_amd import hip   # noqa: E402

LOG2E = 1.4426950408889634
NAMES = {0: "default", 1: "tile_sync", 11: "bwd_setprio", 13: "fwd_hybrid32_la3", 14: "fwd_hybrid32_la2"}


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def parse_variants(spec):
    out = []
    for tok in spec.split(","):
        tok = tok.strip()
        pre = tok.endswith("p")
        v = int(tok[:-1] if pre else tok)
        out.append((NAMES.get(v, f"v{v}") + ("+q_prescaled" if pre else ""), v, pre))
    return out


def reference(q, k, v, do, B, H, N, Nkv, dh, scale, bwd):
    inner = H * dh
    split = lambda x, n: x.double().reshape(B, n, H, dh).permute(0, 2, 1, 3).requires_grad_(bwd)
    qr, kr, vr = split(q, N), split(k, Nkv), split(v, Nkv)
    s = torch.einsum("bhid,bhjd->bhij", qr, kr) * scale
    orf = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), vr)
    back = lambda x, n: x.permute(0, 2, 1, 3).reshape(B * n, inner)
    ref = dict(o=back(orf, N).detach(), lse=torch.logsumexp(s, -1).detach())
    if bwd:
        orf.backward(do.double().reshape(B, N, H, dh).permute(0, 2, 1, 3))
        ref.update(dq=back(qr.grad, N), dk=back(kr.grad, Nkv), dv=back(vr.grad, Nkv))
    return ref


def case(dh, N, Nkv, B, variants, bwd, check=True, rounds=1, spike=False):
    H = 8
    inner = H * dh
    g = torch.Generator().manual_seed(dh + N + (7 if spike else 0))
    mk32 = lambda n, s=1.0: (torch.randn(B * n, inner, generator=g) * s).cuda()
    q32, k, v, do = mk32(N, 1.5), mk32(Nkv, 1.5).to(torch.bfloat16), mk32(Nkv).to(torch.bfloat16), mk32(N).to(torch.bfloat16)
    if spike:       # keys of the LAST tile of every (batch, head) line up with the queries: scores jump by tens of nats there
        kk = k.float().reshape(B, Nkv, inner)
        qq = q32.reshape(B, N, inner)
        kk[:, -16:, :] = qq[:, :16, :] * 6.0
        k = kk.reshape(B * Nkv, inner).to(torch.bfloat16)
    scale = dh ** -0.5
    q_plain = q32.to(torch.bfloat16)
    q_pre = (q32 * (scale * LOG2E)).to(torch.bfloat16)
    rp = (N + 63) // 64 * 64
    out = dict(shape=dict(dh=dh, N=N, Nkv=Nkv, B=B, H=H, spike=spike))
    refs = {}
    if check:
        with torch.enable_grad():
            if any(not p for _, _, p in variants):
                refs[False] = reference(q_plain, k, v, do, B, H, N, Nkv, dh, scale, bwd)
            if any(p for _, _, p in variants):
                # the reference's q is what the kernel's q' MEANS: q' / (scale log2 e); gradients are those of that q
                refs[True] = reference(q_pre.double() / (scale * LOG2E), k, v, do, B, H, N, Nkv, dh, scale, bwd)
    flops_f = 4.0 * B * H * N * Nkv * dh
    o = torch.empty_like(q_plain)
    lse = torch.empty(B, H, rp, dtype=torch.float32, device="cuda")
    delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q_plain), torch.empty_like(k), torch.empty_like(v)
    L = hip.lib()

    def select(var):
        assert L.cl_debug_attention_variant(var) == 0, f"unknown attention variant {var}"

    fwd = lambda pre: hip.attention_fwd_v2(q_pre if pre else q_plain, k, v, o, lse, B, H, N, Nkv, dh, scale, q_prescaled=pre)
    row_ws = torch.empty(lse.numel() * 8, dtype=torch.float32, device="cuda")     # (-lse, -delta) triples: the engine passes it too
    bw = lambda pre: hip.attention_bwd_v2(q_pre if pre else q_plain, k, v, o, do, lse, delta, dq, dk, dv, B, H, N, Nkv, dh, scale,
                                          q_prescaled=pre, row_ws=row_ws if pre else None)
    runs = {}
    for name, var, pre in variants:
        select(var)
        o.fill_(float("nan")); lse.fill_(float("nan"))
        fwd(pre)
        r = {}
        rf = refs.get(pre)
        if rf is not None:
            torch.cuda.synchronize()
            r["o_err"] = rel(o, rf["o"]); r["lse_err"] = rel(lse[:, :, :N] * 0.6931471805599453, rf["lse"])
        if bwd:
            bw(pre)
            if rf is not None:
                torch.cuda.synchronize()
                r.update(dq_err=rel(dq, rf["dq"]), dk_err=rel(dk, rf["dk"]), dv_err=rel(dv, rf["dv"]))
        runs[name] = dict(r, f=[], b=[])
    # interleaved A/B (guide rule 24): the FIRST variant timed in a process measures 15-20 % slow (clock ramp after the fp64
    # reference), so every variant is timed once per round, round-robin, after a warm-up; median / min over the rounds
    select(variants[0][1])
    timeit(lambda: fwd(variants[0][2]), iters=60, warm=10)
    for _ in range(max(rounds, 1)):
        for name, var, pre in variants:
            select(var)
            runs[name]["f"].append(timeit(lambda: fwd(pre), iters=10, warm=2))
            if bwd:
                runs[name]["b"].append(timeit(lambda: bw(pre), iters=10, warm=2))
    for name, r in runs.items():
        f = sorted(r.pop("f")); bb = sorted(r.pop("b"))
        r["fwd_us_median"] = round(f[len(f) // 2] * 1e3, 1); r["fwd_us_min"] = round(f[0] * 1e3, 1)
        r["fwd_tflops"] = round(flops_f / f[len(f) // 2] * 1e-9, 1)
        if bb:
            r["bwd_us_median"] = round(bb[len(bb) // 2] * 1e3, 1); r["bwd_us_min"] = round(bb[0] * 1e3, 1)
            r["bwd_tflops"] = round(2.5 * flops_f / bb[len(bb) // 2] * 1e-9, 1)
        out[name] = r
    select(0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--spike", action="store_true", help="add the case that forces the optimistic forward's second pass")
    ap.add_argument("--variants", default="14,0p", help="comma list of cl_debug_attention_variant codes, `p` suffix = pre-scaled Q")
    ap.add_argument("--shapes", default="40,4096,4096,8;80,1024,1024,8;40,4096,4096,32;40,1024,1024,2;40,256,128,8")
    ap.add_argument("--rounds", type=int, default=5, help="interleaved A/B: median / min over the rounds")
    ap.add_argument("--out", default=None)
    ap.add_argument("--lib", default=None, help="load this build of the library instead (probe builds: tools/build_probes.sh attn_bwd_abl)")
    args = ap.parse_args()
    if args.lib:
        hip.LIB_PATH = os.path.abspath(args.lib)
    variants = parse_variants(args.variants)
    res = []
    cases = [tuple(int(x) for x in sh.split(",")) + (False,) for sh in args.shapes.split(";")]
    if args.spike:
        cases.append((40, 4096, 4096, 2, True))
    for dh, N, Nkv, B, spike in cases:
        r = case(dh, N, Nkv, B, variants, args.bwd, check=(B <= 8 and not args.no_check), rounds=args.rounds, spike=spike)
        print(json.dumps(r), flush=True)
        res.append(r)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f)


if __name__ == "__main__":
    main()
