"""Headline benchmark of the CtrLoRA hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], weak scaling over GPUs for configs[2]): 512x512 LoRA fine-tuning of
`configs/ctrlora_finetune_sd15_rank128.yaml` -- ControlNetFinetune (rank-128 LoRA) + frozen SD1.5 UNet --
per-GPU batch 8, bf16 storage / fp32 accumulate, synthetic latents (z, hint latent, text context, noise, t),
random-init weights.  One step = p_losses forward + hand-written backward + LoRA-only gradient all-reduce
(RCCL) + fused AdamW, driven through the drop-in cldm API (ControlFinetuneLDM.p_losses / configure_optimizers).

Prints ONE JSON line (rank 0).  `value` = images/s over all ranks.  Also reported:
  roofline      whole-step MFMA roofline: SURVEY.md 8(d) algorithmic FLOPs per image (1.996 TF for r128: forward
                + data-gradients + LoRA/zero-conv weight-gradients, no recompute, no frozen dW) x images/s / 2.5 PF,
                plus the dominant kernel (implicit-GEMM conv 320->320 @64x64) timed with HIP events on this stream.
  ddim          DDIM denoise steps/s (CFG 7.5, batch 16, hint latent encoded once) on the same silicon.
  cpu_baseline  the oracle (CPU restatement of the reference) on the host cores: one TIMED optimizer step of
                BASELINE configs[0] (rank 32, bs 1) and, under ddim.cpu_baseline, two timed DDIM steps with CFG.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_TFLOP_PER_IMAGE = {32: 1.913, 64: 1.941, 128: 1.996, 256: 2.11, 512: 2.33}
DDIM_TFLOP_PER_STEP_IMAGE = 2.207
STOCK_BF16_IMAGES_PER_S = 42.3   # profiles/r02_compare_precision.json: stock kernels, bf16 autocast, B=8, one MI355X


def stock_reference():
    """(images/s, source) of the reference's modules on PyTorch-ROCm's own kernels under bf16 autocast, B = 8, one MI355X: the
    newest tests/tools/compare_stock.py result under profiles/ (re-measured in round 5), else the round-2 figure."""
    for name in ("r06_compare_precision.json", "r05_compare_precision.json", "r02_compare_precision.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return float(json.load(f)["stock_bf16_autocast"]["images_per_s"]), "profiles/" + name
        except Exception:
            continue
    return STOCK_BF16_IMAGES_PER_S, "profiles/r02_compare_precision.json"
def src_hash(rel):
    """sha256 (first 12 hex digits) of a source file of this tree: static measurement artefacts under profiles/ record the hash of
    the kernel source they were measured on, and the bench marks them `stale` when the source has changed since."""
    import hashlib
    try:
        with open(os.path.join(ROOT, rel), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:12]
    except OSError:
        return None


def provenance(meta, rel="ctrlora_amd/csrc/gemm.hip"):
    """{"measured_at_commit", "source_sha256", "stale"} for a static artefact whose JSON carries the first two."""
    have = src_hash(rel)
    was = (meta or {}).get("source_sha256")
    return dict(measured_at_commit=(meta or {}).get("measured_at_commit"), source=rel, source_sha256_then=was, source_sha256_now=have,
                stale=bool(was is None or have is None or was != have))


PEAK_HBM_TBS = 8.0          # TB/s, MI355X_MICROARCH.md
PEAK_BF16_TFLOPS = 2500.0


def build_model(config, seed, rank_override=None, tiny=False, mutate=None, fast_init=None):
    """The drop-in path: YAML -> instantiate_from_config, VAE / CLIP replaced by Identity (synthetic latents).
    `mutate(params)` may edit the YAML's model params before instantiation (tests).  fast_init (default: on at full width unless
    CTRLORA_FAST_INIT=0): the default distributions drawn from one random block (ctrlora_amd/fastinit.py) instead of torch's
    per-layer calls -- other VALUES than torch.manual_seed(seed) + default init gives; tests whose absolute bf16 gates were
    calibrated on those pass False."""
    import contextlib
    from ldm.util import instantiate_from_config
    with open(os.path.join(ROOT, "configs", config)) as f:
        cfg = yaml.safe_load(f)["model"]
    p = cfg["params"]
    if mutate is not None:
        mutate(p)
    p["first_stage_config"] = {"target": "torch.nn.Identity"}
    p["cond_stage_config"] = {"target": "torch.nn.Identity"}
    if tiny:
        for k in ("control_stage_config", "unet_config"):
            p[k]["params"].update(model_channels=64, context_dim=96)
        p["control_stage_config"]["params"]["lora_rank"] = 32
    torch.manual_seed(seed)
    # full width: torch's per-layer default initialisation of 1.3 G parameters is ~25 s of host time per build
    # (ctrlora_amd/fastinit.py: the same distribution from one random block in ~3 s)
    from ctrlora_amd.fastinit import fill_default_init, skip_default_init
    if fast_init is None:
        fast_init = os.environ.get("CTRLORA_FAST_INIT", "1") != "0"
    fast_init = bool(fast_init) and not tiny
    with (skip_default_init() if fast_init else contextlib.nullcontext()):
        model = instantiate_from_config(cfg)
    if fast_init:
        fill_default_init(model, seed)
    # re-draw the zero-initialised parameters (zero convs, proj_out, out conv, LoRA up) so that no path is
    # trivially zero
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for _, prm in model.named_parameters():
            if prm.numel() > 0 and float(prm.abs().max()) == 0.0:
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.02)
    return model


def synth(B, H, ctx_dim, device, seed, n):
    g = torch.Generator(device="cpu").manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g).to(device)
    return dict(z=[mk(B, 4, H, H) for _ in range(n)], hint=[mk(B, 4, H, H) * 0.9 for _ in range(n)],
                ctx=[mk(B, 77, ctx_dim) for _ in range(n)], noise=[mk(B, 4, H, H) for _ in range(n)],
                t=[torch.randint(0, 1000, (B,), generator=g).to(device) for _ in range(n)])


def conv_kernel_probe(device, dtype, iters=30):
    """Dominant kernel: ResBlock conv 320->320 @ 64x64, B=8 -- gemm_fl_kernel<bf16,256,160> (implicit-GEMM, stride-1
    conv mode) -- timed with HIP events on the stream it is launched on (torch's current stream).  `traffic` is
    the HBM bytes per launch measured with rocprofv3 PMC passes on the same launch (profiles/dominant_kernel_traffic.json)."""
    from ctrlora_amd import hip
    B, H, C = 8, 64, 320
    x = torch.randn(B * H * H, C, device=device).to(dtype)
    w = (torch.randn(C, 9 * C, device=device) * 0.02).to(dtype)
    bias = torch.zeros(C, device=device)
    out = torch.empty(B * H * H, C, device=device, dtype=dtype)
    run = lambda: hip.gemm(x, w, out, bias=bias, mode=hip.CONV_S1, conv=(B, H, H, H, H), k1=C)
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * B * H * H * C * 9 * C
    traffic, tmeta = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")) as f:
            tmeta = json.load(f)
            traffic = tmeta["hbm_bytes_per_launch"]
    except Exception:
        pass
    return dict(kernel="gemm_fl_kernel<bf16,256x160,8 waves> conv3x3 320->320 @64x64 B8 (60.4 GFLOP/launch)",
                ms=round(ms, 4), achieved=round(flops / ms * 1e-9, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                frac=round(flops / ms * 1e-9 / PEAK_BF16_TFLOPS, 4), traffic=traffic, traffic_provenance=provenance(tmeta))


def dominant_kernel_rocprof():
    """The same launch INSIDE the replayed training step, from this round's rocprofv3 kernel trace joined with the launch tags
    (tools/prof_shapes.py -> profiles/r05_final/train_shapes_in_step.txt): static here, published beside the live HIP-event figure
    because hot isolated launches run ~8 % faster than the launch does in the step."""
    path = next((q for q in (os.path.join(ROOT, "profiles", r, "train_shapes_in_step.txt") for r in ("r06_final3", "r06_final", "r05_final"))
                 if os.path.exists(q)), "")
    meta = None
    try:
        with open(os.path.join(os.path.dirname(path), "provenance.json")) as f:
            meta = json.load(f)
    except Exception:
        pass
    try:
        us = n = 0.0
        for ln in open(path):
            f = ln.split()
            # rows: us/step n/step avg_us TF/s TB/s of_roof mode M N K1 K2 act res kernel...
            if len(f) > 13 and f[6] == "1" and f[7:11] == ["32768", "320", "320", "0"] and f[11] == "0":
                us += float(f[0]); n += float(f[1])
        if not n:
            return None
        avg = us / n
        return dict(us_per_launch=round(avg, 2), launches_per_step=int(n), tflops=round(60.4e3 / avg, 1),
                    frac=round(60.4e3 / avg / PEAK_BF16_TFLOPS, 4), table=os.path.relpath(path, ROOT) + " (static)", **provenance(meta))
    except Exception:
        return None


def family_census(model, opt, data, reps=10):
    """Time-weighted MFMA roofline of the two contraction kernel families, measured live: every hip.gemm /
    attention call of ONE eager training step is recorded (shape signature + algorithmic FLOPs), then every unique
    signature is re-launched in isolation on this stream between HIP events.  achieved = sum(FLOPs) / sum(time x calls):
    the figure the step actually gets from the family, not its best shape."""
    import collections
    from ctrlora_amd import hip
    calls = {"gemm": collections.OrderedDict(), "attention": collections.OrderedDict(), "hbm": collections.OrderedDict()}
    o_gemm, o_af, o_ab = hip.gemm, hip.attention_fwd_v2, hip.attention_bwd_v2

    # HBM-bound family ("reported per-kernel ... vs 8 TB/s"): GroupNorm(+SiLU) / LayerNorm / GEGLU forward and
    # backward and the column sums.  ALGORITHMIC bytes = every operand once (a two-pass normalisation re-reads its input
    # from L2 / Infinity Cache, not counted), in the storage dtype.
    def _bytes_hbm(name, a, kw):
        es = a[0].element_size()
        M, C = a[0].shape
        if name == "groupnorm_fwd" or name == "layernorm_fwd":
            return 2 * M * C * es
        if name == "groupnorm_bwd" or name == "layernorm_bwd":
            return (4 if kw.get("accum") is not None else 3) * M * C * es
        if name == "geglu_fwd":
            return 3 * M * (C // 2) * es
        if name == "geglu_bwd":
            return 5 * M * (C // 2) * es
        return M * C * es                                            # colsum

    hbm_orig = {n: getattr(hip, n) for n in ("groupnorm_fwd", "groupnorm_bwd", "layernorm_fwd", "layernorm_bwd",
                                             "geglu_fwd", "geglu_bwd", "colsum")}

    def _mk_hbm(name, fn):
        def rec(*a, **kw):
            key = (name, tuple(a[0].shape), kw.get("accum") is not None, kw.get("dgamma") is not None)
            e = calls["hbm"].setdefault(key, [0, 0.0, lambda: fn(*a, **kw), _bytes_hbm(name, a, kw)])
            e[0] += 1
            return fn(*a, **kw)
        return rec

    def rec_gemm(a1, w1, out, **kw):
        M = out.shape[0] if kw.get("M") is None else kw["M"]
        N = out.shape[1] if kw.get("N") is None else kw["N"]
        k1 = a1.shape[1] if kw.get("k1") is None else kw["k1"]
        mode, a2 = kw.get("mode", hip.LINEAR), kw.get("a2")
        # grouped second segment: an output column sees ONE group's K2 (= columns of W2), not the whole of A2
        k2 = 0 if a2 is None else (kw["w2"].shape[1] if kw.get("a2_group_n") else a2.shape[1])
        key = (mode, M, N, k1, k2, kw.get("conv"), kw.get("residual") is not None, bool(kw.get("out_f32", False)),
               kw.get("act", 0))
        if not kw.get("atomic", False):
            taps = 1 if mode == hip.LINEAR else 9
            fl = 2.0 * M * N * (taps * k1 + k2)
            # algorithmic HBM bytes of the launch: operands once (a conv reads its input tensor once), result once
            esz = a1.element_size()
            conv = kw.get("conv")
            a_rows = M if conv is None else conv[0] * conv[1] * conv[2]
            n_out = N // 2 if kw.get("act", 0) == hip.ACT_GEGLU else N
            a1_cols = k1 * (N // kw["a1_group_n"]) if kw.get("a1_group_n") else k1      # grouped first segment reads every group's columns once
            a2_cols = k2 * (N // kw["a2_group_n"]) if kw.get("a2_group_n") else k2
            by = esz * (a_rows * a1_cols + M * a2_cols + N * (taps * k1 + k2)) + M * n_out * (4 if kw.get("out_f32") else esz)
            if kw.get("residual") is not None:
                by += M * n_out * esz
            e = calls["gemm"].setdefault(key, [0, fl, lambda: o_gemm(a1, w1, out, **kw), by])
            e[0] += 1
        return o_gemm(a1, w1, out, **kw)

    def rec_af(q, k, v, o, lse, B, H, N, Nkv, dh, scale, **kw):
        e = calls["attention"].setdefault(("fwd", B, H, N, Nkv, dh), [0, 4.0 * B * H * N * Nkv * dh,
                                                                   lambda: o_af(q, k, v, o, lse, B, H, N, Nkv, dh, scale, **kw)])
        e[0] += 1
        return o_af(q, k, v, o, lse, B, H, N, Nkv, dh, scale, **kw)

    def rec_ab(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, N, Nkv, dh, scale, **kw):
        nmm = 5 if dk is not None else 3          # S, dP, dQ (+ dK, dV): algorithmic, the two kernels recompute S / dP
        e = calls["attention"].setdefault(("bwd", B, H, N, Nkv, dh, dk is not None), [
            0, 2.0 * nmm * B * H * N * Nkv * dh,
            lambda: o_ab(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, N, Nkv, dh, scale, **kw)])
        e[0] += 1
        return o_ab(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, N, Nkv, dh, scale, **kw)

    hip.gemm, hip.attention_fwd_v2, hip.attention_bwd_v2 = rec_gemm, rec_af, rec_ab
    for n, fn in hbm_orig.items():
        setattr(hip, n, _mk_hbm(n, fn))
    # rank 0 runs this alone: no collective may be issued from the recorded step (eager data-parallel hooks off)
    dp = getattr(model, "dp", None)
    dp_was = None if dp is None else dp.enabled
    if dp is not None:
        dp.enabled = False
    try:
        opt.zero_grad()
        cond = {"c_crossattn": [data["ctx"][0]], "c_concat": [data["hint"][0]]}
        model.engine_train_step(data["z"][0], cond, data["t"][0], data["noise"][0])
        torch.cuda.synchronize()
    finally:
        hip.gemm, hip.attention_fwd_v2, hip.attention_bwd_v2 = o_gemm, o_af, o_ab
        for n, fn in hbm_orig.items():
            setattr(hip, n, fn)
        if dp is not None:
            dp.enabled = dp_was
    out = {}
    for fam, tab in calls.items():
        tot_us, tot_fl, n, ideal_us, mem_us = 0.0, 0.0, 0, 0.0, 0.0
        for key, ent in tab.items():
            cnt, fl, run = ent[:3]
            for _ in range(2):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            if fam == "hbm":
                ent.append(us)
            if fam == "attention":
                ent.append(us)
            tot_us += us * cnt; tot_fl += fl * cnt; n += cnt
            t_mfma = fl / (PEAK_BF16_TFLOPS * 1e6)                      # us at the dense bf16 MFMA peak
            t_hbm = (ent[3] / (PEAK_HBM_TBS * 1e6)) if len(ent) > 3 else 0.0   # us at the HBM peak
            ideal_us += max(t_mfma, t_hbm) * cnt
            if t_hbm > t_mfma:
                mem_us += us * cnt
        if fam == "hbm":
            if tot_us > 0:
                tot_by = sum(e[3] * e[0] for e in tab.values())
                per = {}
                for key, e in tab.items():
                    d = per.setdefault(key[0], [0, 0.0, 0.0])
                    d[0] += e[0]; d[1] += e[3] * e[0]; d[2] += e[4] * e[0]
                tbs = tot_by / tot_us * 1e-6
                out[fam] = dict(bound="hbm", achieved=round(tbs * 1e3, 1), peak=PEAK_HBM_TBS * 1e3, unit="GB/s",
                                frac=round(tbs / PEAK_HBM_TBS, 4), ms_per_step=round(tot_us * 1e-3, 2), launches_per_step=n,
                                unique_shapes=len(tab), algorithmic_MB_per_step=round(tot_by * 1e-6, 1),
                                kernels="GroupNorm(+SiLU) / LayerNorm / GEGLU forward + backward, column sums; every operand "
                                        "counted once; timed per signature in isolation with HIP events",
                                per_kernel={k: dict(calls=v[0], ms=round(v[2] * 1e-3, 3), GBps=round(v[1] / v[2] * 1e-3, 1))
                                            for k, v in per.items()},
                                # where the family's time is: the five signatures with the largest time x calls
                                top=[dict(kernel=k[0], rows_cols=list(k[1]), accum=k[2], trainable=k[3], calls=e[0], us=round(e[4], 1),
                                          GBps=round(e[3] / e[4] * 1e-3, 1))
                                     for k, e in sorted(tab.items(), key=lambda kv: -kv[1][4] * kv[1][0])[:6]])
            continue
        if tot_us > 0:
            tf = tot_fl / tot_us * 1e-6
            out[fam] = dict(achieved=round(tf, 1), frac=round(tf / PEAK_BF16_TFLOPS, 4), unit="TFLOP/s",
                            ms_per_step=round(tot_us * 1e-3, 2), launches_per_step=n, unique_shapes=len(tab),
                            gflop_per_step=round(tot_fl * 1e-9, 1))
            if fam == "attention":
                # the north_star's ">= 50 % MFMA on the attention kernel" as one field per kernel: USEFUL (algorithmic) FLOPs of a
                # signature / its launch time / the dense bf16 peak.  "fwd" = attn_fwd40_kernel at d_head 40; "bwd" = the
                # attn_bwd_dq + attn_bwd_dkv pair of one hip.attention_bwd call (their split is in profiles/*/train_kernel_stats)
                out[fam]["per_signature"] = [
                    dict(kind=k[0], B=k[1], H=k[2], N=k[3], Nkv=k[4], d_head=k[5], calls=e[0], us=round(e[3], 1),
                         useful_tflops=round(e[1] / e[3] * 1e-6, 1), useful_frac=round(e[1] / e[3] * 1e-6 / PEAK_BF16_TFLOPS, 4))
                    for k, e in sorted(tab.items(), key=lambda kv: -kv[1][3] * kv[1][0]) if len(e) > 3]
                dom = [r for r in out[fam]["per_signature"] if r["N"] == r["Nkv"] == 4096]
                out[fam]["useful_frac"] = {r["kind"] + ("" if r["kind"] == "fwd" else "_dq_dkv_pair"): r["useful_frac"] for r in dom}
            if fam == "gemm":
                # every launch against ITS OWN roofline, max(FLOPs / MFMA peak, algorithmic bytes / HBM peak): the K = 320
                # products of the 64x64 level are below the machine balance (160 FLOP/B against 312) and HBM-bound
                out[fam]["frac_of_shape_rooflines"] = round(ideal_us / tot_us, 4)
                out[fam]["hbm_bound_share_of_time"] = round(mem_us / tot_us, 4)
    return out


def vae_bench(device, dtype, B=8, iters=3):
    """First-stage cost of one REAL training step: the reference VAE-encodes the B target images
    (ddpm.py:773) and the B condition images (cldm_ctrlora_finetune.py:76-77) of every step -- 2B encodes of 512x512
    images on the engine's AutoencoderKL (random-init SD ddconfig)."""
    from ldm.models.autoencoder import AutoencoderKL
    dd = dict(attn_resolutions=[], ch=128, ch_mult=[1, 2, 4, 4], double_z=True, dropout=0.0, in_channels=3, num_res_blocks=2,
              out_ch=3, resolution=256, z_channels=4)
    torch.manual_seed(0)
    vae = AutoencoderKL(ddconfig=dd, lossconfig=dict(target="torch.nn.Identity"), embed_dim=4).to(device).eval()
    vae.engine_dtype = dtype
    x = torch.rand(2 * B, 3, 512, 512, device=device) * 2 - 1
    with torch.no_grad():
        vae.encode(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            post = vae.encode(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    assert torch.isfinite(post.mean).all() and "_enc" in vae.__dict__
    gf = 1117.0 * 2 * B              # GFLOP: 1 117 per 512x512 encode
    out = dict(images=2 * B, ms=round(ms, 2), tflops=round(gf / ms, 1), mfma_frac=round(gf / ms / PEAK_BF16_TFLOPS, 4))
    try:
        out["roofline"] = first_stage_roofline(device, dtype, 2 * B)
    except Exception as e:
        print(f"[bench] first-stage roofline probe failed: {type(e).__name__}: {e}", file=sys.stderr)
    return out


def first_stage_roofline(device, dtype, NB):
    """The encoder's kernels at its four resolutions, each timed alone with HIP events on the launching stream (the encode is one
    serial chain of them: profiles/r05_vae/vae_kernel_stats.txt has the in-run totals): the 3x3 convs of the ResnetBlocks
    (ldm/modules/diffusionmodules/model.py:97-149) against the MFMA peak, GroupNorm(32)+swish against the HBM peak on
    algorithmic bytes (one read + one write of the tensor)."""
    from ctrlora_amd import hip

    def timed(run, iters):
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    rows = []
    for H, C, n_conv, n_gn in ((512, 128, 4, 4), (256, 256, 3, 3), (128, 512, 3, 3), (64, 512, 8, 9)):
        M = NB * H * H
        x = torch.randn(M, C, device=device).to(dtype)
        w = (torch.randn(C, 9 * C, device=device) * 0.02).to(dtype)
        bias = torch.zeros(C, device=device)
        y = torch.empty(M, C, device=device, dtype=dtype)
        ms_c = timed(lambda: hip.gemm(x, w, y, bias=bias, mode=hip.CONV_S1, conv=(NB, H, H, H, H), k1=C), 4)
        fl = 2.0 * M * C * 9 * C
        gamma, beta = torch.ones(C, device=device), torch.zeros(C, device=device)
        stats = torch.empty(NB, 32, 2, device=device)
        ws = torch.empty(max(hip.groupnorm_ws(NB, H * H, C), 1 << 20), device=device)
        ms_g = timed(lambda: hip.groupnorm_fwd(x, y, gamma, beta, NB, H * H, 1e-6, True, stats, ws), 4)
        by = 2.0 * M * C * 2
        rows.append(dict(level=f"{H}x{H}x{C}", conv3x3=dict(n_per_encode=n_conv, ms=round(ms_c, 3), tflops=round(fl / ms_c * 1e-9, 1),
                                                            mfma_frac=round(fl / ms_c * 1e-9 / PEAK_BF16_TFLOPS, 4)),
                         groupnorm_swish=dict(n_per_encode=n_gn, ms=round(ms_g, 3), tb_per_s=round(by / ms_g * 1e-9, 2),
                                              hbm_frac=round(by / ms_g * 1e-9 / 8.0, 4))))
        del x, w, y, ws
    t_conv = sum(r["conv3x3"]["ms"] * r["conv3x3"]["n_per_encode"] for r in rows)
    t_gn = sum(r["groupnorm_swish"]["ms"] * r["groupnorm_swish"]["n_per_encode"] for r in rows)
    return dict(levels=rows, ms_in_same_channel_convs=round(t_conv, 2), ms_in_groupnorms=round(t_gn, 2),
                note="isolated launches at the encoder's shapes (16 images); the channel-changing convs, the two stride-2 convs, "
                     "conv_in and the 4096-token attention are the remainder of `ms`")


def ddim_bench(device, dtype, B=16, S=50, tiny=False, loops=5, warm_loops=10, extras=True):
    from cldm.ddim_hacked import DDIMSampler
    model = build_model("inference/ctrlora_sd15_rank128_1lora.yaml", 0, tiny=tiny).to(device).eval()
    model.set_engine_dtype(dtype)
    cd = model.control_model.context_dim
    H = 64
    g = torch.Generator().manual_seed(7)
    hint = torch.randn(B, 4, H, H, generator=g).to(device)
    cond = {"c_concat": [hint], "c_crossattn": [torch.randn(B, 77, cd, generator=g).to(device)]}
    unc = {"c_concat": [hint], "c_crossattn": [torch.randn(B, 77, cd, generator=g).to(device)]}
    x_T = torch.randn(B, 4, H, H, generator=g).to(device)
    sampler = DDIMSampler(model)
    sampler.reuse_graph = True               # same conditioning tensors, same weights: capture once, replay in every loop
    run = lambda s: sampler.sample(s, B, (4, H, H), cond, verbose=False, eta=0.0, x_T=x_T,
                                   unconditional_guidance_scale=7.5, unconditional_conditioning=unc)
    run(6)                                   # kernel attribute set-up, allocator, capture path
    for _ in range(warm_loops):              # warm-up loops at the timed length (the first one captures the S-step graph)
        run(S)
    torch.cuda.synchronize()
    times = []
    for _ in range(loops):
        t0 = time.perf_counter()
        out, _ = run(S)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    assert torch.isfinite(out).all()
    times.sort()
    dt = times[len(times) // 2]              # median loop
    sps = S / dt
    hits = getattr(sampler, "graph_hits", 0)
    # (a) ONE cold sample() call as rounds 1-2 timed it and as a one-shot user pays it: fresh sampler, eager first step,
    #     capture of the step graph, S - 2 replays
    dt_cold = None
    if extras:
        cold = DDIMSampler(model)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cold.sample(S, B, (4, H, H), cond, verbose=False, eta=0.0, x_T=x_T, unconditional_guidance_scale=7.5,
                    unconditional_conditioning=unc)
        torch.cuda.synchronize()
        dt_cold = time.perf_counter() - t0
        del cold
    # (b) condition IMAGES instead of latents (what scripts/sample.py hands over): hoisted = the VAE encode once per call and
    #     the posterior re-sampled on the device in every apply_model; reference_faithful = the encoder inside every
    #     apply_model call as the reference runs it (4.44 TFLOP per image and step)
    image_leg = None
    try:
        if extras:
            image_leg = ddim_image_hint_bench(model, device, dtype, B, H, cd, S=20 if not tiny else 4)
    except Exception as e:      # the headline leg above stands on its own
        print(f"[bench] DDIM image-hint leg failed ({type(e).__name__}: {e})", file=sys.stderr)
    cold_leg = None if dt_cold is None else dict(
        steps_per_s=round(S / dt_cold, 3), note="one sample() call on a fresh sampler incl. the eager first step and the graph "
                                               "capture (the methodology of BENCH_r01/r02)")
    return dict(cold=cold_leg,
                image_hint=image_leg,
                metric="DDIM denoise steps/s (CFG 7.5, both passes, all B images)", value=round(sps, 3), batch=B, S=S,
                ms_per_step=round(dt / S * 1e3, 2), best=round(S / times[0], 3), loops=loops,
                mfma_frac=round(DDIM_TFLOP_PER_STEP_IMAGE * B * sps / PEAK_BF16_TFLOPS, 4),
                warmup_loops=warm_loops, graph_captures_in_timed_loops=0 if hits >= loops else None,
                note=f"median of {loops} full S={S} loops (best loop in `best`) after {warm_loops} warm-up loops of the same length; the "
                     "denoise-step graph is captured once in the warm-up and replayed S times per timed loop; hint latent given "
                     "(VAE encode hoisted out of the loop); cond+uncond batched")


def ddim_image_hint_bench(model, device, dtype, B, H, cd, S=20):
    """DDIM with 512x512 condition IMAGES (B x 3 x 8H x 8H in [0, 1]) through the model's own first stage: the product
    default (encode hoisted out of the loop, posterior re-sampled per call) and the reference's schedule (encode inside every
    apply_model: cldm_ctrlora_inference.py:165-172, ddim_hacked.py:181-231)."""
    from cldm.ddim_hacked import DDIMSampler
    from ldm.models.autoencoder import AutoencoderKL
    dd = dict(attn_resolutions=[], ch=128, ch_mult=[1, 2, 4, 4], double_z=True, dropout=0.0, in_channels=3, num_res_blocks=2,
              out_ch=3, resolution=256, z_channels=4)
    torch.manual_seed(0)
    vae = AutoencoderKL(ddconfig=dd, lossconfig=dict(target="torch.nn.Identity"), embed_dim=4).to(device).eval()
    vae.engine_dtype = dtype
    prev_vae = model.first_stage_model
    model.first_stage_model = vae            # (build_model put an Identity there: the core legs take latents)
    try:
        return _ddim_image_hint_legs(model, device, B, H, cd, S)
    finally:
        model.first_stage_model = prev_vae


def _ddim_image_hint_legs(model, device, B, H, cd, S):
    from cldm.ddim_hacked import DDIMSampler
    g = torch.Generator().manual_seed(11)
    img = torch.rand(B, 3, 8 * H, 8 * H, generator=g).to(device)
    cond = {"c_concat": [img], "c_crossattn": [torch.randn(B, 77, cd, generator=g).to(device)]}
    unc = {"c_concat": [img], "c_crossattn": [torch.randn(B, 77, cd, generator=g).to(device)]}
    x_T = torch.randn(B, 4, H, H, generator=g).to(device)
    out = {}
    for name, hoist in (("hoisted", True), ("reference_faithful", False)):
        sampler = DDIMSampler(model)
        sampler.reuse_graph = True
        sampler.hoist_hint_encode = hoist
        run = lambda: sampler.sample(S, B, (4, H, H), cond, verbose=False, eta=0.0, x_T=x_T,
                                     unconditional_guidance_scale=7.5, unconditional_conditioning=unc)
        run()
        run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            x, _ = run()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        assert torch.isfinite(x).all()
        sps = S / min(ts)
        tf = 4.440 if not hoist else DDIM_TFLOP_PER_STEP_IMAGE
        out[name] = dict(steps_per_s=round(sps, 3), ms_per_step=round(1e3 / sps, 2), S=S,
                         tflop_per_step_image=tf, mfma_frac=round(tf * B * sps / PEAK_BF16_TFLOPS, 4))
        del sampler
    out["note"] = ("condition images 512x512; hoisted: one VAE encode per sample() call (amortised over S), posterior re-sampled on "
                   "the device every apply_model; reference_faithful: encoder inside every apply_model call, 4.44 TFLOP per "
                   "image and step (SURVEY.md 8d); best of 2 loops after 2 warm-up loops")
    return out


def pretrain_bench_dp(device, dtype, world, rank, B=8, steps=9, warmup=9, tiny=False):
    """BASELINE.json configs[3] under data parallelism (`bench.py --gpus N --pretrain-only`): eager launches, every rank draws its
    own task per step (datasets/multi_task_scheduler.py:59 -- ranks may train different banks), base-ControlNet gradients
    (~1.4 GB fp32) leave as 32 MB buckets from the backward's stage hooks (BankedGradAllReduce.attach), the live banks after
    it; PretrainAdamW with grad_scale 1 / N.  Reports aggregate images/s (max over ranks) and what the exchange costs:
    the same steps with the exchange switched off (measurement only, after the timed region)."""
    import torch.distributed as dist
    model = build_model("ctrlora_pretrain_sd15_9tasks_rank128.yaml", 0, tiny=tiny).to(device).train()
    model.set_engine_dtype(dtype)
    model.learning_rate = 1e-5
    dp = model.init_data_parallel()
    opt = model.configure_optimizers()
    tasks = list(model.control_model.tasks)
    data = synth(B, 64, model.control_model.context_dim, device, 4321 + rank, 2)
    rng = np.random.RandomState(97 + rank)          # per-rank task stream, as the reference's unseeded sampler gives

    def step(i, exchange=True):
        j = i % 2
        cond = {"c_crossattn": [data["ctx"][j]], "c_concat": [data["hint"][j]], "task": tasks[int(rng.randint(len(tasks)))]}
        dp.enabled = exchange
        opt.zero_grad()
        loss, _ = model.p_losses(data["z"][j], cond, data["t"][j], noise=data["noise"][j])
        loss.backward()
        opt.step()
        return loss

    def timed(n, exchange):
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            loss = step(i, exchange)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        tm = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        return float(tm) / n, loss

    for i in range(warmup):
        step(i)
    dt, loss = timed(steps, True)
    launches = dp.inner.last_launches
    tail = dp.inner.exposed_tail_elems
    dt_off, _ = timed(steps, False)
    ex = model.control_model.executor()
    return dict(metric="Base-ControlNet multi-task pre-training images/sec (data parallel)", value=round(world * B / dt, 2),
                unit="images/s", n_gpus=world, ms_per_step=round(dt * 1e3, 2), batch_per_gpu=B, steps=steps, warmup=warmup,
                launch="eager launches, bucketed all-reduce of the base-ControlNet gradients from the backward's stage hooks",
                data_parallel=dict(base_payload_MB_per_step=round(ex.tr.numel * 4 / 1e6, 1), bucket_MB=32,
                                   buckets_launched_during_backward=int(launches), tail_MB_reduced_after_backward=round(tail * 4 / 1e6, 1),
                                   ms_per_step_without_exchange=round(dt_off * 1e3, 2),
                                   exposed_allreduce_ms_per_step=round((dt - dt_off) * 1e3, 2)),
                loss=round(float(loss.detach()), 5), dtype=str(dtype).replace("torch.", ""),
                config="ctrlora_pretrain_sd15_9tasks_rank128.yaml, 512x512 (latent 64x64), per-rank random task per step, synthetic")


def pretrain_bench(device, dtype, B=8, steps=9, warmup=9, tiny=False):
    """BASELINE.json configs[3] on ONE GPU (not the headline metric; evidence that Base-ControlNet pre-training runs at
    full width): ctrlora_pretrain_sd15_9tasks_rank128.yaml, every ControlNet weight + the step's task bank trained,
    the task changes every step as BatchSchedulerSampler makes it (datasets/multi_task_scheduler.py), eager launches
    (the bank switch re-packs that bank's LoRA copies between steps), PretrainAdamW with the reference's torch 1.13
    zero_grad semantics.  Synthetic latents, random-init weights, latent hint."""
    model = build_model("ctrlora_pretrain_sd15_9tasks_rank128.yaml", 0, tiny=tiny).to(device).train()
    model.set_engine_dtype(dtype)
    model.learning_rate = 1e-5
    opt = model.configure_optimizers()
    tasks = list(model.control_model.tasks)
    data = synth(B, 64, model.control_model.context_dim, device, 4321, 2)

    def step(i):
        j = i % 2
        cond = {"c_crossattn": [data["ctx"][j]], "c_concat": [data["hint"][j]], "task": tasks[i % len(tasks)]}
        opt.zero_grad()
        loss, _ = model.p_losses(data["z"][j], cond, data["t"][j], noise=data["noise"][j])
        loss.backward()
        opt.step()
        return loss

    for i in range(warmup):              # every bank gets its first gradient: from here on all nine are live
        loss = step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warmup + i)
    torch.cuda.synchronize()
    dt_eager = (time.perf_counter() - t0) / steps
    assert torch.isfinite(loss)
    # the same step as per-task hipGraph replays (ctrlora_amd.train.GraphedPretrainStep): one pass captures the nine graphs
    # (each capture call runs its step eagerly), then `steps` timed replays in task round-robin
    launch, dt = "eager", dt_eager
    try:
        from ctrlora_amd.train import GraphedPretrainStep
        gstep = GraphedPretrainStep(model, opt, data["z"][0], data["ctx"][0], data["hint"][0], data["t"][0], data["noise"][0])
        run = lambda i: gstep(tasks[i % len(tasks)], data["z"][i % 2], data["ctx"][i % 2], data["hint"][i % 2], data["t"][i % 2],
                              data["noise"][i % 2])
        for i in range(len(tasks)):
            loss = run(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = run(len(tasks) + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        assert torch.isfinite(loss) and len(gstep.graphs) == len(tasks)
        launch = f"hipGraph replay, one graph per task ({len(gstep.graphs)} graphs, one memory pool)"
    except Exception as e:
        print(f"[bench] pre-training graph capture failed ({type(e).__name__}: {e}); reporting the eager figure", file=sys.stderr)
    ntrain = sum(p.numel() for p in model.control_model.parameters())
    # algorithmic FLOPs per image: the fine-tuning figure (forward CN + UNet, data gradients) + the ControlNet's dense weight
    # gradients (every conv / linear of the ControlNet once more: ~F_cn = 0.30 TF) instead of LoRA-only ones
    tf_img = 1.996 + 0.30
    return dict(metric="Base-ControlNet multi-task pre-training images/sec (1 GPU)", value=round(B / dt, 2), launch=launch,
                eager_ms_per_step=round(dt_eager * 1e3, 2), mfma_frac=round(tf_img * B / dt / PEAK_BF16_TFLOPS, 4),
                tflop_per_image=tf_img,
                unit="images/s", ms_per_step=round(dt * 1e3, 2), batch=B, tasks=len(tasks), steps=steps, warmup=warmup,
                trainable_params_M=round(ntrain / 1e6, 1), loss=round(float(loss.detach()), 5),
                peak_mem_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), dtype=str(dtype).replace("torch.", ""),
                config="ctrlora_pretrain_sd15_9tasks_rank128.yaml, 512x512 (latent 64x64), task round-robin, synthetic")


def cpu_baseline(rank_lora=32, threads=32):
    """The oracle (CPU restatement of the reference modules, oracle/ref_model.py) on the host cores, TIMED, not
    extrapolated: ONE genuine optimizer step of BASELINE.json configs[0] -- ctrlora_finetune_sd15_rank32, bs = 1,
    512x512 (latent 64x64), fp32: q_sample + ControlNet(r32) + UNet forward, autograd backward to the 246 trainable
    tensors, AdamW on them -- and TWO DDIM denoise steps with classifier-free guidance (4 forwards) at bs = 1.
    Bounded sample (~20-40 s of host work); the reference itself additionally recomputes activations
    (checkpoint()) and forms ~0.9 G dead weight gradients, so it is slower than this port."""
    from oracle import arch, ref_model as R
    n_thr = max(1, min(threads, cpus_available()))
    torch.set_num_threads(n_thr)
    cfg = arch.ArchCfg(lora_rank=rank_lora)
    sd_cn = arch.make_state(arch.controlnet_shapes(cfg), 0)
    sd_un = arch.make_state(arch.unet_shapes(cfg), 0)
    train = [k for k in sd_cn if arch.is_trainable(k)]
    for k in train:
        sd_cn[k].requires_grad_(True)
    g = torch.Generator().manual_seed(0)
    z, hint, noise = (torch.randn(1, 4, 64, 64, generator=g) for _ in range(3))
    ctx, ctx_u = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
    t = torch.randint(0, 1000, (1,), generator=g)
    sched = R.make_schedule()
    t0 = time.perf_counter()
    loss, _ = R.p_losses(sd_cn, sd_un, cfg, sched, z, t, ctx, hint, noise)
    loss.backward()
    with torch.no_grad():
        for k in train:
            p, _, _ = R.adamw_step(sd_cn[k], sd_cn[k].grad, torch.zeros_like(sd_cn[k]), torch.zeros_like(sd_cn[k]), 1, 1e-5)
            sd_cn[k].copy_(p)
    step_s = time.perf_counter() - t0
    assert torch.isfinite(loss)

    def eps_fn(x, tt, c):
        with torch.no_grad():
            return R.apply_model(sd_cn, sd_un, cfg, x, tt, ctx if c else ctx_u, hint)

    S = 2
    t0 = time.perf_counter()
    x, _ = R.ddim_sample(eps_fn, sched, S, noise, scale=7.5, uncond=True)
    ddim_s = (time.perf_counter() - t0) / S
    assert torch.isfinite(x).all()
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    base = dict(cores=n_thr, kind="port", cpu=cpu)
    train_b = dict(value=round(1.0 / step_s, 5), unit="images/s", **base,
                   sample=f"ONE timed optimizer step of the oracle: ctrlora_finetune_sd15_rank{rank_lora}, bs 1, 512x512 "
                          f"(latent 64x64), fp32, p_losses forward + autograd backward (246 trainable tensors) + AdamW, "
                          f"{n_thr} threads: {step_s:.1f} s (hint latent given: no VAE / CLIP in the sample)")
    ddim_b = dict(value=round(1.0 / ddim_s, 5), unit="denoise steps/s", **base,
                  sample=f"{S} timed DDIM steps of the oracle with CFG 7.5 (2 forwards each), bs 1, latent 64x64, fp32, "
                         f"{n_thr} threads: {ddim_s:.1f} s per step")
    return train_b, ddim_b


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: re-run this file under torch.distributed.run, one rank per GPU of this
    node (what the reference's `Trainer(strategy='ddp', devices=-1)` does from one command,
    scripts/train_ctrlora_finetune.py:122-126).  Rank 0's JSON line is passed through.  If the hipGraph-replay data-parallel
    path dies (first contact of segment graphs with RCCL is the likeliest failure: capture invalidated by the watchdog
    thread, a collective timing out), the run is repeated ONCE with --no-graph (eager launches, bucketed all-reduces issued
    from the backward hooks) so that a number is still measured; the line then says `"launch": "eager"`."""
    import subprocess

    def run(extra):
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__), *argv, *extra]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        return r.returncode, lines, r.stdout

    rc, lines, out = run([])
    if (rc != 0 or not lines) and "--no-graph" not in argv:
        print(f"[bench] {n}-rank run failed (rc {rc}); retrying once with --no-graph (eager data parallel)", file=sys.stderr)
        rc, lines, out = run(["--no-graph"])
    if lines:
        print(lines[-1])
    else:
        sys.stdout.write(out)
    return rc if lines else (rc or 1)


def dist_dry_run(args, world, rank, local):
    """The launch path of `--gpus N` without the model (`--dist-dry-run`): process group on 127.0.0.1 (RCCL when the host has
    GPUs, gloo otherwise), W warm-up + K timed "steps" that all-reduce one 4 MB bucket asynchronously and wait for it -- the
    exchange pattern of ctrlora_amd.parallel.GradAllReduce -- bracketed exactly like the measured region (synchronize, barrier,
    MAX over ranks), one JSON line from rank 0.  Cheap first contact for a multi-GPU node; runs on the CPU in the test suite."""
    import datetime
    gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if gpu else torch.device("cpu")
    if gpu:
        torch.cuda.set_device(local)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    multi = world > 1 or "RANK" in os.environ
    if multi:
        kw = dict(device_id=device) if gpu else {}
        dist.init_process_group("nccl" if gpu else "gloo", timeout=datetime.timedelta(seconds=120), **kw)
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    bucket = torch.empty(1 << 20, dtype=torch.float32, device=device)

    def step():
        bucket.fill_(float(rank + 1))
        if multi:
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, async_op=True).wait()

    for _ in range(args.warmup):
        step()
    sync()
    if multi:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if multi:
        dist.barrier()
    sync()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    per_rank = [float(dt) / args.steps * 1e3]
    if multi:
        allt = [torch.zeros_like(dt) for _ in range(dist.get_world_size())]
        dist.all_gather(allt, dt)
        per_rank = [float(x) / args.steps * 1e3 for x in allt]
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    want = world * (world + 1) / 2 if multi else float(rank + 1)
    ok = bool(torch.all(bucket == want))
    okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    if multi:
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"metric": "launch-path dry run (no model): 4 MB all-reduce per step", "value": round(args.steps / float(dt), 3),
                          "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(dt) / args.steps * 1e3, 3), "dry_run": True,
                          "world": (dist.get_world_size() if multi else 1), "ms_per_step_by_rank": [round(v, 3) for v in per_rank],
                          "backend": (dist.get_backend() if multi else "none"), "device": device.type,
                          "allreduce_sum_correct_on_every_rank": bool(int(okt))}))
    if multi:
        dist.destroy_process_group()
    return 0 if int(okt) else 1


_T0 = time.perf_counter()


def cpus_available() -> int:
    """CPUs this process may actually use: the smaller of its affinity mask and its cgroup CPU quota (a GPU box of the pool
    reports 256 CPUs to os.cpu_count() and to the affinity mask under a quota of 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:   # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _tick(label):
    """CTRLORA_BENCH_TRACE_TIMES=1: wall-clock marks on stderr (where does a launch spend its time outside the timed region?)."""
    if os.environ.get("CTRLORA_BENCH_TRACE_TIMES") == "1":
        print(f"[bench +{time.perf_counter() - _T0:7.1f} s] {label}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--rank-lora", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ddim", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--ddim-only", action="store_true", help="profiling aid: run only the DDIM leg")
    ap.add_argument("--ddim-loops", type=int, default=5, help="timed S = 50 loops of the DDIM leg")
    ap.add_argument("--ddim-warm", type=int, default=10, help="warm-up loops of the DDIM leg")
    ap.add_argument("--ddim-core-only", action="store_true", help="profiling aid: skip the cold-call and image-hint DDIM legs")
    ap.add_argument("--force-split-graphs", action="store_true",
                    help="test aid: use the multi-rank structure (segment graphs with bucketed all-reduces in between, "
                         "then the AdamW graph) even with one rank")
    ap.add_argument("--probe-only", action="store_true",
                    help="profiling aid: run only the dominant-kernel probe (the rocprofv3 --stats summary of this "
                         "command, profiles/r01_dominant_kernel_stats.csv, is what roofline.ms_per_launch is checked against)")
    ap.add_argument("--pretrain-only", action="store_true",
                    help="evidence run, not the headline: one-GPU Base-ControlNet pre-training steps (BASELINE configs[3])")
    ap.add_argument("--tiny", action="store_true", help="debug: narrow model")
    ap.add_argument("--dist-dry-run", action="store_true",
                    help="launch-path check without the model: the ranks are started, rendezvous on 127.0.0.1, all-reduce a 4 MB "
                         "bucket per step inside the same barrier / MAX-over-ranks bracket as the timed region, rank 0 prints one "
                         "JSON line.  RCCL on GPUs, gloo on a CPU-only host (tests/test_bench_spawn.py runs it with 2 ranks)")
    ap.add_argument("--tag-gemm", default=None, metavar="JSON",
                    help="profiling aid: tag every contraction launch with its product signature (extra empty workgroups, "
                         "csrc/debug_hooks.h) and write the tag table here -- tools/prof_shapes.py joins it with a rocprofv3 "
                         "kernel trace of THIS run into a per-shape table of the replayed step")
    args = ap.parse_args()
    if args.tag_gemm:
        import atexit
        from ctrlora_amd import hip as _hip
        _hip.lib().cl_debug_gemm_tag(1)
        atexit.register(lambda: json.dump(_hip.gemm_tags(), open(args.tag_gemm, "w")))

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))        # `python bench.py --gpus N` starts its own ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.dist_dry_run:
        return dist_dry_run(args, world, rank, local)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world == 1 and "RANK" not in os.environ:
        torch.set_num_threads(max(1, min(torch.get_num_threads(), 16, cpus_available())))   # host side of the model build
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        # N ranks build their 1.3 G-parameter model on the host at the same time: share the cores this process may actually
        # run on (cpus_available(): affinity mask and cgroup quota, not os.cpu_count() -- under a 16-CPU quota cpu_count() = 256 threads
        # per rank spin against each other -- a ONE-rank torchrun launch took 476 s to build the model that a plain launch
        # builds in 10 s, profiles/r06_verify/torchrun_timing.txt) and cap the share: the build does not scale past ~16 threads
        avail = cpus_available()
        _tick(f"cpu_count {os.cpu_count()}, usable {avail}, torch threads before {torch.get_num_threads()}, "
              f"OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')}")
        torch.set_num_threads(max(1, min(16, avail // world)))
        _tick("init_process_group ...")
        dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=900))
        _tick("process group up")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    def _leave():      # (single-leg runs under a rank environment: shut the process group down like the main path does)
        if dist.is_initialized():
            dist.destroy_process_group()

    if args.ddim_only:
        print(json.dumps(ddim_bench(device, dtype, tiny=args.tiny, loops=args.ddim_loops, warm_loops=args.ddim_warm,
                                    extras=not args.ddim_core_only)))
        return _leave()
    if args.probe_only:
        print(json.dumps(conv_kernel_probe(device, dtype, iters=200)))
        return _leave()
    if args.pretrain_only:
        if world > 1:
            out = pretrain_bench_dp(device, dtype, world, rank, B=args.batch, tiny=args.tiny)
            if rank == 0:
                print(json.dumps(out))
            dist.barrier()
            dist.destroy_process_group()
            return
        print(json.dumps(pretrain_bench(device, dtype, B=args.batch, tiny=args.tiny)))
        return _leave()

    model = build_model(f"ctrlora_finetune_sd15_rank{args.rank_lora}.yaml", 0, tiny=args.tiny).to(device).train()
    model.set_engine_dtype(dtype)
    model.learning_rate = 1e-5
    _tick("model built")
    if world > 1:
        from ctrlora_amd.parallel import GradAllReduce
        model.dp = GradAllReduce([model.control_model.executor()])
    opt = model.configure_optimizers()
    B, H = args.batch, 64
    n_in = 4
    data = synth(B, H, model.control_model.context_dim, device, 1234 + rank, n_in)

    graphed = None
    if not args.no_graph:
        # the whole step (zero_grad, forward, hand-written backward, AdamW, re-pack) as hipGraph replays;
        # with N > 1 ranks the gradient all-reduce runs between two graphs (ctrlora_amd/train.py)
        from ctrlora_amd.train import GraphedTrainStep
        try:
            graphed = GraphedTrainStep(model, opt, data["z"][0], data["ctx"][0], data["hint"][0], data["t"][0],
                                       data["noise"][0], split_graphs=True if args.force_split_graphs else None)
        except Exception as e:   # capture is an optimisation: never lose the measurement to it
            print(f"[bench] rank {rank}: hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches",
                  file=sys.stderr)
            graphed = None
        if world > 1:
            # the ranks must agree: segment graphs and the eager hooks cut the gradient buffer into different buckets, so a
            # mixed job would dead-lock in its first mismatched collective
            okf = torch.tensor([0 if graphed is None else 1], device=device, dtype=torch.int32)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf) == 0:
                graphed = None
        if graphed is None:
            args.no_graph = True
            if model.dp is not None:
                model.dp.enabled = True
                opt.pre_step_hook = model.dp.wait

    def step(i):
        j = i % n_in
        if graphed is not None:
            return graphed(data["z"][j], data["ctx"][j], data["hint"][j], data["t"][j], data["noise"][j])
        opt.zero_grad()
        cond = {"c_crossattn": [data["ctx"][j]], "c_concat": [data["hint"][j]]}
        loss, _ = model.p_losses(data["z"][j], cond, data["t"][j], noise=data["noise"][j])
        loss.backward()
        opt.step()
        return loss

    _tick("step ready (graph captured)" if graphed is not None else "step ready (eager)")
    for i in range(args.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _tick("warm-up done")
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _tick("timed region done")
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    final_loss = float(loss)
    assert final_loss == final_loss, "loss is NaN"

    dp_info = None
    if world > 1:
        # what the exchange costs: the same K steps with the gradient all-reduce switched off (every rank then keeps its own
        # gradients -- measurement only, AFTER the timed region); exposed = with - without
        ex0 = model.control_model.executor()
        dp_info = {"payload_MB_per_step": round(ex0.tr.numel * 4 / 1e6, 1), "collective": "RCCL all-reduce (SUM), fp32, "
                   "LoRA + zero-conv + norm gradients only (flat buffer in backward-completion order)"}
        try:
            if graphed is not None:
                saved_fn, graphed._reduce_fn = graphed._reduce_fn, (lambda buf: None)
                dp_info.update(segments=len(graphed.segments), bucket_MB=32)
            else:
                model.dp.enabled = False
                opt.pre_step_hook = None
            for i in range(2):
                step(i)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step(i)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t_no = torch.tensor([time.perf_counter() - t1], device=device, dtype=torch.float64)
            dist.all_reduce(t_no, op=dist.ReduceOp.MAX)
            ms_no = float(t_no) / args.steps * 1e3
            dp_info.update(ms_per_step_without_exchange=round(ms_no, 2),
                           exposed_allreduce_ms_per_step=round(dt / args.steps * 1e3 - ms_no, 2))
            if graphed is not None:
                graphed._reduce_fn = saved_fn
        except Exception as e:
            print(f"[bench] exchange-off measurement failed on rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)

    if rank == 0:
        ips = world * B * args.steps / dt
        tf_img = TRAIN_TFLOP_PER_IMAGE.get(args.rank_lora, 1.996)
        out = {
            "metric": "512x512 LoRA-finetune images/sec", "value": round(ips, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"ctrlora_finetune_sd15_rank{args.rank_lora}.yaml, canny-style latent hint, "
                                   f"512x512 (latent 64x64), per-GPU batch {B}, {args.dtype} storage / fp32 accumulate, "
                                   "synthetic latents + random-init weights, LoRA+zero-conv+norm trainables, fused AdamW",
                       "global_batch": world * B, "parallelism": f"dp{world}", "lora_rank": args.rank_lora,
                       "launch": "eager" if args.no_graph else ("hipGraph replay" if graphed is None or graphed.mode == "one" else
                                                                 f"hipGraph replay, {len(graphed.segments)} backward segments with the "
                                                                 "LoRA-gradient all-reduce of each bucket overlapping the next segment"),
                       "gemm_launch_table_entries": int(__import__("ctrlora_amd.hip", fromlist=["lib"]).lib().cl_gemm_tune_size())},
            "loss": round(final_loss, 5),
        }
        if dp_info is not None:
            out["config"]["data_parallel"] = dp_info
        if world == 1 and args.rank_lora == 128 and args.dtype == "bf16" and not args.tiny and B == 8:
            # context, not credit: the reference's own modules on PyTorch-ROCm eager kernels, bf16 autocast, same workload and
            # GPU model (tests/tools/compare_stock.py -> profiles/r02_compare_precision.json); not re-measured in this run
            stock_ips, stock_src = stock_reference()
            try:
                with open(os.path.join(ROOT, stock_src)) as f:
                    smeta = json.load(f)
            except Exception:
                smeta = None
            out["vs_stock"] = {"value": round(ips / stock_ips, 2), "stock_images_per_s": stock_ips,
                               "provenance": dict(measured_at_commit=(smeta or {}).get("measured_at_commit"),
                                                  stale=not bool((smeta or {}).get("measured_at_commit")),
                                                  note="the comparator runs PyTorch-ROCm's kernels, not this tree's: stale = no commit recorded"),
                               "kind": f"static: {stock_src} (reference modules, torch.autocast(bf16), PyTorch-ROCm eager, B=8, "
                                       "1x MI355X; tests/tools/compare_stock.py)"}
        achieved = tf_img * ips / world          # per-GPU TFLOP/s
        roof = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None,
                "basis": f"{tf_img} TFLOP/image algorithmic (SURVEY.md 8d), per GPU"}
        roof["whole_step"] = {"achieved": roof["achieved"], "frac": roof["frac"], "basis": roof.pop("basis")}
        if not args.tiny and args.dtype == "bf16":   # rank 0 only; no collective inside
            dk = conv_kernel_probe(device, dtype)
            fam = family_census(model, opt, data)
            # Contract fields = the DOMINANT KERNEL of the step (gemm_fl_kernel<256 x 160, conv-s1>: 14 % of the step's GPU time,
            # profiles/r04_prof/train_kernel_stats_steady.txt) at its dominant shape, timed live with HIP events on the launching
            # stream: the figure the rocprofv3 --stats row of the same launch has to agree with (profiles/r04_prof/
            # dominant_kernel_stats.csv).  The time-weighted figure of the whole GEMM family, the attention and the HBM-bound
            # families and the whole step stay alongside -- the whole-step fraction is the honest summary of the step.
            roof.update(achieved=dk["achieved"], frac=dk["frac"], traffic=dk["traffic"], kernel=dk["kernel"],
                        ms_per_launch=dk["ms"], algorithmic_bytes_per_launch=43_800_000,
                        frac_of="the DOMINANT KERNEL at its dominant shape (16 % of the step's GPU time), 30 hot back-to-back "
                                "launches between HIP events; the step as a whole is roofline.whole_step, the time-weighted "
                                "contraction family roofline.family -- since round 4 (rounds 1-3 reported the family figure here)",
                        rocprof=dominant_kernel_rocprof(),
                        traffic_provenance=dk.get("traffic_provenance"),
                        traffic_kind="rocprofv3 TCC passes on this launch, FETCH_SIZE (x2: gfx950 correction) + WRITE_SIZE "
                                     "(profiles/dominant_kernel_traffic.json names the commit and the kernel-source hash it was taken on; "
                                     "`traffic_provenance.stale` = csrc/gemm.hip has changed since); static in this run")
            gf = fam.get("gemm")
            if gf:
                roof["family"] = dict(gf, kernel="gemm_fl / gemm kernel family (implicit-GEMM 3x3 conv + linear + LoRA-fused linear), "
                                                 "time-weighted over all shapes of one step")
            if "attention" in fam:
                roof["attention_family"] = fam["attention"]
            if "hbm" in fam:
                roof["norm_elementwise_family"] = fam["hbm"]
        out["roofline"] = roof
    if rank == 0 and not args.tiny and args.dtype == "bf16" and not args.no_vae:
        try:   # end to end = the core step + the first-stage encodes the reference performs inside every step
            vb = vae_bench(device, dtype, B)
            step_ms = dt / args.steps * 1e3
            out["end_to_end"] = dict(vae_encode=vb, ms_per_step=round(step_ms + vb["ms"], 2),
                                     images_per_s_per_gpu=round(B / (step_ms + vb["ms"]) * 1e3, 2),
                                     note="core step (value) + VAE encode of the B target and B condition images on the engine, "
                                          "timed separately and added; CLIP text encoding (~1 % of the FLOPs) not included")
        except Exception as e:
            print(f"[bench] VAE leg failed: {type(e).__name__}: {e}", file=sys.stderr)
    # DDIM leg: every rank samples its own batch (replicas, no collective); aggregate = sum over ranks
    ddim = None
    if not args.no_ddim:
        del graphed, model, opt
        torch.cuda.empty_cache()
        try:
            ddim = ddim_bench(device, dtype, tiny=args.tiny, loops=args.ddim_loops, warm_loops=args.ddim_warm)
            if world > 1:
                tt = torch.tensor([ddim["S"] / ddim["value"]], device=device, dtype=torch.float64)   # seconds per loop
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ddim["value"] = round(world * ddim["S"] / float(tt), 3)
                ddim["ms_per_step"] = round(float(tt) / ddim["S"] * 1e3, 2)
                ddim["mfma_frac"] = round(DDIM_TFLOP_PER_STEP_IMAGE * ddim["batch"] * ddim["value"] / world / PEAK_BF16_TFLOPS, 4)
                ddim["note"] += f"; {world} independent replicas (one batch of {ddim['batch']} per GPU), value = sum"
        except Exception as e:
            print(f"[bench] DDIM leg failed on rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)
            ddim = None
            if world > 1:   # keep the ranks in step
                tt = torch.tensor([0.0], device=device, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        if ddim is not None:
            out["ddim"] = ddim
        if world == 1 and not args.no_cpu_baseline and not args.tiny:
            train_b, ddim_b = cpu_baseline()
            out["cpu_baseline"] = train_b
            if ddim is not None:
                out["ddim"]["cpu_baseline"] = ddim_b
        print(json.dumps(out))
    _tick("line printed")
    if dist.is_initialized():
        dist.destroy_process_group()
    _tick("process group destroyed")


if __name__ == "__main__":
    main()
