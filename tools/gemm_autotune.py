"""Measured launch table for the contraction kernels (csrc/gemm.hip).

Records every `hip.gemm` call of the workloads the benchmarks run -- one LoRA fine-tuning step (rank 128, B = 8,
latent 64x64), one CFG DDIM denoise step (B = 16 -> 32 rows through both networks), one VAE encode / decode at 512x512,
optionally one pre-training step -- and, for each distinct product signature (dtype, mode, M, N, K1, K2, GEGLU), times
the tile configurations and split-K factors the launcher accepts for it, in isolation, between HIP events on the
launching stream.  A candidate replaces the built-in choice only if it is >= `--gain` faster in two separate
measurements and its result agrees with the built-in launch's.  Output: ctrlora_amd/gemm_tuned_gfx950.json, which
`ctrlora_amd.hip.lib()` registers through `cl_gemm_tune_set` at start-up.

Run on the GPU box with the table disabled:   CTRLORA_GEMM_TUNED=0 python tools/gemm_autotune.py [--quick]
"""
import argparse
import collections
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CTRLORA_GEMM_TUNED"] = "0"
import bench  # noqa: E402
from ctrlora_amd import hip  # noqa: E402

FL = (10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 31, 32, 33, 35, 36)   # full-line (LDS-DMA, 128-byte K lines) configurations
W80 = (31, 32, 35, 36)                                           # 128 x 80 tiles (linear products, N % 80 == 0)
W320 = (33,)                                             # 128 x 320 full-N tiles (linear products, N % 320 == 0)
PERSIST = (25, 26, 27, 28, 29, 30)                      # persistent forms of 16 / 17 / 20 / 21 / 10 / 11 (linear only)
W160 = (2, 5, 10, 12, 14, 16, 18, 20, 23, 25, 27, 29)   # 160-column tiles
W128 = (1, 7, 11, 13, 15, 17, 19, 21, 22, 26, 28, 30)   # 128-column tiles
GEGLU_OK = (2, 10, 12, 14, 16, 18, 20, 23, 25, 27, 29, 40, 44, 47)  # a value / gate wave pair per 160-column tile (40 / 47: in-register pairing)
DEEP = (42, 43, 44, 45, 46)                              # generic kernel, 8-slot ring: 64x64 / 64x128 / 64x160 / 128x64 / 128x128
W4 = (40, 41, 47, 48)                                    # loader / consumer kernel (gemm_w4.hip): 256 x 160 / 256 x 128 tiles; 47 / 48 = persistent


class Recorder:
    """Wraps hip.gemm: first call of every signature is kept (tensors stay alive through the closure)."""

    def __init__(self):
        self.calls = collections.OrderedDict()
        self.tag = ""

    def __enter__(self):
        self.orig = hip.gemm

        def rec(a1, w1, out, **kw):
            M = out.shape[0] if kw.get("M") is None else kw["M"]
            N = out.shape[1] if kw.get("N") is None else kw["N"]
            k1 = a1.shape[1] if kw.get("k1") is None else kw["k1"]
            mode, a2 = kw.get("mode", hip.LINEAR), kw.get("a2")
            # grouped second segment (a2_group_n): the launcher's signature carries the per-group K2 = columns of W2
            k2 = 0 if a2 is None else (kw["w2"].shape[1] if kw.get("a2_group_n") else a2.shape[1])
            geglu = 1 if kw.get("act", 0) == hip.ACT_GEGLU else 0
            if not kw.get("atomic", False) and kw.get("act", 0) != hip.ACT_GEGLU_SPLIT:   # (act 3 has one kernel: nothing to choose)
                key = (hip.dt(a1), int(mode), int(M), int(N), int(k1), int(k2), geglu)
                e = self.calls.get(key)
                if e is None:
                    # private copies of the output (and of an in-place residual) so that re-launching is idempotent
                    o2 = torch.empty_like(out)
                    kw2 = dict(kw)
                    if kw.get("residual") is not None:
                        kw2["residual"] = kw["residual"].clone()
                    e = self.calls[key] = dict(n={}, run=lambda a1=a1, w1=w1, o2=o2, kw2=kw2: self.orig(a1, w1, o2, **kw2),
                                               out=o2)
                e["n"][self.tag] = e["n"].get(self.tag, 0) + 1
            return self.orig(a1, w1, out, **kw)

        hip.gemm = rec
        return self

    def __exit__(self, *a):
        hip.gemm = self.orig


def time_us(run, reps):
    """Average duration of one launch inside a replayed hipGraph of `reps` back-to-back launches: what the launch
    costs in the captured training / DDIM step (an eager loop from Python measures the ~9 us host launch path
    instead for every product below that)."""
    run()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay(); g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (2 * reps)
    del g
    return us


def candidates(key):
    _, mode, M, N, K1, K2, geglu = key
    taps = 1 if mode == hip.LINEAR else 9
    fl_ok = K1 % 64 == 0 and K2 % 64 == 0 and not (mode != hip.LINEAR and K2) and M > 128 and N >= 96
    cfgs = [0, 24] if N % 64 == 0 else [0]
    for c in (1, 2, 5, 7, 22, 23) + FL + PERSIST + W4 + DEEP:
        if c in DEEP and ((c == 44 and N % 160) or (c in (43, 46) and N % 128 and N % 160 == 0) or (c in (42, 45) and N % 64) or M * N > 8192 * 2560):
            continue
        if c in W4 and (not fl_ok or (c in (40, 47) and N % 160) or (c in (41, 48) and N % 128 and N % 160 == 0) or M < 1024):
            continue
        if c in (47, 48) and M * N < 2 * 256 * 256 * 160:
            continue                      # persistent: only where a workgroup would own several tiles
        if c in FL and not fl_ok:
            continue
        if c in PERSIST and not (fl_ok and mode == hip.LINEAR and M * N >= 256 * 160 * 512):
            continue                      # persistent walk only where a workgroup would own several tiles
        if c in W160 and N % 160:
            continue
        if c in W80 and (N % 80 or (mode != hip.LINEAR and c not in (35, 36)) or M > 16384):
            continue
        if c in (35, 36) and mode != hip.LINEAR and M > 2048:
            continue                      # conv modes: the small-M levels only (8x8, 16x16)
        if c in W320 and (N % 320 or mode != hip.LINEAR or M < 8192):
            continue
        if c in W128 and N % 128 and N % 160 == 0:
            continue                      # keep the tile width the heuristic would use for this N
        cfgs.append(c)
    if mode == hip.LINEAR and K1 in (320, 640) and K2 in (0, 128) and N % 32 == 0 and M >= 128:
        cfgs.append(34)                   # x-stationary streaming kernel (gemm_xs.hip); its split column = column runs per group
    if geglu:
        cfgs = [c for c in cfgs if c in GEGLU_OK]
    steps = (taps * K1 + K2) // 64
    sks = [0, 1] + [s for s in (2, 3, 4, 6, 8, 12, 16) if s * 2 <= steps]
    return cfgs, sks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="training step only")
    ap.add_argument("--ranks", default="128", help="comma-separated LoRA ranks whose training step is recorded "
                    "(configs/ctrlora_finetune_sd15_rank<r>.yaml); DDIM / VAE workloads use rank 128")
    ap.add_argument("--merge", default=None, help="existing table: its entries are kept, only NEW signatures are searched")
    ap.add_argument("--only-new", action="store_true", help="with --merge: search every signature that is not in the table "
                    "(no rank filter) -- e.g. after the engine started issuing new product shapes")
    ap.add_argument("--retry-cfgs", default="", help="with --merge: comma list of NEW tile configurations to offer to signatures "
                    "that already have an entry (or keep the rules): only these (at split 0, 1, 2) are timed against the current choice")
    ap.add_argument("--budget-s", type=float, default=0.0, help="stop searching after this many seconds (0 = no limit) and write "
                    "the table with what was found so far (GPU minutes are metered)")
    ap.add_argument("--gain", type=float, default=0.03)
    ap.add_argument("--reps", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "ctrlora_amd", "gemm_tuned_gfx950.json"))
    ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "gemm_autotune.log"))
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dtype = torch.bfloat16
    L = hip.lib()
    L.cl_gemm_tune_clear()
    rec = Recorder()

    # ---- workloads -------------------------------------------------------------------------------------------
    kept = []
    if args.merge and os.path.exists(args.merge):
        with open(args.merge) as f:
            kept = [list(map(int, r)) for r in json.load(f)["entries"]]
    known = {tuple(r[:7]) for r in kept}
    rank_set = {int(r) for r in args.ranks.split(",")}
    for rank in [int(r) for r in args.ranks.split(",")]:
        model = bench.build_model(f"ctrlora_finetune_sd15_rank{rank}.yaml", 0).to(device).train()
        model.set_engine_dtype(dtype)
        model.learning_rate = 1e-5
        opt = model.configure_optimizers()
        data = bench.synth(8, 64, model.control_model.context_dim, device, 1234, 1)
        cond = {"c_crossattn": [data["ctx"][0]], "c_concat": [data["hint"][0]]}
        opt.zero_grad()
        model.engine_train_step(data["z"][0], cond, data["t"][0], data["noise"][0])
        torch.cuda.synchronize()
        with rec:
            rec.tag = "train" if rank == 128 else f"train_r{rank}"
            opt.zero_grad()
            model.engine_train_step(data["z"][0], cond, data["t"][0], data["noise"][0])
            torch.cuda.synchronize()
        del model, opt
        torch.cuda.empty_cache()
    if not args.quick:
        from cldm.ddim_hacked import DDIMSampler
        minf = bench.build_model("inference/ctrlora_sd15_rank128_1lora.yaml", 0).to(device).eval()
        minf.set_engine_dtype(dtype)
        cd, B, H = minf.control_model.context_dim, 16, 64
        g = torch.Generator().manual_seed(7)
        hint = torch.randn(B, 4, H, H, generator=g).to(device)
        c = {"c_concat": [hint], "c_crossattn": [torch.randn(B, 77, cd, generator=g).to(device)]}
        u = {"c_concat": [hint], "c_crossattn": [torch.randn(B, 77, cd, generator=g).to(device)]}
        sampler = DDIMSampler(minf)
        sampler.use_graph = False
        with rec:
            rec.tag = "ddim"
            sampler.sample(2, B, (4, H, H), c, verbose=False, eta=0.0, unconditional_guidance_scale=7.5,
                           unconditional_conditioning=u)
            torch.cuda.synchronize()
        for k in rec.calls.values():               # two denoise steps were recorded
            if "ddim" in k["n"]:
                k["n"]["ddim"] = max(1, k["n"]["ddim"] // 2)
        del minf, sampler
        from ldm.models.autoencoder import AutoencoderKL
        dd = dict(attn_resolutions=[], ch=128, ch_mult=[1, 2, 4, 4], double_z=True, dropout=0.0, in_channels=3,
                  num_res_blocks=2, out_ch=3, resolution=256, z_channels=4)
        torch.manual_seed(0)
        vae = AutoencoderKL(ddconfig=dd, lossconfig=dict(target="torch.nn.Identity"), embed_dim=4).to(device).eval()
        vae.engine_dtype = dtype
        with torch.no_grad(), rec:
            rec.tag = "vae_enc"
            post = vae.encode(torch.rand(16, 3, 512, 512, device=device) * 2 - 1)
            rec.tag = "vae_dec"
            vae.decode(post.mean[:4].contiguous())
            torch.cuda.synchronize()
        del vae
    torch.cuda.empty_cache()

    # ---- search ----------------------------------------------------------------------------------------------
    os.makedirs(os.path.dirname(args.log), exist_ok=True)
    log = open(args.log, "w")
    entries, rows = [], []
    t_start = time.time()
    # heaviest signatures first, so that a time budget cuts the tail
    order = sorted(rec.calls.items(), key=lambda kv: -sum(kv[1]["n"].values()) * kv[0][2] * kv[0][3] * (kv[0][4] + kv[0][5]))
    for key, e in order:
        if args.budget_s and time.time() - t_start > args.budget_s:
            log.write(f"time budget of {args.budget_s:.0f} s reached: {key} and the lighter signatures keep their current launch\n")
            break
        retry = [int(c) for c in args.retry_cfgs.split(",") if c]
        only = None
        if key in known or (args.merge and args.retry_cfgs and key not in known and not args.only_new):
            if not retry:
                continue                                 # --merge: measured before, entry kept as it is
            cfgs0, _ = candidates(key)
            only = [c for c in retry if c in cfgs0]
            if not only:
                continue
        elif args.merge and not args.only_new and not (key[5] or key[3] in rank_set or key[4] in rank_set):
            continue                                     # --merge: rank-independent signature, searched when the table was made
        run, out = e["run"], e["out"]
        cur = next((r for r in kept if tuple(r[:7]) == key), None)
        if only is not None and cur is not None:         # the current choice is the table's entry, not the rules
            L.cl_gemm_force_config(cur[7]); L.cl_gemm_force_splitk(cur[8])
        else:
            L.cl_gemm_force_config(-1); L.cl_gemm_force_splitk(0)
        run(); torch.cuda.synchronize()
        ref = out.float().clone()
        scale = float(ref.abs().max()) + 1e-20
        base = time_us(run, args.reps)
        cfgs, sks = candidates(key)
        if only is not None:
            cfgs, sks = only, ([0, 1, 2, 4] if 34 in only else ([0, 1, 2, 4, 8] if set(only) <= {35, 36} and key[1] != hip.LINEAR else [0, 1, 2]))
        best = (base, -1, 0)
        per_cfg = []
        for c in cfgs:                                   # tile configuration at the launcher's own split rule
            L.cl_gemm_force_config(c); L.cl_gemm_force_splitk(0)
            try:
                us = time_us(run, args.reps)
            except hip.HipError:
                continue
            per_cfg.append((us, c))
        per_cfg.sort()
        trials = [(us, c, 0) for us, c in per_cfg]
        for us, c in per_cfg[:3]:                        # split factors for the three fastest tiles
            for sk in sks[1:]:
                L.cl_gemm_force_config(c); L.cl_gemm_force_splitk(sk)
                try:
                    trials.append((time_us(run, args.reps), c, sk))
                except hip.HipError:
                    pass
        trials.sort()
        chosen = None
        for us, c, sk in trials[:4]:
            if us > base * (1.0 - args.gain):
                break
            L.cl_gemm_force_config(c); L.cl_gemm_force_splitk(sk)
            out.zero_()
            run(); torch.cuda.synchronize()
            err = float((out.float() - ref).abs().max()) / scale
            us2 = time_us(run, 2 * args.reps)
            if only is not None and cur is not None:
                L.cl_gemm_force_config(cur[7]); L.cl_gemm_force_splitk(cur[8])
            else:
                L.cl_gemm_force_config(-1); L.cl_gemm_force_splitk(0)
            base2 = time_us(run, 2 * args.reps)
            ok = err < 2e-2 and us2 < base2 * (1.0 - args.gain)
            log.write(f"  confirm {key} cfg {c} sk {sk}: {us2:.1f} vs {base2:.1f} us, err {err:.2e} -> {'take' if ok else 'drop'}\n")
            if ok:
                chosen = (us2, base2, c, sk)
                break
        L.cl_gemm_force_config(-1); L.cl_gemm_force_splitk(0)
        calls = sum(e["n"].values())
        if chosen:
            us2, base2, c, sk = chosen
            entries.append(list(key) + [c, sk])
            rows.append((calls * (base2 - us2), key, e["n"], base2, us2, c, sk))
        else:
            rows.append((0.0, key, e["n"], base, base, -1, 0))
        log.write(f"{key} calls {e['n']} base {base:.1f} us; best trials {[(round(u, 1), c, s) for u, c, s in trials[:4]]}\n")
        log.flush()
    rows.sort(key=lambda r: -r[0])
    saved = collections.Counter()
    for gain_us, key, n, b, u, c, sk in rows:
        for tag, cnt in n.items():
            saved[tag] += cnt * (b - u)
    summary = {tag: round(v * 1e-3, 3) for tag, v in saved.items()}
    print(f"signatures {len(rec.calls)}, tuned {len(entries)}, predicted saving (ms, isolated timings): {summary}, "
          f"search {time.time() - t_start:.0f} s")
    for gain_us, key, n, b, u, c, sk in rows[:40]:
        print(f"  {gain_us:8.0f} us  {key}  {n}  {b:7.1f} -> {u:7.1f} us  cfg {c} sk {sk}")
    # in-graph census of the training step under the chosen launches (per signature: calls x us, TF/s)
    cen = []
    for gain_us, key, n, b, u, c, sk in rows:
        if n.get("train"):
            taps = 1 if key[1] == hip.LINEAR else 9
            fl = 2.0 * key[2] * key[3] * (taps * key[4] + key[5])
            cen.append((n["train"] * u, n["train"], u, fl / u * 1e-6, key, c, sk))
    cen.sort(reverse=True)
    with open(os.path.join(os.path.dirname(args.log), "gemm_census_train_ingraph.txt"), "w") as f:
        f.write(f"training step, {len(cen)} signatures, {sum(r[0] for r in cen) * 1e-3:.2f} ms per step "
                "(launch time inside a replayed hipGraph, sum over calls)\n")
        f.write(f"{'tot_us':>8} {'n':>4} {'us':>8} {'TF/s':>7}  (dtype, mode, M, N, K1, K2, geglu)  cfg sk (-1 = built-in)\n")
        for t, cnt, us, tf, key, c, sk in cen:
            f.write(f"{t:8.0f} {cnt:4d} {us:8.1f} {tf:7.1f}  {key}  {c} {sk}\n")
    head = {"device": torch.cuda.get_device_name(0), "columns": "dtype mode M N K1 K2 geglu cfg splitk",
            "gain_threshold": args.gain, "predicted_saving_ms": summary}
    with open(args.out, "w") as f:          # one entry per line: reviewable diffs
        f.write("{" + ", ".join(f"{json.dumps(k)}: {json.dumps(v)}" for k, v in head.items()) + ',\n"entries": [\n')
        fresh = {tuple(r[:7]) for r in entries}           # a retried signature's new choice replaces its old entry
        f.write(",\n".join(json.dumps(r) for r in sorted([r for r in kept if tuple(r[:7]) not in fresh] + entries)))
        f.write("\n]}\n")
    log.close()


if __name__ == "__main__":
    main()
