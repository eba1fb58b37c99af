"""GPU parity, round 3: the holes VERDICT r2 named.

  * BASELINE.json configs[4] at ITS OWN shape: the inference executor (rank-128 inference YAML, SD1.5 width, latent
    64x64, bf16, LoRA folded into one packed weight, decoder K/V cached, CFG batched as 2B) against the oracle evaluated
    in fp32 on the GPU -- eps of one call, eps through the context-K/V cache, B = 16 as the bench runs it, and the sample
    after S = 4 DDIM steps with guidance 7.5 -- with gates taken from the same-precision comparator (the oracle under
    torch.autocast(bfloat16) through PyTorch-ROCm's kernels) measured in the same test;
  * the LoRA fold against the two-segment executor on the same weights, with a SMALL up-projection (1e-3 N(0,1): the
    update is below the bf16 quantum of the base weight) -- the fold must not be less accurate than not folding;
  * the conv-tap weight gradient (cl_wgrad_desc.tap) at production shapes vs fp64;
  * ONE Base-ControlNet pre-training step at SD1.5 width: every gradient of control_model (360 M base + the task's bank)
    vs the oracle on the GPU;
  * the bf16 error of the pre-training test does NOT grow when the parameters follow the fp32 trajectory (the growth seen
    in test_pretrain.py is parameter divergence under AdamW's sign-like first steps at lr 1e-3, not a kernel error).

Reference lines: cldm/ddim_hacked.py:181-231, cldm/cldm_ctrlora_inference.py:156-178, cldm/lora.py:285-318,
cldm/cldm_ctrlora_pretrain.py:95-111,174-182.  The oracle is the checker only.
"""

import pytest
import torch

from tests.util import rel_l2
from tests.test_gpu_bench_shapes import BF16_EPS, _bf, _need_gpu, _netcfg, _record

pytestmark = pytest.mark.gpu


def _oracle_eps(cfg, sd_cn, sd_un, x, t, ctx, hint, autocast=None):
    """oracle.apply_model on the GPU (stock kernels; fp32, or bf16 autocast = the same-precision comparator)."""
    from oracle import ref_model as R
    dev = torch.device("cuda")
    cn = {k: v.detach().to(dev) for k, v in sd_cn.items()}
    un = {k: v.detach().to(dev) for k, v in sd_un.items()}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast is not None):
        return R.apply_model(cn, un, cfg, x.to(dev), t.to(dev), ctx.to(dev), hint.to(dev)).float()


def _inference_model():
    import bench
    model = bench.build_model("inference/ctrlora_sd15_rank128_1lora.yaml", 0).cuda().eval()
    model.set_engine_dtype(torch.bfloat16)
    sd_cn = {k: v.detach().clone() for k, v in model.control_model.bank_state(0).items()}
    sd_un = {k: v.detach().clone() for k, v in model.model.diffusion_model.state_dict().items()}
    return model, sd_cn, sd_un


def test_inference_executor_sd15_latent64_eps_vs_oracle():
    """configs[4]: eps of the inference executor (folded LoRA) at B = 4 and B = 16, with and without the context K/V
    cache, vs the fp32 oracle; the bf16-autocast oracle is measured beside it."""
    _need_gpu()
    from oracle import arch
    cfg = arch.SD15
    model, sd_cn, sd_un = _inference_model()
    eng = model.engine()
    assert all(cn.merge_lora for cn in eng.controls) and eng.controls[0]._b.linears[0].W32 is not None
    g = torch.Generator().manual_seed(11)
    for B in (4, 16):
        x, hint = torch.randn(B, 4, 64, 64, generator=g).cuda(), (torch.randn(B, 4, 64, 64, generator=g) * 0.9).cuda()
        ctx = torch.randn(B, 77, cfg.context_dim, generator=g).cuda()
        t = torch.randint(0, 1000, (B,), generator=g).cuda()
        cond = {"c_concat": [hint], "c_crossattn": [ctx]}
        eps = model.apply_model(x, t, cond)
        ref = _oracle_eps(cfg, sd_cn, sd_un, x, t, ctx, hint)
        e = rel_l2(eps, ref)
        # the same call through the context-K/V cache (first call fills it, the second one reads it)
        eng.cache_context_kv = True
        eng.reset_context_cache()
        try:
            model.apply_model(x, t, cond)
            eps_kv = model.apply_model(x, t, cond)
        finally:
            eng.cache_context_kv = False
            eng.reset_context_cache()
        e_kv = rel_l2(eps_kv, ref)
        cmp_ = rel_l2(_oracle_eps(cfg, sd_cn, sd_un, x, t, ctx, hint, autocast=True), ref)     # measured at both batch sizes (round 6)
        _record("inference_eps_vs_oracle", B=B, eps=e, eps_kv_cached=e_kv, comparator_bf16_autocast=cmp_)
        assert e < BF16_EPS and e_kv < BF16_EPS, (B, e, e_kv)
        assert rel_l2(eps_kv, eps) < 1e-6            # the cache holds exactly what the uncached call computes
        if cmp_ is not None:
            assert e < 1.3 * cmp_ + 1e-3, (e, cmp_)
        if B == 16:
            # round 5: LayerNorm as the prologue of the product that reads it (csrc/gemm_xs.hip): 42 of the 69 launches gone
            # (the C = 320 / 640 levels of both networks: 14 transformer blocks x 3 norms) and the same ACCURACY as with every
            # LayerNorm as its own launch.  The two eps tensors themselves differ at the bf16 noise level (measured 7.9e-3 for
            # 8.6e-3 against the oracle): the statistics are summed in another order, a few normalised values round the other way,
            # and ~100 bf16-rounded layers amplify any perturbation to their own rounding noise -- so the gate is each form
            # against the ORACLE, and the two errors against each other
            from ctrlora_amd import hip
            calls = {"n": 0}
            orig = hip.layernorm_fwd

            def counting(*a, **k):
                calls["n"] += 1
                return orig(*a, **k)
            hip.layernorm_fwd = counting
            try:
                assert hip.LN_PROLOGUE and hip.XS_ENABLED
                calls["n"] = 0
                model.apply_model(x, t, cond)
                n_on = calls["n"]
                hip.LN_PROLOGUE = False
                calls["n"] = 0
                eps_off = model.apply_model(x, t, cond)
                n_off = calls["n"]
            finally:
                hip.LN_PROLOGUE = True
                hip.layernorm_fwd = orig
            d, e_off = rel_l2(eps, eps_off), rel_l2(eps_off, ref)
            _record("inference_ln_prologue_vs_separate_layernorm", eps_diff=d, ln_launches_on=n_on, ln_launches_off=n_off,
                    eps_on_vs_oracle=e, eps_off_vs_oracle=e_off)
            assert n_off == 69 and n_on == 27, (n_on, n_off)
            assert d < BF16_EPS and e_off < BF16_EPS and abs(e - e_off) < 0.15 * e_off, (d, e, e_off)


def test_lora_fold_is_as_accurate_as_the_two_segment_executor_small_update():
    """W + B A folded into ONE bf16 weight (single rounding of the fp32 sum) vs the K-segment product [x | xA^T].[W | B]^T,
    same weights, up-projection 1e-3 N(0,1) (update below the bf16 quantum of W): eps of both vs the fp32 oracle, and the
    LoRA's own contribution d = eps(B) - eps(B = 0) in fp32 for scale."""
    _need_gpu()
    from ctrlora_amd.engine import ControlNetE, CtrLoRAEngine
    from oracle import arch
    cfg = arch.SD15
    model, sd_cn, sd_un = _inference_model()
    gen = torch.Generator().manual_seed(5)
    small = dict(sd_cn)
    zero = dict(sd_cn)
    for k, v in sd_cn.items():
        if k.endswith("lora_layer.up.weight"):
            small[k] = (torch.randn(v.shape, generator=gen) * 1e-3).to(v.device)
            zero[k] = torch.zeros_like(v)
    unet = model.engine().unet
    B = 4
    x, hint = torch.randn(B, 4, 64, 64, generator=gen).cuda(), (torch.randn(B, 4, 64, 64, generator=gen) * 0.9).cuda()
    ctx = torch.randn(B, 77, cfg.context_dim, generator=gen).cuda()
    t = torch.randint(0, 1000, (B,), generator=gen).cuda()
    ref = _oracle_eps(cfg, small, sd_un, x, t, ctx, hint)
    lora_part = rel_l2(ref, _oracle_eps(cfg, zero, sd_un, x, t, ctx, hint))     # how much of eps the LoRA is responsible for
    errs = {}
    for name, merge in (("folded", True), ("two_segment", False)):
        cn = ControlNetE(small, _netcfg(cfg), torch.bfloat16, torch.device("cuda"), need_bwd=False, merge_lora=merge)
        assert cn.merge_lora == merge
        eng = CtrLoRAEngine.from_executors(unet, [cn])
        errs[name] = rel_l2(eng.forward(x, t, ctx, [hint]), ref)
        del eng, cn
    _record("lora_fold_small_update", lora_share_of_eps=lora_part, **errs)
    assert errs["folded"] < BF16_EPS and errs["two_segment"] < BF16_EPS
    assert errs["folded"] < 1.15 * errs["two_segment"] + 5e-4, errs


def test_ddim_cfg_sampler_sd15_latent64_vs_oracle_sampler():
    """configs[4] through DDIMSampler.sample as bench.py calls it (hipGraph replay, K/V cache, CFG 7.5 batched as 2B):
    S = 4 steps from the same x_T vs the oracle's sampler in fp32 on the GPU; gate = 1.5 x the bf16-autocast oracle's own
    deviation from fp32, measured here."""
    _need_gpu()
    from cldm.ddim_hacked import DDIMSampler
    from oracle import arch, ref_model as R
    cfg = arch.SD15
    model, sd_cn, sd_un = _inference_model()
    B, H, S = 4, 64, 4
    g = torch.Generator().manual_seed(7)
    hint = (torch.randn(B, 4, H, H, generator=g) * 0.9).cuda()
    ctx, ctx_u = torch.randn(B, 77, cfg.context_dim, generator=g).cuda(), torch.randn(B, 77, cfg.context_dim, generator=g).cuda()
    x_T = torch.randn(B, 4, H, H, generator=g).cuda()
    cond = {"c_concat": [hint], "c_crossattn": [ctx]}
    unc = {"c_concat": [hint], "c_crossattn": [ctx_u]}
    s = DDIMSampler(model)
    x, _ = s.sample(S, B, (4, H, H), cond, verbose=False, eta=0.0, x_T=x_T, unconditional_guidance_scale=7.5,
                    unconditional_conditioning=unc)
    assert list(s.ddim_timesteps) == [1, 251, 501, 751]
    sched = R.make_schedule()          # tables stay on the host (the oracle's sampler indexes them with numpy); x lives on the GPU

    def sampler(autocast):
        fn = lambda xx, tt, c: _oracle_eps(cfg, sd_cn, sd_un, xx, tt, ctx if c else ctx_u, hint, autocast=autocast)
        out, _ = R.ddim_sample(fn, sched, S, x_T, scale=7.5, uncond=True)
        return out.float()

    ref = sampler(None)
    e = rel_l2(x, ref)
    cmp_ = rel_l2(sampler(True), ref)
    _record("ddim_s4_cfg_vs_oracle", x_engine_bf16=e, x_comparator_bf16_autocast=cmp_)
    assert torch.isfinite(x).all()
    assert e < 1.5 * cmp_ + 2e-3, (e, cmp_)
    assert e < 8e-2                                   # absolute backstop: guidance 7.5 amplifies eps_c - eps_u


def test_ddim50_cfg_sampler_config5_trajectory_vs_oracle_sampler():
    """BASELINE configs[4] at ITS OWN LENGTH (VERDICT r5 'weak' #1): DDIMSampler.sample with S = 50, CFG 7.5 batched as 2B,
    hipGraph replay with the device cursor walking all 50 indices, K/V cache, folded LoRA, latent 64x64 -- against the oracle's
    sampler in fp32 on the GPU, step by step (log_every_t = 1).  The gate is the same-precision comparator's own 50-step
    deviation (the oracle under bf16 autocast), measured in this run; the errors after 1, 4, 10, 25, 50 steps are recorded.
    Integer bookkeeping: the 50 timesteps equal the reference's table (tests/golden/schedule.pt, S50_eta0.0) exactly.
    Reference: cldm/ddim_hacked.py:123-231."""
    _need_gpu()
    import os
    import numpy as np
    from cldm.ddim_hacked import DDIMSampler
    from oracle import arch, ref_model as R
    cfg = arch.SD15
    model, sd_cn, sd_un = _inference_model()
    B, H, S = 2, 64, 50
    g = torch.Generator().manual_seed(50)
    hint = (torch.randn(B, 4, H, H, generator=g) * 0.9).cuda()
    ctx, ctx_u = torch.randn(B, 77, cfg.context_dim, generator=g).cuda(), torch.randn(B, 77, cfg.context_dim, generator=g).cuda()
    x_T = torch.randn(B, 4, H, H, generator=g).cuda()
    cond = {"c_concat": [hint], "c_crossattn": [ctx]}
    unc = {"c_concat": [hint], "c_crossattn": [ctx_u]}
    s = DDIMSampler(model)
    x, inter = s.sample(S, B, (4, H, H), cond, verbose=False, eta=0.0, x_T=x_T, unconditional_guidance_scale=7.5,
                        unconditional_conditioning=unc, log_every_t=1)
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "schedule.pt"), weights_only=False)["ddim"]["S50_eta0.0"]
    assert np.array_equal(np.asarray(s.ddim_timesteps), np.asarray(gold["timesteps"]))        # bit-exact index bookkeeping
    traj = inter["x_inter"]
    assert len(traj) == S + 1 and torch.equal(traj[-1], x)
    sched = R.make_schedule()

    def sampler(autocast):
        fn = lambda xx, tt, c: _oracle_eps(cfg, sd_cn, sd_un, xx, tt, ctx if c else ctx_u, hint, autocast=autocast)
        keep = []
        out, _ = R.ddim_sample(fn, sched, S, x_T, scale=7.5, uncond=True, keep=keep)
        return [k.float() for k in keep]

    ref = sampler(None)
    cmp_ = sampler(True)
    marks = (1, 4, 10, 25, 50)
    e = {k: rel_l2(traj[k], ref[k - 1]) for k in marks}
    c = {k: rel_l2(cmp_[k - 1], ref[k - 1]) for k in marks}
    _record("ddim_s50_cfg_vs_oracle", B=B, engine_bf16={str(k): v for k, v in e.items()},
            comparator_bf16_autocast={str(k): v for k, v in c.items()})
    assert torch.isfinite(x).all()
    for k in marks:
        assert e[k] < 1.5 * c[k] + 2e-3, (k, e, c)
    # the same graph replayed again from the same x_T gives the same sample bit for bit (device cursor rewound, caches reused)
    x2, _ = s.sample(S, B, (4, H, H), cond, verbose=False, eta=0.0, x_T=x_T, unconditional_guidance_scale=7.5,
                     unconditional_conditioning=unc)
    assert torch.equal(x2, x)


# ------------------------------------------------------------------------------ conv-tap weight gradient, production shapes

@pytest.mark.parametrize("B,H,Cin,Cout,stride", [(8, 64, 320, 320, 1), (8, 64, 320, 320, 2), (8, 64, 4, 320, 1),
                                                 (8, 16, 1280, 1280, 1)])
def test_conv_tap_weight_gradient_production_shapes(B, H, Cin, Cout, stride):
    """dW[o][tap][i] of a 3x3 conv formed as nine grouped problems whose x operand is gathered inside the kernel's
    addressing (cl_wgrad_desc.tap): ResBlock conv 320 -> 320 at 64x64, the stride-2 Downsample, the 4-channel input conv
    (I padded to 32) and a deep 1280-channel level, vs torch's conv weight gradient in fp64 on the same bf16 values."""
    _need_gpu()
    from ctrlora_amd import hip
    from ctrlora_amd.engine.packing import rup
    g = torch.Generator().manual_seed(B * H + Cin + stride)
    Ip = rup(Cin, 32)
    Ho = H // stride
    x = torch.zeros(B * H * H, Ip)
    x[:, :Cin] = torch.randn(B * H * H, Cin, generator=g)
    dy = torch.randn(B * Ho * Ho, Cout, generator=g) * 0.1
    xb, dyb = _bf(x).cuda(), _bf(dy).cuda()
    dW = torch.zeros(Cout, 9 * Ip, dtype=torch.float32, device="cuda")
    probs = [(dyb, xb, dW[:, tp * Ip:(tp + 1) * Ip], 1.0, (tp, H, H, Ho, Ho, stride, 1)) for tp in range(9)]
    hip.weight_grad_tn_group(probs)
    torch.cuda.synchronize()
    x64 = xb.double().view(B, H, H, Ip)[..., :Cin].permute(0, 3, 1, 2).contiguous()
    dy64 = dyb.double().view(B, Ho, Ho, Cout).permute(0, 3, 1, 2).contiguous()
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    torch.nn.functional.conv2d(x64, w, stride=stride, padding=1).backward(dy64)
    ref = torch.zeros(Cout, 3, 3, Ip, dtype=torch.float64, device="cuda")
    ref[..., :Cin] = w.grad.permute(0, 2, 3, 1)
    e = rel_l2(dW, ref.view(Cout, 9 * Ip))
    per_tap = max(rel_l2(dW[:, tp * Ip:tp * Ip + Cin], ref.view(Cout, 9, Ip)[:, tp, :Cin]) for tp in range(9))
    _record("conv_tap_wgrad", shape=[B, H, Cin, Cout, stride], rel=e, worst_tap=per_tap)
    assert e < 1e-5 and per_tap < 1e-5, (e, per_tap)          # bf16 products are exact in fp32; only the summation order differs
    if Ip > Cin:
        assert float(dW.view(Cout, 9, Ip)[:, :, Cin:].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(8, 64, 64, 320, 320), (8, 32, 32, 640, 640), (8, 16, 16, 1280, 1280), (8, 8, 8, 1280, 1280),
                                            (2, 16, 8, 96, 160), (1, 8, 32, 64, 352), (3, 32, 64, 32, 64), (2, 64, 64, 4, 320)])
def test_conv_weight_gradient_three_taps_per_problem_vs_fp64_and_the_single_tap_form(B, H, W, Cin, Cout):
    """cl_wgrad_desc.tap = 16 + ky: the three taps of a kernel row of a stride-1 conv from ONE problem (wgrad_row3_kernel: dy tile
    loaded once per step, one x tile with a halo pixel per image-row segment) vs torch's conv weight gradient in fp64 on the
    same bf16 values, and vs the nine single-tap problems on the same call (same products, another summation order).  Grids:
    the four levels of the batch-8 step (W = 64 / 32: a step inside one image row; W = 16 / 8: a step spans 2 / 4 rows, across
    sample boundaries at the 8x8 level), non-square grids, ragged channel counts, split and unsplit m."""
    _need_gpu()
    from ctrlora_amd import hip
    from ctrlora_amd.engine.packing import rup
    g = torch.Generator().manual_seed(B * H + W + Cin)
    Ip = rup(Cin, 32)
    x = torch.zeros(B * H * W, Ip)
    x[:, :Cin] = torch.randn(B * H * W, Cin, generator=g)
    dy = torch.randn(B * H * W, Cout, generator=g) * 0.1
    xb, dyb = _bf(x).cuda(), _bf(dy).cuda()
    dW3 = torch.full((Cout, 9 * Ip), 0.25, dtype=torch.float32, device="cuda")       # accumulates INTO the gradient buffer
    dW9 = torch.full((Cout, 9 * Ip), 0.25, dtype=torch.float32, device="cuda")
    hip.weight_grad_tn_group([(dyb, xb, dW3[:, 3 * ky * Ip:(3 * ky + 1) * Ip], 0.5, (16 + ky, H, W, H, W, 1, 1)) for ky in range(3)])
    hip.weight_grad_tn_group([(dyb, xb, dW9[:, tp * Ip:(tp + 1) * Ip], 0.5, (tp, H, W, H, W, 1, 1)) for tp in range(9)])
    dW3b = torch.full((Cout, 9 * Ip), 0.25, dtype=torch.float32, device="cuda")
    hip.weight_grad_tn_group([(dyb, xb, dW3b[:, 3 * ky * Ip:(3 * ky + 1) * Ip], 0.5, (16 + ky, H, W, H, W, 1, 1)) for ky in range(3)])
    torch.cuda.synchronize()
    assert torch.equal(dW3, dW3b)                                                     # deterministic (slabs + fixed-order reduce)
    x64 = xb.double().view(B, H, W, Ip)[..., :Cin].permute(0, 3, 1, 2).contiguous()
    dy64 = dyb.double().view(B, H, W, Cout).permute(0, 3, 1, 2).contiguous()
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    torch.nn.functional.conv2d(x64, w, padding=1).backward(dy64)
    ref = torch.zeros(Cout, 3, 3, Ip, dtype=torch.float64, device="cuda")
    ref[..., :Cin] = w.grad.permute(0, 2, 3, 1)
    ref = 0.25 + 0.5 * ref.view(Cout, 9 * Ip)
    e3, e9 = rel_l2(dW3 - 0.25, ref - 0.25), rel_l2(dW9 - 0.25, ref - 0.25)
    per_tap = max(rel_l2(dW3[:, tp * Ip:tp * Ip + Cin] - 0.25, ref[:, tp * Ip:tp * Ip + Cin] - 0.25) for tp in range(9))
    _record("conv_row3_wgrad", shape=[B, H, W, Cin, Cout], rel=e3, single_tap_rel=e9, worst_tap=per_tap)
    assert e3 < 1e-5 and per_tap < 1e-5 and e9 < 1e-5, (e3, e9, per_tap)
    if Ip > Cin:
        assert float((dW3.view(Cout, 9, Ip)[:, :, Cin:] - 0.25).abs().max()) == 0.0
    # descriptors the kernel does not cover are refused, not guessed at
    bad = torch.zeros(Cout, 9 * Ip, dtype=torch.float32, device="cuda")
    with pytest.raises(hip.HipError):
        hip.weight_grad_tn_group([(dyb, xb, bad[:, :Ip], 1.0, (16, H, W, H // 2, W // 2, 2, 1))])   # stride 2
    with pytest.raises(hip.HipError):
        hip.weight_grad_tn_group([(dyb, xb, bad[:, :Ip], 1.0, (19, H, W, H, W, 1, 1))])             # no such kernel row


# ------------------------------------------------------------------------------ pre-training at SD1.5 width

@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pretraining_step_sd15_width_all_gradients_vs_oracle(dtype):
    """BASELINE configs[3] at full width (ctrlora_pretrain_sd15_9tasks_rank128.yaml, B = 2, latent 64x64): loss and the
    gradient of EVERY control_model parameter that takes part (360 M base weights incl. the nine-tap conv weights + the
    task's LoRA bank) vs oracle.p_losses + autograd in fp32 on the GPU."""
    _need_gpu()
    import bench
    from oracle import arch, ref_model as R
    cfg = arch.SD15
    task = "canny"
    # fast_init=False: the absolute bf16 gates below were calibrated (round 3) on torch's own seeded default initialisation; the
    # block-window draw of ctrlora_amd/fastinit.py is another sample of the same distributions on which the same kernels measure
    # 5.2e-2 / 2.2e-2 instead of 4.1e-2 / 1.6e-2 (fp32: 1e-5 on both) -- profiles/r05_final/pretrain_gate_vs_weight_draw.txt
    m = bench.build_model("ctrlora_pretrain_sd15_9tasks_rank128.yaml", 0, fast_init=False).cuda().train()
    m.set_engine_dtype(dtype)
    m.learning_rate = 1e-5
    cm = m.control_model
    cm.switch_lora(task)
    sd_cn = {k: v.detach().clone() for k, v in cm._executor_state().items()}
    sd_un = {k: v.detach().clone() for k, v in m.model.diffusion_model.state_dict().items()}
    opt = m.configure_optimizers()
    d = bench.synth(2, 64, cfg.context_dim, "cuda", 31, 1)
    z, ctx, hint, t, noise = d["z"][0], d["ctx"][0], d["hint"][0], d["t"][0], d["noise"][0]
    opt.zero_grad()
    loss, _ = m.p_losses(z, {"c_crossattn": [ctx], "c_concat": [hint], "task": task}, t, noise=noise)
    loss.backward()
    torch.cuda.synchronize()
    ours = {}
    for n, p in cm.named_parameters():
        if n.startswith("loras_dict."):
            continue
        ours[n] = p.grad
    assert opt.active == [task]
    # the oracle: every tensor of the ControlNet requires grad
    dev = torch.device("cuda")
    cn = {k: v.to(dev).clone().requires_grad_(True) for k, v in sd_cn.items()}
    un = {k: v.to(dev) for k, v in sd_un.items()}
    sched = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in R.make_schedule().items()}
    loss_o, _ = R.p_losses(cn, un, cfg, sched, z, t, ctx, hint, noise)
    loss_o.backward()
    errs = []
    missing = [k for k in cn if k not in ours]
    assert not missing, missing[:5]
    for k, v in cn.items():
        if v.grad is None:
            continue
        a, b = ours[k].detach().double().reshape(-1), v.grad.detach().double().reshape(-1)     # on the GPU: 400 M elements
        errs.append((float((a - b).norm() / (b.norm() + 1e-30)), k, v.grad.numel()))
    errs.sort(reverse=True)
    n_el = sum(e[2] for e in errs)
    conv_errs = [e for e in errs if e[1].endswith(".weight") and cn[e[1]].dim() == 4 and cn[e[1]].shape[-1] == 3]
    med = errs[len(errs) // 2][0]
    _record("pretrain_sd15_step", dtype=str(dtype), loss=float(loss), loss_oracle=float(loss_o), tensors=len(errs),
            elements_M=round(n_el / 1e6, 1), grad_max=errs[0][0], grad_max_name=errs[0][1], grad_median=med,
            conv3x3_weight_max=conv_errs[0][0], conv3x3_weight_name=conv_errs[0][1])
    assert n_el > 380e6 and len(conv_errs) >= 20
    if dtype == torch.float32:
        assert abs(float(loss) - float(loss_o)) < 1e-4 * float(loss_o)
        assert errs[0][0] < 5e-4, errs[:5]
    else:
        assert abs(float(loss) - float(loss_o)) < 2e-2 * float(loss_o)
        assert errs[0][0] < 5.5e-2 and med < 2.1e-2, (errs[:5], med)      # measured 4.2e-2 / 1.6e-2 (profiles/r03_parity_measured_b.jsonl)


def test_pretraining_bf16_error_is_flat_on_the_fp32_parameter_trajectory():
    """tests/test_pretrain.py sees the worst bf16 gradient error grow 6e-2 -> 1.3e-1 -> 1.8e-1 over three optimizer steps
    (tiny width, lr 1e-3).  Here the bf16 model is given the fp32 model's parameters before every step: if the kernels
    were at fault the error would still grow; it stays at the step-0 level, i.e. the growth is the two parameter
    trajectories drifting apart (AdamW's first steps are lr * sign(g): a bf16-noisy sign moves a weight 2 * lr the other
    way, and lr = 1e-3 is 5 % of the 0.02-scale weights)."""
    _need_gpu()
    from tests.golden.make_golden_pretrain import LR, SEQ, step_inputs
    from tests.test_pretrain import _model
    mf, cfg = _model(torch.float32)
    mb, _ = _model(torch.bfloat16)
    mf.learning_rate = mb.learning_rate = LR
    of, ob = mf.configure_optimizers(), mb.configure_optimizers()
    cu = lambda v: v.cuda()
    worst = []
    for i, task in enumerate(SEQ):
        with torch.no_grad():      # bf16 model <- fp32 model's parameters (writes through to the flat masters), then re-pack
            src = dict(mf.control_model.named_parameters())
            for n, p in mb.control_model.named_parameters():
                p.copy_(src[n])
        mb.control_model._on_state_loaded()
        grads = []
        for m, opt in ((mf, of), (mb, ob)):
            inp = step_inputs(cfg, i)
            cond = dict(c_crossattn=[cu(inp["ctx"])], c_concat=[cu(inp["hint_z"])], task=task)
            opt.zero_grad()
            loss, _ = m.p_losses(cu(inp["z"]), cond, cu(inp["t"]), noise=cu(inp["noise"]))
            loss.backward()
            torch.cuda.synchronize()
            grads.append({n: p.grad.detach().clone() for n, p in m.control_model.named_parameters()
                          if p.grad is not None and float(p.grad.abs().max()) > 0})
        gf, gb = grads
        assert set(gf) == set(gb)
        e = max((rel_l2(gb[n], gf[n]), n) for n in gf)
        worst.append(e)
        of.step()
    _record("pretrain_bf16_on_fp32_trajectory", worst_per_step=[w[0] for w in worst], names=[w[1] for w in worst])
    assert max(w[0] for w in worst) < 7.5e-2, worst          # measured 5.4e-2 / 4.6e-2 / 5.6e-2
    assert worst[-1][0] < 1.5 * worst[0][0] + 1e-2, worst          # no growth along the trajectory


# ------------------------------------------------------------------------------ grouped LoRA products (q | k | v in one launch)

@pytest.mark.parametrize("M,K,N,r,G", [(32768, 320, 320, 128, 3), (8192, 640, 640, 128, 3), (2048, 1280, 1280, 128, 3),
                                       (616, 768, 640, 128, 2), (616, 768, 320, 128, 2), (512, 1280, 1280, 128, 3),
                                       (4096, 320, 320, 32, 3), (1000, 64, 64, 32, 2)])
def test_grouped_lora_products_vs_fp64(M, K, N, r, G):
    """cl_gemm with grouped K segments (csrc/gemm.h a2_group_n / a1_group_n): G LoRACompatibleLinears that share their input
    (cldm/lora.py:285-291; to_q | to_k | to_v) as one forward launch, one u = [dy_g B_g] launch and one dx launch, vs fp64
    on the same bf16 values, at the production shapes of the three attention levels, the text-context projections
    (M = 8 x 77), rank 32 (second segment below a 128-byte line) and a tiny ragged case."""
    _need_gpu()
    from ctrlora_amd import hip
    g = torch.Generator().manual_seed(M + N + r + G)
    bf = lambda *s, sc=1.0: _bf(torch.randn(*s, generator=g) * sc).cuda()
    x = bf(M, K)
    W, A, Bm = bf(G * N, K, sc=K ** -0.5), bf(G * r, K, sc=K ** -0.5), bf(G * N, r, sc=0.05)
    dy = bf(M, G * N, sc=0.1)
    t = torch.empty(M, G * r, dtype=torch.bfloat16, device="cuda")
    y = torch.empty(M, G * N, dtype=torch.bfloat16, device="cuda")
    hip.gemm(x, A, t)
    hip.gemm(x, W, y, a2=t, w2=Bm, a2_group_n=N)
    x64, W64, A64, B64, dy64 = x.double(), W.double(), A.double(), Bm.double(), dy.double()
    t64 = t.double()                                              # the kernel's own bf16 t (what the second segment reads)
    y_ref = torch.cat([x64 @ W64[i * N:(i + 1) * N].T + t64[:, i * r:(i + 1) * r] @ B64[i * N:(i + 1) * N].T for i in range(G)], 1)
    e_t = rel_l2(t, x64 @ A64.T)
    e_y = rel_l2(y, y_ref)
    # backward: u_g = dy_g B_g ; dx = sum_g dy_g W_g + u_g A_g
    Bt = torch.cat([Bm[i * N:(i + 1) * N].t().contiguous() for i in range(G)], 0)        # [G r, N]
    Wt = W.view(G, N, K).permute(2, 0, 1).reshape(K, G * N).contiguous()                  # [K, G N]
    At = A.t().contiguous()                                                              # [K, G r]
    u = torch.empty(M, G * r, dtype=torch.bfloat16, device="cuda")
    e_u = None
    if r % 64 == 0:
        hip.gemm(dy, Bt, u, k1=N, a1_group_n=r)
        u_ref = torch.cat([dy64[:, i * N:(i + 1) * N] @ B64[i * N:(i + 1) * N] for i in range(G)], 1)
        e_u = rel_l2(u, u_ref)
    else:
        with pytest.raises(hip.HipError):                          # no tile narrower than the group: the engine falls back
            hip.gemm(dy, Bt, u, k1=N, a1_group_n=r)
        for i in range(G):
            hip.gemm(dy[:, i * N:(i + 1) * N], Bt[i * r:(i + 1) * r], u[:, i * r:(i + 1) * r])
    dx = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    hip.gemm(dy, Wt, dx, a2=u, w2=At)
    dx_ref = dy64 @ W64 + u.double() @ A64
    e_dx = rel_l2(dx, dx_ref)
    _record("grouped_lora", shape=[M, K, N, r, G], t=e_t, y=e_y, u=e_u, dx=e_dx)
    tol = 2.5e-3                                                   # one bf16 rounding of the result (2^-9 = 1.95e-3 worst, ~1.7e-3 rms)
    assert e_t < tol and e_y < tol and e_dx < tol and (e_u is None or e_u < tol), (e_t, e_y, e_u, e_dx)


# ------------------------------------------------------------------------------ one-launch GroupNorm (register-resident slab)

@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("C,HW,silu,train", [(640, 1024, True, False), (1280, 1024, False, True), (1920, 256, True, False),
                                             (960, 1024, True, False), (1280, 256, False, True), (2560, 64, True, False),
                                             (2560, 256, True, False), (640, 256, True, True), (1280, 60, True, True),
                                             (2560, 4, True, True), (1920, 4, True, True), (2560, 1, True, False)])
def test_groupnorm_one_launch_production_shapes(dtype, C, HW, silu, train):
    """GroupNorm32 (+SiLU) forward / backward at the 32x32, 16x16 and 8x8 levels of SD1.5 (B = 8), where the one-launch
    kernels (csrc/norm.hip gn1_*: the (sample, channel-block) slab stays in registers between statistics and apply) take
    over from the two-launch form: vs torch in fp64, and the two forms against each other (cl_debug_groupnorm_form); a ragged
    pixel count too, and 128-px images (2x2 / 1x1 levels: fewer pixels than channels in the block -- ADVICE r3: the
    workgroup needs one thread per channel).  ldm/modules/diffusionmodules/util.py:217-219, openaimodel.py:201-202, attention.py:88-89."""
    _need_gpu()
    from ctrlora_amd import hip
    B, eps = 8, (1e-5 if silu else 1e-6)
    g = torch.Generator().manual_seed(C + HW)
    x = (torch.randn(B * HW, C, generator=g) * 1.5 + 0.3).to(dtype)
    dy = torch.randn(B * HW, C, generator=g).to(dtype)
    acc = torch.randn(B * HW, C, generator=g).to(dtype)
    gamma = 1 + 0.2 * torch.randn(C, generator=g); beta = 0.2 * torch.randn(C, generator=g)
    dev = torch.device("cuda")
    xr = x.to(dev).double().reshape(B, HW, C).permute(0, 2, 1).requires_grad_(True)
    gr, br = gamma.to(dev).double().requires_grad_(True), beta.to(dev).double().requires_grad_(True)
    yr = torch.nn.functional.group_norm(xr, 32, gr, br, eps)
    if silu:
        yr = torch.nn.functional.silu(yr)
    yr.backward(dy.to(dev).double().reshape(B, HW, C).permute(0, 2, 1))
    y_ref = yr.detach().permute(0, 2, 1).reshape(B * HW, C)
    dx_ref = xr.grad.permute(0, 2, 1).reshape(B * HW, C) + acc.to(dev).double()
    xd, dyd, accd, gd, bd = x.to(dev), dy.to(dev), acc.to(dev), gamma.to(dev), beta.to(dev)
    out = {}
    for form in (35, 34):                    # one-launch on / off
        hip.lib().cl_debug_groupnorm_form(0, int(form == 35))
        try:
            y = torch.empty_like(xd); dx = torch.empty_like(xd)
            stats = torch.empty(B, 32, 2, device=dev); ws = torch.zeros(hip.groupnorm_ws(B, HW, C), device=dev)
            dgam, dbet = (torch.zeros(C, device=dev), torch.zeros(C, device=dev)) if train else (None, None)
            hip.groupnorm_fwd(xd, y, gd, bd, B, HW, eps, silu, stats, ws)
            hip.groupnorm_bwd(xd, dyd, dx, gd, bd, stats, B, HW, silu, ws, accum=accd, dgamma=dgam, dbeta=dbet)
            torch.cuda.synchronize()
            out[form] = (y, dx, stats, dgam, dbet)
        finally:
            hip.lib().cl_debug_groupnorm_form(0, 1)
    y, dx, stats, dgam, dbet = out[35]
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    e_y, e_dx = rel_l2(y, y_ref), rel_l2(dx, dx_ref)
    mean_ref = xr.detach().reshape(B, 32, -1).mean(-1)
    e_mu = float((stats[:, :, 0].double() - mean_ref).abs().max())
    _record("groupnorm_one_launch", dtype=str(dtype), shape=[C, HW, silu, train], y=e_y, dx=e_dx, mean_abs=e_mu,
            y_vs_two_launch=rel_l2(y, out[34][0]), dx_vs_two_launch=rel_l2(dx, out[34][1]))
    assert e_y < tol and e_dx < 2 * tol and e_mu < 1e-4, (e_y, e_dx, e_mu)
    assert rel_l2(stats, out[34][2]) < 1e-5
    assert rel_l2(y, out[34][0]) < tol and rel_l2(dx, out[34][1]) < 2 * tol
    if train:
        assert rel_l2(dgam, gr.grad) < (2e-5 if dtype == torch.float32 else 1e-2)
        assert rel_l2(dbet, br.grad) < (2e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("M,D", [(32768, 320), (8192, 640), (2048, 1280), (1030, 320), (1027, 640)])
def test_layernorm_forward_production_shapes(M, D):
    """nn.LayerNorm forward (attention.py:263-265) at the token counts of the three attention levels (B = 8) and ragged row
    counts vs torch fp64, statistics included."""
    _need_gpu()
    from ctrlora_amd import hip
    g = torch.Generator().manual_seed(M + D)
    x = _bf(torch.randn(M, D, generator=g) * 2 + 0.5).cuda()
    gamma = (1 + 0.2 * torch.randn(D, generator=g)).cuda(); beta = (0.2 * torch.randn(D, generator=g)).cuda()
    ref = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-5)
    y = torch.empty_like(x); stats = torch.empty(M, 2, device="cuda")
    hip.layernorm_fwd(x, y, gamma, beta, 1e-5, stats)
    torch.cuda.synchronize()
    e = rel_l2(y, ref)
    mu = x.double().mean(1)
    _record("layernorm_fwd", shape=[M, D], y=e, mean_abs=float((stats[:, 0].double() - mu).abs().max()))
    assert e < 6e-3 and float((stats[:, 0].double() - mu).abs().max()) < 1e-5
