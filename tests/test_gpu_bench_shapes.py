"""GPU parity at the BENCHMARKED shapes (BASELINE.json configs[1] / configs[4]): SD1.5 width, latent 64x64.

The whole-model tests in test_gpu_parity.py run at latent <= 24x16; the paths that produce the headline
number -- 256-row ping-pong tiles over M = 32768 with XCD remap, 128-row mid-size tiles, attention at
N = 4096 / 1024, split-K at the 8x8 level, the two-stream forward, GraphedTrainStep, the graphed DDIM loop --
are checked here:

  * vs tests/golden/model_sd15_64.pt, produced by the UNMODIFIED reference (tests/golden/make_golden.py
    --only-sd15-64: ControlFinetuneLDM.p_losses + backward at B = 2, latent 64x64, rank 128);
  * vs the oracle (oracle/ref_model.py) evaluated on the GPU in fp32 through PyTorch-ROCm's stock kernels
    (checker only) where the CPU oracle would take minutes (B = 8);
  * per-kernel, vs fp64 torch, at the production shapes of the probes that used to live outside pytest.

Tolerances (rel-L2): fp32 parity mode 1e-4 eps / 5e-4 gradients; bf16 mode: see BF16_* below -- set from the
same-precision comparator (profiles/r02_compare_precision.json: the reference restatement under
torch.autocast(bfloat16) on the same GPU has the same order of error against fp32).
"""
import json
import os

import pytest
import torch

from tests.util import GOLDEN, ROOT, rel_l2

pytestmark = pytest.mark.gpu

# measured (profiles/r02_parity_measured.jsonl, r03): eps 8.9e-3 .. 9.6e-3, worst gradient 3.2e-2 .. 3.7e-2, median 1.28e-2;
# the reference's own bf16-autocast path: 1.1e-2 / 3.2e-2 / 1.2e-2.  Gates = ~1.3 x measured, so a real regression trips them.
BF16_EPS, BF16_GRAD_MAX, BF16_GRAD_MEDIAN = 1.3e-2, 5e-2, 1.7e-2
# Round 6: the static numbers above stay as BACKSTOPS; the gate proper of every whole-model bf16 check is
# k x the same-precision comparator -- the oracle under torch.autocast(bfloat16) through PyTorch-ROCm's own kernels, i.e. what the
# reference's bf16 training does -- measured IN THE SAME TEST on the same inputs: k = 1.3 for eps and the median gradient, 1.5 for
# the worst single gradient (a maximum over 246 tensors: the noisier statistic).
K_CMP, K_CMP_MAX = 1.3, 1.5


def _comparator_vs(ref_eps, ref_grad_of, cfg, sd_cn, sd_un, z, t, ctx, hint, noise):
    """(eps, worst gradient, median gradient) error of the bf16-autocast oracle against a reference: ref_eps a tensor,
    ref_grad_of(name, grad) -> rel-L2 of that gradient against the reference's."""
    _, eps_c, grads_c = _oracle_on_gpu(cfg, sd_cn, sd_un, z, t, ctx, hint, noise, autocast=torch.bfloat16)
    ge = sorted((ref_grad_of(n, g) for n, g in grads_c.items()), reverse=True)
    return rel_l2(eps_c, ref_eps), ge[0], ge[len(ge) // 2]


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ctrlora_amd import hip
    hip.lib()


def _record(name, **vals):
    """Measured errors go to gpurun_out/parity_measured.jsonl so that gates can be set from evidence."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=name, **vals)) + "\n")
    except OSError:
        pass


def _netcfg(c):
    from ctrlora_amd.engine import NetCfg
    return NetCfg(c.in_channels, c.out_channels, c.model_channels, c.channel_mult, c.num_res_blocks,
                  c.attention_resolutions, c.num_heads, c.context_dim)


def _oracle_on_gpu(cfg, sd_cn, sd_un, z, t, ctx, hint, noise, autocast=None):
    """oracle.p_losses + autograd on the GPU (stock kernels, fp32 or bf16 autocast): the CHECKER."""
    from oracle import arch, ref_model as R
    dev = torch.device("cuda")
    cn = {k: v.detach().to(dev).clone() for k, v in sd_cn.items()}
    un = {k: v.detach().to(dev) for k, v in sd_un.items()}
    for k in cn:
        if arch.is_trainable(k):
            cn[k].requires_grad_(True)
    sched = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in R.make_schedule().items()}
    with torch.autocast("cuda", dtype=autocast or torch.bfloat16, enabled=autocast is not None):
        loss, eps = R.p_losses(cn, un, cfg, sched, z.to(dev), t.to(dev), ctx.to(dev), hint.to(dev), noise.to(dev))
    loss.backward()
    grads = {k: v.grad.detach().float() for k, v in cn.items() if v.grad is not None}
    return float(loss), eps.detach().float(), grads


# ------------------------------------------------------------------------------ whole model, 64x64, vs the reference

@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sd15_latent64_forward_backward_vs_reference_golden(dtype):
    _need_gpu()
    from ctrlora_amd.engine import CtrLoRAEngine
    from oracle import arch, ref_model as R
    from tests.golden.make_golden import inputs_for
    path = os.path.join(GOLDEN, "model_sd15_64.pt")
    gold = torch.load(path, weights_only=False)
    meta = gold["meta"]
    assert (meta["B"], meta["H"]) == (2, 64)
    cfg = arch.SD15
    inp = inputs_for(cfg, meta["B"], meta["H"], meta["seed"])
    sd_cn = arch.make_state(arch.controlnet_shapes(cfg), meta["seed"])
    sd_un = arch.make_state(arch.unet_shapes(cfg), meta["seed"])
    eng = CtrLoRAEngine(sd_un, [sd_cn], _netcfg(cfg), dtype=dtype, device="cuda")
    x_noisy = R.q_sample(R.make_schedule(), inp["z"], inp["t"], inp["noise"])
    assert torch.equal(x_noisy, gold["x_noisy"])
    cu = lambda v: v.cuda()
    eps = eng.forward(cu(x_noisy), cu(inp["t"]), cu(inp["ctx"]), [cu(inp["hint_z"])], record=True)
    e_eps = rel_l2(eps, gold["eps"])
    loss = float(((eps.cpu() - inp["noise"]) ** 2).mean())
    eng.zero_grad()
    eng.backward(2.0 * (eps - cu(inp["noise"])) / eps.numel())
    torch.cuda.synchronize()
    gs = gold["grad_sampled"]
    items = eng.controls[0].tr.items
    assert len(items) == 246 and set(t.name for t in items) == set(gs)
    errs, norm_errs = [], []
    for t in items:
        g = gs[t.name]
        got = t.grad.detach().float().flatten().cpu()
        assert list(t.grad.shape) == g["shape"]
        errs.append((rel_l2(got[g["idx"]], g["vals"]), t.name))
        norm_errs.append(abs(float(got.double().norm()) - g["l2"]) / (g["l2"] + 1e-30))
    errs.sort(reverse=True)
    med = errs[len(errs) // 2][0]
    _record("sd15_64_vs_reference", dtype=str(dtype), eps=e_eps, loss=loss, loss_ref=gold["loss"], grad_max=errs[0][0],
            grad_max_name=errs[0][1], grad_median=med, grad_norm_max=max(norm_errs))
    if dtype == torch.float32:
        assert e_eps < 1e-4
        assert abs(loss - gold["loss"]) < 1e-4 * gold["loss"]
        assert errs[0][0] < 5e-4, errs[:5]
        assert max(norm_errs) < 5e-4
    else:
        # the comparator on the same inputs, against the same reference tensors
        c_eps, c_max, c_med = _comparator_vs(
            gold["eps"], lambda n, g: rel_l2(g.flatten().cpu()[gs[n]["idx"]], gs[n]["vals"]), cfg, sd_cn, sd_un,
            inp["z"], inp["t"], inp["ctx"], inp["hint_z"], inp["noise"])
        _record("sd15_64_vs_reference_comparator", eps=c_eps, grad_max=c_max, grad_median=c_med)
        assert e_eps < K_CMP * c_eps and errs[0][0] < K_CMP_MAX * c_max and med < K_CMP * c_med, (e_eps, errs[0], med, c_eps, c_max, c_med)
        assert e_eps < BF16_EPS
        assert abs(loss - gold["loss"]) < 2e-2 * gold["loss"]
        assert errs[0][0] < BF16_GRAD_MAX, errs[:5]
        assert med < BF16_GRAD_MEDIAN
        assert max(norm_errs) < BF16_GRAD_MAX


# ------------------------------------------------------------------------------ the bench's own step: B = 8, graphed

def test_graphed_two_stream_train_step_b8_latent64_matches_eager_and_oracle():
    """BASELINE configs[1] exactly as bench.py runs it (ControlFinetuneLDM from the rank-128 YAML, B = 8, latent
    64x64, bf16, GraphedTrainStep = hipGraph replay of zero_grad + two-stream forward + backward + AdamW):
      (a) graph replay == eager single-stream bf16 launches, bit for bit (every reduction is deterministic);
      (b) eps / loss / all 246 gradients vs the fp32 engine and vs the oracle in fp32 on the GPU."""
    _need_gpu()
    import bench
    from ctrlora_amd.engine import CtrLoRAEngine
    from ctrlora_amd.train import GraphedTrainStep
    from oracle import arch
    cfg = arch.SD15
    B, H = 8, 64
    model = bench.build_model("ctrlora_finetune_sd15_rank128.yaml", 0).cuda().train()
    model.set_engine_dtype(torch.bfloat16)
    model.learning_rate = 0.0                       # AdamW with lr = 0 leaves the weights alone: gradients comparable
    sd_cn = {k: v.detach().cpu().clone() for k, v in model.control_model.state_dict().items()}
    sd_un = {k: v.detach().cpu().clone() for k, v in model.model.diffusion_model.state_dict().items()}
    opt = model.configure_optimizers()
    d = bench.synth(B, H, cfg.context_dim, "cuda", 99, 1)
    z, ctx, hint, t, noise = d["z"][0], d["ctx"][0], d["hint"][0], d["t"][0], d["noise"][0]
    g = GraphedTrainStep(model, opt, z, ctx, hint, t, noise, warmup=1)
    loss_g = float(g(z, ctx, hint, t, noise))
    loss_g2 = float(g(z, ctx, hint, t, noise))
    assert loss_g == loss_g2                          # replay is deterministic
    ex = model.control_model.executor()
    grads_g = ex.tr.flat_grad.clone()
    torch.cuda.synchronize()
    # (a) eager launches, same weights: with the same two-stream structure (same per-stream split-K scratch, hence the
    # same summation order) the result must be bit-identical; on ONE stream the split-K factors of the ControlNet
    # trunk differ (64 MiB default scratch instead of the side stream's 32 MiB), i.e. fp32 summation order changes
    # and bf16 roundings flip: bounded at bf16 noise level
    from ctrlora_amd import hip
    x_noisy = model.q_sample(z, t, noise)
    res = {}
    for overlap in (True, False):
        eng_b = CtrLoRAEngine(sd_un, [sd_cn], _netcfg(cfg), dtype=torch.bfloat16, device="cuda")
        eng_b.overlap_streams = overlap
        eps_b = eng_b.forward(x_noisy, t, ctx, [hint], record=True)
        eng_b.zero_grad()
        d_eps, loss_t = torch.empty_like(eps_b), torch.zeros((), device="cuda")
        hip.mse_loss(eps_b.contiguous(), noise.contiguous(), d_eps, loss_t)
        eng_b.backward(d_eps)
        torch.cuda.synchronize()
        tr_b = eng_b.controls[0].tr
        assert [t_.name for t_ in tr_b.items] == [t_.name for t_ in ex.tr.items]
        res[overlap] = (float(loss_t), rel_l2(grads_g, tr_b.flat_grad), bool(torch.equal(tr_b.flat_grad, grads_g)))
        del eng_b
    _record("graphed_b8", loss_graph=loss_g, loss_eager_two_stream=res[True][0], loss_eager_one_stream=res[False][0],
            graph_vs_eager_two_stream=res[True][1], bit_identical_two_stream=res[True][2],
            graph_vs_eager_one_stream=res[False][1])
    assert abs(loss_g - res[True][0]) < 1e-6 * abs(loss_g) and res[True][1] < 1e-6
    assert abs(loss_g - res[False][0]) < 1e-4 * abs(loss_g) and res[False][1] < 5e-3
    # (b) fp32 engine and the oracle on the GPU
    eng_f = CtrLoRAEngine(sd_un, [sd_cn], _netcfg(cfg), dtype=torch.float32, device="cuda")
    eps_f = eng_f.forward(x_noisy, t, ctx, [hint], record=True)
    eng_f.zero_grad()
    eng_f.backward(2.0 * (eps_f - noise) / eps_f.numel())
    torch.cuda.synchronize()
    loss_o, eps_o, grads_o = _oracle_on_gpu(cfg, sd_cn, sd_un, z, t, ctx, hint, noise)
    e_f = rel_l2(eps_f, eps_o)
    gf = sorted(((rel_l2(t_.grad, grads_o[t_.name]), t_.name) for t_ in eng_f.controls[0].tr.items), reverse=True)
    eng_g = model.engine()
    eps_b = eng_g.forward(x_noisy, t, ctx, [hint])            # the graphed model's own executors, no-grad forward
    e_b = rel_l2(eps_b, eps_o)
    by_name = {t_.name: t_ for t_ in ex.tr.items}
    gb = sorted(((rel_l2(grads_g[t_.offset:t_.offset + t_.master.numel()].view(t_.shape), grads_o[n]), n)
                 for n, t_ in by_name.items()), reverse=True)
    _record("graphed_b8_vs_oracle", eps_f32=e_f, grad_f32_max=gf[0][0], eps_bf16=e_b, grad_bf16_max=gb[0][0],
            grad_bf16_max_name=gb[0][1], grad_bf16_median=gb[len(gb) // 2][0], loss_oracle=loss_o)
    assert len(gf) == 246 and len(gb) == 246
    assert e_f < 1e-4 and gf[0][0] < 5e-4, (e_f, gf[:3])
    assert abs(loss_g - loss_o) < 2e-2 * loss_o
    c_eps, c_max, c_med = _comparator_vs(eps_o, lambda n, g: rel_l2(g, grads_o[n]), cfg, sd_cn, sd_un, z, t, ctx, hint, noise)
    _record("graphed_b8_vs_oracle_comparator", eps=c_eps, grad_max=c_max, grad_median=c_med)
    assert e_b < K_CMP * c_eps and gb[0][0] < K_CMP_MAX * c_max and gb[len(gb) // 2][0] < K_CMP * c_med, (e_b, gb[:3], c_eps, c_max, c_med)
    assert e_b < BF16_EPS and gb[0][0] < BF16_GRAD_MAX and gb[len(gb) // 2][0] < BF16_GRAD_MEDIAN, (e_b, gb[:3])


def test_graphed_ddim_b16_latent64_matches_eager_loop():
    """BASELINE configs[4] as bench.py runs it (rank-128 inference YAML, B = 16, CFG 7.5 batched as 2B = 32, latent
    64x64, bf16): the hipGraph-replayed loop gives the eager loop's samples."""
    _need_gpu()
    import bench
    from cldm.ddim_hacked import DDIMSampler
    model = bench.build_model("inference/ctrlora_sd15_rank128_1lora.yaml", 0).cuda().eval()
    model.set_engine_dtype(torch.bfloat16)
    B, H, S = 16, 64, 5
    g = torch.Generator().manual_seed(7)
    cd = model.control_model.context_dim
    hint = torch.randn(B, 4, H, H, generator=g).cuda()
    cond = {"c_concat": [hint], "c_crossattn": [torch.randn(B, 77, cd, generator=g).cuda()]}
    unc = {"c_concat": [hint], "c_crossattn": [torch.randn(B, 77, cd, generator=g).cuda()]}
    x_T = torch.randn(B, 4, H, H, generator=g).cuda()
    outs = []
    for use_graph in (True, False):
        s = DDIMSampler(model)
        s.use_graph = use_graph
        x, inter = s.sample(S, B, (4, H, H), cond, verbose=False, eta=0.0, x_T=x_T, unconditional_guidance_scale=7.5,
                            unconditional_conditioning=unc)
        assert torch.isfinite(x).all()
        outs.append(x)
        assert list(s.ddim_timesteps) == [1, 201, 401, 601, 801]
    d = rel_l2(outs[0], outs[1])
    _record("ddim_graph_vs_eager_b16", rel=d, bit_identical=bool(torch.equal(outs[0], outs[1])))
    assert d < 1e-6


# ------------------------------------------------------------------------------ kernels at production shapes

def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("B,H,Cin,Cout,mode", [
    (8, 64, 320, 320, "s1"), (8, 64, 960, 320, "s1"), (8, 32, 640, 640, "s1"), (8, 16, 2560, 1280, "s1"),
    (8, 8, 1280, 1280, "s1"), (8, 64, 320, 320, "s2"), (8, 16, 1280, 1280, "up2"), (2, 64, 320, 320, "bwd")])
def test_conv3x3_production_shapes(B, H, Cin, Cout, mode):
    """Implicit-GEMM 3x3 conv (gemm_fl ping-pong tiles, split-K at 8x8) vs torch's fp32 conv on the same
    bf16-rounded operands; reference openaimodel.py:254-274, 108-118, 157-159."""
    _need_gpu()
    from ctrlora_amd import hip
    from ctrlora_amd.engine.blocks import Ctx, conv3_bwd_data, conv3_fwd
    from ctrlora_amd.engine.packing import Conv3W
    F = torch.nn.functional
    g = torch.Generator().manual_seed(B * H + Cin)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / (3 * Cin ** 0.5))
    b = torch.randn(Cout, generator=g) * 0.1
    x = torch.randn(B, Cin, H, H, generator=g)
    cw = Conv3W(w, b, torch.bfloat16, "cuda", True)
    ctx = Ctx(torch.bfloat16, torch.device("cuda"), False)
    xt = _bf(x.permute(0, 2, 3, 1).reshape(B * H * H, Cin)).cuda().contiguous()
    wr, xr = _bf(w).float().cuda(), xt.float().reshape(B, H, H, Cin).permute(0, 3, 1, 2)
    if mode == "s1":
        y = conv3_fwd(ctx, cw, xt, B, H, H)
        ref = F.conv2d(xr, wr, b.cuda(), padding=1)
    elif mode == "s2":
        y = conv3_fwd(ctx, cw, xt, B, H, H, mode=hip.CONV_S2)
        ref = F.conv2d(xr, wr, b.cuda(), stride=2, padding=1)
    elif mode == "up2":
        y = conv3_fwd(ctx, cw, xt, B, H, H, mode=hip.CONV_UP2)
        ref = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), wr, b.cuda(), padding=1)
    else:   # data gradient of the stride-1 conv: dy has Cout channels on the same grid
        dy = torch.randn(B, Cout, H, H, generator=g)
        dyt = _bf(dy.permute(0, 2, 3, 1).reshape(B * H * H, Cout)).cuda().contiguous()
        y = conv3_bwd_data(ctx, cw, dyt, B, H, H)
        ref = F.conv_transpose2d(dyt.float().reshape(B, H, H, Cout).permute(0, 3, 1, 2), wr, padding=1)
    Ho = ref.shape[2]
    got = y.float().reshape(B, Ho, Ho, -1).permute(0, 3, 1, 2)[:, :ref.shape[1]]
    e = rel_l2(got, ref)
    _record("conv3x3", shape=[B, H, Cin, Cout, mode], rel=e)
    assert e < 4e-3


def _attention_case(dh, N, Nkv, B, prescaled, spike=False, variant=0, q_std=1.0):
    """Run forward + backward through the C ABI and compare with fp64 softmax(q k^T * scale) v evaluated on the q the
    kernel's input MEANS (q' / (scale log2 e) under the pre-scaled-Q contract).  Returns the five relative errors."""
    from ctrlora_amd import hip
    Hh = 8
    inner = Hh * dh
    g = torch.Generator().manual_seed(dh * 7 + N + Nkv + (13 if spike else 0))
    mk32 = lambda n, s=1.0: (torch.randn(B * n, inner, generator=g) * s).cuda()
    q32 = mk32(N, q_std)
    k, v, do = (_bf(mk32(Nkv, q_std).cpu()).cuda(), _bf(mk32(Nkv).cpu()).cuda(), _bf(mk32(N).cpu()).cuda())
    if spike:      # the last keys of every sample line up with its first queries: scores jump by ~80 nats in the LAST tile
        kk = k.float().reshape(B, Nkv, inner)
        kk[:, -16:, :] = q32.reshape(B, N, inner)[:, :16, :] * 6.0
        k = kk.reshape(B * Nkv, inner).to(torch.bfloat16)
    scale = dh ** -0.5
    c = scale * 1.4426950408889634
    q = (q32 * c).to(torch.bfloat16) if prescaled else q32.to(torch.bfloat16)       # ONE rounding either way
    q_true = q.double() / c if prescaled else q.double()
    o = torch.empty_like(q)
    rp = (N + 63) // 64 * 64
    lse = torch.empty(B, Hh, rp, dtype=torch.float32, device="cuda")
    delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    assert hip.lib().cl_debug_attention_variant(variant) == 0
    try:
        hip.attention_fwd_v2(q, k, v, o, lse, B, Hh, N, Nkv, dh, scale, q_prescaled=prescaled)
        row_ws = torch.empty(lse.numel() * 8, dtype=torch.float32, device="cuda") if prescaled else None   # as the engine does
        hip.attention_bwd_v2(q, k, v, o, do, lse, delta, dq, dk, dv, B, Hh, N, Nkv, dh, scale, q_prescaled=prescaled, row_ws=row_ws)
        torch.cuda.synchronize()
    finally:
        hip.lib().cl_debug_attention_variant(0)
    split = lambda x, n: x.double().reshape(B, n, Hh, dh).permute(0, 2, 1, 3).requires_grad_(True)
    qr, kr, vr = split(q_true, N), split(k, Nkv), split(v, Nkv)
    s = torch.einsum("bhid,bhjd->bhij", qr, kr) * scale
    orf = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), vr)
    orf.backward(do.double().reshape(B, N, Hh, dh).permute(0, 2, 1, 3))
    back = lambda x, n: x.permute(0, 2, 1, 3).reshape(B * n, inner)
    # the kernels keep log-sum-exp in the exp2 domain (log2 of the softmax denominator of scale * log2(e) * s)
    return dict(o=rel_l2(o, back(orf, N)), lse=rel_l2(lse[:, :, :N] * 0.6931471805599453, torch.logsumexp(s, -1)),
                dq=rel_l2(dq, back(qr.grad, N)), dk=rel_l2(dk, back(kr.grad, Nkv)), dv=rel_l2(dv, back(vr.grad, Nkv)))


@pytest.mark.parametrize("prescaled", [False, True])
@pytest.mark.parametrize("dh,N,Nkv,B", [(40, 4096, 4096, 8), (40, 4096, 77, 8), (80, 1024, 1024, 8), (80, 1024, 77, 8),
                                         (160, 256, 256, 8), (160, 256, 77, 8), (160, 64, 64, 8), (160, 64, 77, 8),
                                         (40, 4096, 4096, 1), (32, 500, 500, 2)])
def test_attention_production_shapes_fwd_bwd(dh, N, Nkv, B, prescaled):
    """bf16 flash attention (fwd, dK/dV, dQ) at every (d_head, N, N_kv) of SD1.5 at 512x512, 8 heads, vs fp64
    softmax(QK^T * scale)V on the same bf16-rounded operands; reference ldm/modules/attention.py:163-194.  Both input
    contracts: plain q, and q pre-multiplied by d_head^-0.5 log2(e) (CL_ATTN_Q_PRESCALED: what the engine's to_q projections
    write; at d_head 40 / N 4096 this is the software-pipelined forward of csrc/attention_fwd40.hip).  The gradients are
    those of the TRUE q, k, v in both."""
    _need_gpu()
    e = _attention_case(dh, N, Nkv, B, prescaled)
    _record("attention", shape=[dh, N, Nkv, B], prescaled=prescaled, **e)
    assert e["o"] < 6e-3 and e["lse"] < 1e-5, e
    assert max(e["dq"], e["dk"], e["dv"]) < 1e-2, e


def test_attention_prescaled_forward_second_pass_on_runaway_scores(variant=0):
    """The pre-scaled-Q forward subtracts the row maximum of the FIRST key tile only and checks every row's denominator at
    the end (csrc/attention_fwd40.hip): scores that outgrow that maximum by ~120 log2 units overflow it, and the workgroup
    must repeat its block with the conventional running maximum.  Here the last 16 keys of every sample are aligned with
    its first 16 queries (q.k ~ 540, ~85 nats above everything seen before): results must still match fp64; and an ordinary input with LARGE logits (std 4: row maxima ~ 40 nats, far
    above tile 0's) must pass without NaN / Inf.
    LSE gate: the denominator is the matrix-pipe sum of the bf16-ROUNDED P; for a row that one key dominates it carries that
    key's rounding (<= 2^-9 relative: 2.8e-3 absolute in log2 units, measured 6e-5 relative here), where rows with thousands
    of comparable terms average it out (2e-5 at the production shapes)."""
    _need_gpu()
    e = _attention_case(40, 4096, 4096, 2, True, spike=True, variant=variant)
    _record("attention_spike", variant=variant, **e)
    assert e["o"] < 8e-3 and e["lse"] < 2e-4 and max(e["dq"], e["dk"], e["dv"]) < 1.5e-2, e
    e = _attention_case(40, 4096, 4096, 1, True, variant=variant, q_std=4.0)
    _record("attention_large_logits", variant=variant, **e)
    assert e["o"] < 8e-3 and e["lse"] < 2e-4 and max(e["dq"], e["dk"], e["dv"]) < 1.5e-2, e


@pytest.mark.parametrize("spread", [1.0, 6.0])
def test_attention_prescaled_forward_is_a_pure_function_of_its_inputs(spread):
    """attn_fwd40_kernel issues its MFMAs from inline asm, so every wait state in front of a reader of their results is
    the kernel's own business.  One was missing in round 4's first version (the epilogue's read of the denominator row was
    scheduled in front of the drain): workgroups fell into the second pass at random and ~2 % of the outputs flipped by one
    bf16 ulp from launch to launch -- within every accuracy gate, caught only by the replay-determinism check of the graphed
    step.  Here: eight launches on the same inputs are bitwise equal, with scores that stay in the optimistic range
    (spread 1) and with scores that send part of the workgroups through the second pass (spread 6)."""
    _need_gpu()
    from ctrlora_amd import hip
    B, H, N, dh = 8, 8, 4096, 40
    inner = H * dh
    g = torch.Generator().manual_seed(11)
    mk = lambda s=1.0: (torch.randn(B * N, inner, generator=g) * s).cuda()
    q = (mk(1.5) * (spread * dh ** -0.5 * 1.4426950408889634)).to(torch.bfloat16)
    k, v = mk(1.5).to(torch.bfloat16), mk().to(torch.bfloat16)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
    outs = []
    for _ in range(8):
        o.fill_(float("nan")); lse.fill_(float("nan"))
        hip.attention_fwd_v2(q, k, v, o, lse, B, H, N, N, dh, dh ** -0.5, q_prescaled=True)
        torch.cuda.synchronize()
        outs.append((o.clone(), lse.clone()))
    assert torch.isfinite(outs[0][0].float()).all() and torch.isfinite(outs[0][1]).all()
    for a, b in outs[1:]:
        assert torch.equal(a, outs[0][0]) and torch.equal(b, outs[0][1])


def test_attention_schedules_agree():
    """Every schedule of the bf16 attention kernels that ships -- tile-synchronous (variant 1: the round-1 kernels, still used
    for ragged shapes), the hybrid ping-pong forward (14, with and without the flag), the pre-scaled-Q forward (0 with the flag) -- gives the same O / lse / dQ / dK / dV up to bf16 rounding at N = 4096, d_head 40 and N = 1024, d_head 80."""
    _need_gpu()
    for dh, N, B in ((40, 4096, 2), (80, 1024, 4)):
        errs = {}
        for name, variant, pre in (("sync", 1, False), ("hybrid", 14, False), ("sync+prescaled", 1, True),
                                   ("hybrid+prescaled", 14, True), ("default+prescaled", 0, True),
                                   ("dkv 64 keys per wave (round-6 probe)+prescaled", 21, True)):
            errs[name] = _attention_case(dh, N, N, B, pre, variant=variant, q_std=1.3)
        _record("attention_schedules", dh=dh, N=N, **{k: v["o"] for k, v in errs.items()})
        worst = {k: max(v["o"], v["dq"], v["dk"], v["dv"]) for k, v in errs.items()}
        assert all(v["lse"] < 5e-5 for v in errs.values()) and max(worst.values()) < 8e-3, errs
        assert max(worst.values()) < 1.5 * min(worst.values()) + 1e-3, worst      # no schedule is an outlier


@pytest.mark.parametrize("M,K,N,r", [(32768, 320, 320, 128), (8192, 640, 640, 128), (2048, 1280, 1280, 128),
                                      (32768, 320, 2560, 128), (32768, 1280, 320, 128), (8 * 77, 768, 320, 128),
                                      (512, 1280, 1280, 128), (8, 1280, 1280, 128), (32768, 320, 320, 32)])
def test_lora_fused_linear_production_shapes(M, K, N, r):
    """LoRACompatibleLinear forward / data gradient / LoRA weight gradients with the rank-r branch fused into the
    main MFMA chain, at the (M, K, N) of SD1.5 at 512x512, B = 8; reference cldm/lora.py:285-291."""
    _need_gpu()
    from ctrlora_amd.engine.blocks import Ctx, linear_bwd_data, linear_bwd_lora, linear_fwd
    from ctrlora_amd.engine.packing import LinearW, TrainableSet
    g = torch.Generator().manual_seed(M + K + N + r)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    A = torch.randn(r, K, generator=g) / r
    Bm = torch.randn(N, r, generator=g) * 0.05
    tr = TrainableSet()
    L = LinearW(W, bias, torch.bfloat16, "cuda", True)
    tA, tB = tr.declare("a", A.shape), tr.declare("b", Bm.shape)
    L.attach_lora(tA, tB, "cuda")
    tr.materialize({"a": A, "b": Bm}, "cuda")
    L.repack()
    ctx = Ctx(torch.bfloat16, torch.device("cuda"), True)
    x = _bf(torch.randn(M, K, generator=g)).cuda()
    dy = _bf(torch.randn(M, N, generator=g)).cuda()
    y, t = linear_fwd(ctx, L, x)
    dx, u = linear_bwd_data(ctx, L, dy)
    linear_bwd_lora(ctx, L, x, t, dy, u)
    ctx.flush_wgrad()
    torch.cuda.synchronize()
    d = lambda v: v.double().cuda()
    xr = x.double().requires_grad_(True)
    Wr, Ar, Br = d(_bf(W).float()), d(_bf(A).float()).requires_grad_(True), d(_bf(Bm).float()).requires_grad_(True)
    tr_ = xr @ Ar.t()
    yr = xr @ Wr.t() + d(bias) + tr_ @ Br.t()
    yr.backward(dy.double())
    e = dict(y=rel_l2(y, yr), dx=rel_l2(dx, xr.grad), dA=rel_l2(tA.grad, Ar.grad), dB=rel_l2(tB.grad, Br.grad))
    _record("lora_linear", shape=[M, K, N, r], **e)
    # y / dx are bf16 outputs of a product whose low-rank intermediate (x A^T, dy B) is itself rounded to bf16
    assert e["y"] < 5e-3 and e["dx"] < 5e-3
    assert e["dA"] < 8e-3 and e["dB"] < 8e-3


def test_grouped_weight_gradient_production_stage():
    """One backward stage's weight gradients as ONE grouped launch (cl_weight_grad_tn_group): the problem list of a
    64x64-level transformer block + its zero conv at B = 8 (M = 32768), fp32 accumulation onto existing values."""
    _need_gpu()
    from ctrlora_amd import hip
    g = torch.Generator().manual_seed(5)
    M = 32768
    shapes = [(320, 128), (128, 320), (320, 128), (128, 320), (2560, 128), (128, 320), (320, 128), (128, 1280),
              (320, 320), (1280, 128), (128, 1280)]
    shapes += [(320, 128), (128, 320)] * 6        # 23 problems: one short of the 24-descriptor flush limit
    probs, refs = [], []
    for i, (N, K) in enumerate(shapes):
        dy = _bf(torch.randn(M, N, generator=g)).cuda()
        x = _bf(torch.randn(M, K, generator=g)).cuda()
        dW = torch.ones(N, K, device="cuda")
        scale = 1.0 if i % 3 else 0.5
        probs.append((dy, x, dW, scale))
        refs.append(1.0 + scale * (dy.double().t() @ x.double()))
    hip.weight_grad_tn_group(probs)
    torch.cuda.synchronize()
    worst = max(rel_l2(p[2], r) for p, r in zip(probs, refs))
    _record("grouped_wgrad", worst=worst, n=len(probs))
    assert worst < 2e-5
