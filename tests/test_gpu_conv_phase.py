"""Phase-decomposed Upsample conv and Downsample data gradient (CL_GEMM_CONV_UP2P / CL_GEMM_CONV_T2P; include/ctrlora_hip.h,
csrc/gemm.h) against fp64 references of the reference's own operators -- Upsample.forward = nearest x2 then conv3x3
(ldm/modules/diffusionmodules/openaimodel.py:108-118), and autograd's data gradient of Downsample's stride-2 conv (:150) --
and against the 9-tap modes (CL_GEMM_CONV_UP2 / T2) on the same call.  The decomposition changes only which MACs are issued:
results must agree to rounding (fp32: 1e-5; bf16: the two forms sit at the same distance from fp64).
"""
import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu
F = torch.nn.functional


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")


def _setup(B, H, W, Cin, Cout, dtype, seed):
    from ctrlora_amd.engine.blocks import Ctx
    from ctrlora_amd.engine.packing import Conv3W
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / (3 * Cin ** 0.5))
    b = torch.randn(Cout, generator=g) * 0.1
    cw = Conv3W(w, b, dtype, "cuda", True)
    ctx = Ctx(dtype, torch.device("cuda"), False)
    rnd = (lambda t: t.to(dtype).double()) if dtype == torch.bfloat16 else (lambda t: t.double())
    return g, w, b, cw, ctx, rnd


def _tok(t, dtype):      # NCHW -> [B*H*W, C] in the compute dtype, on the GPU
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C).to(dtype).cuda().contiguous()


def _img(t, B, H, W):    # tokens -> NCHW fp64 on the CPU
    return t.double().cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2)


UP_CASES = [
    # B, H, W, Cin, Cout      (SD1.5 decoder Upsample convs at batch 8; a DDIM-sized one; ragged grids; wide-to-narrow)
    (8, 8, 8, 1280, 1280), (8, 16, 16, 1280, 1280), (8, 32, 32, 640, 640), (32, 16, 16, 1280, 1280),
    (2, 8, 16, 320, 640), (1, 16, 8, 640, 320), (3, 16, 16, 128, 256),
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 4e-3)])
@pytest.mark.parametrize("case", UP_CASES)
def test_upsample_conv_phase_form_matches_fp64_and_the_nine_tap_form(case, dtype, tol):
    _need_gpu()
    from ctrlora_amd import hip
    from ctrlora_amd.engine import blocks
    B, H, W, Cin, Cout = case
    if dtype == torch.bfloat16 and Cin % 64:
        pytest.skip("bf16 needs whole 128-byte channel lines")
    g, w, b, cw, ctx, rnd = _setup(B, H, W, Cin, Cout, dtype, B * H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    emb = torch.randn(B, Cout, generator=g) * 0.2
    res = torch.randn(B, Cout, 2 * H, 2 * W, generator=g)
    xt, rt, et = _tok(x, dtype), _tok(res, dtype), emb.to(dtype).cuda().contiguous()
    ref = F.conv2d(F.interpolate(_img(xt, B, H, W), scale_factor=2, mode="nearest"), rnd(w), b.double(), padding=1)
    ref = ref + et.double().cpu()[:, :, None, None] + _img(rt, B, 2 * H, 2 * W)
    outs = {}
    for phase in (True, False):
        blocks.CONV_PHASE = phase
        try:
            assert blocks._phase_ok(ctx, cw, B, H, W, cw.Ip, cw.Op) == phase
            y = blocks.conv3_fwd(ctx, cw, xt, B, H, W, mode=hip.CONV_UP2, rowbias=et, residual=rt)
            y2 = blocks.conv3_fwd(ctx, cw, xt, B, H, W, mode=hip.CONV_UP2, rowbias=et, residual=rt)
        finally:
            blocks.CONV_PHASE = True
        assert torch.equal(y, y2)                                     # bitwise repeatable
        outs[phase] = _img(y, B, 2 * H, 2 * W)[:, :Cout]
    e_new, e_old = rel_l2(outs[True], ref), rel_l2(outs[False], ref)
    print(f"[conv phase] up2 {case} {dtype}: phase form {e_new:.2e}, nine-tap form {e_old:.2e}")
    assert e_new < tol and e_old < tol
    assert e_new < 1.5 * e_old + 1e-6          # summing the coincident taps (one extra rounding of the weights) costs nothing
    # plain call (no bias-like extras beyond the conv's own bias), output buffer given
    blocks.CONV_PHASE = True
    out = torch.empty(4 * B * H * W, cw.Op, dtype=dtype, device="cuda")
    y = blocks.conv3_fwd(ctx, cw, xt, B, H, W, mode=hip.CONV_UP2, out=out)
    assert y is out
    ref0 = F.conv2d(F.interpolate(_img(xt, B, H, W), scale_factor=2, mode="nearest"), rnd(w), b.double(), padding=1)
    assert rel_l2(_img(y, B, 2 * H, 2 * W)[:, :Cout], ref0) < tol


T2_CASES = [
    # B, Hdy, Wdy, Cin, Cout of the FORWARD stride-2 conv (dy has Cout channels on the Hdy x Wdy grid; dx is 2Hdy x 2Wdy)
    (8, 32, 32, 320, 320), (8, 16, 16, 640, 640), (8, 8, 8, 1280, 1280), (2, 8, 16, 320, 640), (1, 16, 8, 640, 320),
    (4, 16, 16, 128, 128),
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 4e-3)])
@pytest.mark.parametrize("case", T2_CASES)
def test_downsample_data_gradient_phase_form_matches_autograd_and_the_nine_tap_form(case, dtype, tol):
    _need_gpu()
    from ctrlora_amd import hip
    from ctrlora_amd.engine import blocks
    B, H, W, Cin, Cout = case
    if dtype == torch.bfloat16 and Cout % 64:
        pytest.skip("bf16 needs whole 128-byte channel lines")
    g, w, b, cw, ctx, rnd = _setup(B, H, W, Cin, Cout, dtype, B * W + Cout)
    dy = torch.randn(B, Cout, H, W, generator=g)
    acc = torch.randn(B, Cin, 2 * H, 2 * W, generator=g)
    dyt, at = _tok(dy, dtype), _tok(acc, dtype)
    xin = torch.zeros(B, Cin, 2 * H, 2 * W, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin, rnd(w), None, stride=2, padding=1).backward(_img(dyt, B, H, W))
    ref = xin.grad + _img(at, B, 2 * H, 2 * W)
    outs = {}
    rule = blocks._phase_ok
    # the engine takes the phase form only where its tile grid fills the chip (no K split): 512 / 256 tiles at the two upper
    # levels of the batch-8 step, 128 at the 8x8 level (stays on the nine-tap mode's split-K)
    assert rule(ctx, cw, B, H, W, cw.Op, cw.Ip, t2=True) == (4 * B * H * W // 128 * ((cw.Ip + 159) // 160) >= 192)
    for phase in (True, False):
        blocks.CONV_PHASE = phase
        blocks._phase_ok = lambda *a, **k: rule(*a, **{**k, "t2": False})     # the KERNEL is exercised at every shape
        try:
            assert blocks._phase_ok(ctx, cw, B, H, W, cw.Op, cw.Ip) == phase
            dx = blocks.conv3_bwd_data(ctx, cw, dyt, B, H, W, fwd_mode=hip.CONV_S2, accum=at)
            dx2 = blocks.conv3_bwd_data(ctx, cw, dyt, B, H, W, fwd_mode=hip.CONV_S2, accum=at)
        finally:
            blocks.CONV_PHASE = True
            blocks._phase_ok = rule
        assert torch.equal(dx, dx2)
        outs[phase] = _img(dx, B, 2 * H, 2 * W)[:, :Cin]
    e_new, e_old = rel_l2(outs[True], ref), rel_l2(outs[False], ref)
    print(f"[conv phase] t2 {case} {dtype}: phase form {e_new:.2e}, nine-tap form {e_old:.2e}")
    assert e_new < tol and e_old < tol
    # the two forms issue the same non-zero products (the nine-tap form adds exact zeros): they differ by summation order only
    assert rel_l2(outs[True], outs[False]) < (1e-6 if dtype == torch.float32 else 3e-3)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 4e-3)])
@pytest.mark.parametrize("case", UP_CASES)
def test_upsample_conv_data_gradient_on_the_source_grid_matches_autograd(case, dtype, tol):
    """d/dx of conv3x3(nearest_x2(x)) (openaimodel.py:108-118 under autograd) as ONE 4x4 stride-2 window product on the source
    grid (CL_GEMM_CONV_S2K4) against fp64 autograd and against the stride-1 data gradient + 2x2 sum pool it replaces."""
    _need_gpu()
    from ctrlora_amd import hip
    from ctrlora_amd.engine import blocks
    B, H, W, Cin, Cout = case
    if dtype == torch.bfloat16 and Cout % 64:
        pytest.skip("bf16 needs whole 128-byte channel lines")
    g, w, b, cw, ctx, rnd = _setup(B, H, W, Cin, Cout, dtype, B + H + Cout)
    dy = torch.randn(B, Cout, 2 * H, 2 * W, generator=g)
    dyt = _tok(dy, dtype)
    xin = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(F.interpolate(xin, scale_factor=2, mode="nearest"), rnd(w), None, padding=1).backward(_img(dyt, B, 2 * H, 2 * W))
    ref = xin.grad
    outs = {}
    for phase in (True, False):
        blocks.CONV_PHASE = phase
        try:
            dx = blocks.conv3_bwd_data(ctx, cw, dyt, B, 2 * H, 2 * W, fwd_mode=hip.CONV_UP2)
            dx2 = blocks.conv3_bwd_data(ctx, cw, dyt, B, 2 * H, 2 * W, fwd_mode=hip.CONV_UP2)
        finally:
            blocks.CONV_PHASE = True
        assert torch.equal(dx, dx2) and dx.shape == (B * H * W, cw.Ip)
        outs[phase] = _img(dx, B, H, W)[:, :Cin]
    e_new, e_old = rel_l2(outs[True], ref), rel_l2(outs[False], ref)
    print(f"[conv phase] up2 data gradient {case} {dtype}: source-grid form {e_new:.2e}, stride-1 + pool form {e_old:.2e}")
    assert e_new < tol and e_old < 2 * tol          # (the old form rounds the upsampled-grid gradient to bf16 before pooling)
    assert e_new < 1.5 * e_old + 1e-6


def test_phase_forms_fall_back_where_they_do_not_apply():
    """Trainable convs (pre-training repacks their weights every step), grids that do not fill a 128-row tile and channel
    counts that are not whole lines keep the nine-tap modes; the library refuses malformed phase calls instead of guessing."""
    _need_gpu()
    from ctrlora_amd import hip
    from ctrlora_amd.engine import blocks
    g, w, b, cw, ctx, _ = _setup(1, 4, 4, 64, 64, torch.bfloat16, 3)
    assert not blocks._phase_ok(ctx, cw, 1, 4, 4, cw.Ip, cw.Op)               # 16 source pixels
    assert blocks._phase_ok(ctx, cw, 2, 8, 8, cw.Ip, cw.Op) is False          # N = 64 < 96
    g, w, b, cw, ctx, _ = _setup(2, 8, 8, 96, 128, torch.bfloat16, 4)
    assert not blocks._phase_ok(ctx, cw, 2, 8, 8, cw.Ip, cw.Op)               # 96 channels: not whole 128-byte lines in bf16
    x = torch.randn(2 * 8 * 8, cw.Ip, device="cuda").to(torch.bfloat16)
    y = blocks.conv3_fwd(ctx, cw, x, 2, 8, 8, mode=hip.CONV_UP2)              # runs on the nine-tap mode
    assert y.shape == (4 * 128, cw.Op) and bool(torch.isfinite(y.float()).all())
    out = torch.empty(4 * 128, cw.Op, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(hip.HipError):
        hip.gemm(x, cw.Wp, out, mode=hip.CONV_UP2P, conv=(2, 8, 8, 16, 16), k1=cw.Ip, N=cw.Op)
    with pytest.raises(hip.HipError):                                         # M must be 4 B Hin Win
        g2, w2, b2, cw2, ctx2, _ = _setup(2, 8, 8, 128, 128, torch.bfloat16, 5)
        x2 = torch.randn(128, 128, device="cuda").to(torch.bfloat16)
        hip.gemm(x2, cw2.phase_weights("up2"), out[:256], mode=hip.CONV_UP2P, conv=(2, 8, 8, 16, 16), k1=128, N=128, M=256)
