"""GPU parity of the loader / consumer tile kernel (csrc/gemm_w4.hip, launch configurations 40 / 41 and their persistent forms 47 / 48)
through the C ABI (cl_gemm):
ResBlock 3x3 convolutions (ldm/modules/diffusionmodules/openaimodel.py:203,229) in both of its forms -- the halo-resident image
(stride-1 convs whose 256-row tile lies inside one image: 64x64, 32x32, 16x16 levels) and the per-tap DMA form (everything else) --
and deep-K linears with the LoRA K segment (cldm/lora.py:285-291).  References: torch conv2d / matmul in fp64 on the bf16-rounded
operands (fp32 output: 2e-5 = fp32 summation order; bf16 output: 2.5e-3 = one rounding), the ping-pong tile kernels on the same
call, and the kernel itself launched again (bit-identical: the LDS ring protocol has no data race)."""
import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ctrlora_amd import hip
    return hip, hip.lib()


def _mk(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("B,H,W,C,N,cfg,what", [
    (2, 64, 64, 320, 320, 40, "halo, 64x64 level, five chunks"),
    (8, 32, 32, 640, 640, 40, "halo, 32x32 level, K split over workgroups"),
    (4, 16, 16, 1280, 320, 40, "halo, 16x16 level: one image per tile"),
    (1, 64, 64, 64, 160, 40, "halo, a single chunk"),
    (3, 32, 32, 128, 128, 41, "halo, 128-column tiles"),
    (3, 33, 31, 640, 320, 40, "per-tap DMA: ragged geometry"),
    (2, 8, 8, 1280, 1280, 40, "per-tap DMA: tiles span images (8x8 level)"),
])
def test_w4_conv3x3_vs_fp64_and_pingpong_tiles(B, H, W, C, N, cfg, what):
    hip, L = _need_gpu()
    g = torch.Generator().manual_seed(B * 1000 + H + C + N)
    x = _mk(g, B * H * W, C)
    w = _mk(g, N, 9 * C, scale=0.03)                      # [O][ky][kx][I]
    bias = torch.randn(N, generator=g)
    want = torch.nn.functional.conv2d(x.double().view(B, H, W, C).permute(0, 3, 1, 2),
                                      w.double().view(N, 3, 3, C).permute(0, 3, 1, 2), bias.double(), padding=1)
    want = want.permute(0, 2, 3, 1).reshape(B * H * W, N)
    outs = []
    try:
        for c in (cfg, cfg, -1):
            L.cl_gemm_force_config(c)
            y = torch.full((B * H * W, N), float("nan"), dtype=torch.float32, device="cuda")
            hip.gemm(x.cuda(), w.cuda(), y, bias=bias.cuda(), mode=hip.CONV_S1, conv=(B, H, W, H, W), k1=C, out_f32=True,
                     dtype=hip.BF16)
            outs.append(y.cpu())
    finally:
        L.cl_gemm_force_config(-1)
    assert rel_l2(outs[0].double(), want) < 2e-5, what
    assert torch.equal(outs[0], outs[1]), what                 # bitwise repeatable
    assert rel_l2(outs[0].double(), outs[2].double()) < 2e-5   # the launcher's own choice agrees


@pytest.mark.parametrize("M,N,K,r,cfg", [(4096, 320, 1280, 128, 40), (2048, 1280, 5120, 0, 40), (1000, 640, 2560, 0, 40),
                                          (2100, 512, 1024, 64, 41), (300, 160, 256, 0, 40)])
def test_w4_linear_lora_residual_rowbias_vs_fp64(M, N, K, r, cfg):
    hip, L = _need_gpu()
    g = torch.Generator().manual_seed(M + N + K + r)
    x, W = _mk(g, M, K), _mk(g, N, K, scale=0.03)
    t = _mk(g, M, r) if r else None
    Bup = _mk(g, N, r, scale=0.05) if r else None
    bias, res = torch.randn(N, generator=g), _mk(g, M, N)
    want = x.double() @ W.double().t() + bias.double()
    if r:
        want += t.double() @ Bup.double().t()
    want = 0.5 * want + 2.0 * res.double()
    cu = lambda v: None if v is None else v.cuda()
    outs = []
    try:
        for c in (cfg, cfg):
            L.cl_gemm_force_config(c)
            y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            hip.gemm(cu(x), cu(W), y, a2=cu(t), w2=cu(Bup), bias=cu(bias), residual=cu(res), alpha=0.5, beta=2.0)
            outs.append(y.cpu())
    finally:
        L.cl_gemm_force_config(-1)
    assert rel_l2(outs[0].double(), want) < 2.5e-3
    assert torch.equal(outs[0], outs[1])


def test_w4_conv_time_embedding_rowbias_silu_and_fused_geglu():
    """The ResBlock's `h + emb_out[:, :, None, None]` epilogue (openaimodel.py:272) + SiLU on a halo conv, and the fused GEGLU of the
    inference forwards (attention.py:49-56) with its value / gate pairing done in registers."""
    hip, L = _need_gpu()
    g = torch.Generator().manual_seed(77)
    B, H, C, N = 2, 32, 320, 320
    x, w = _mk(g, B * H * H, C), _mk(g, N, 9 * C, scale=0.03)
    bias, emb = torch.randn(N, generator=g), _mk(g, B, N)
    want = torch.nn.functional.conv2d(x.double().view(B, H, H, C).permute(0, 3, 1, 2), w.double().view(N, 3, 3, C).permute(0, 3, 1, 2),
                                      bias.double(), padding=1) + emb.double()[:, :, None, None]
    want = torch.nn.functional.silu(want).permute(0, 2, 3, 1).reshape(B * H * H, N)
    try:
        L.cl_gemm_force_config(40)
        y = torch.full((B * H * H, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        hip.gemm(x.cuda(), w.cuda(), y, bias=bias.cuda(), rowbias=emb.cuda(), rows_per_batch=H * H, act=hip.ACT_SILU, mode=hip.CONV_S1,
                 conv=(B, H, H, H, H), k1=C)
        # fused GEGLU: W's rows permuted per 160-column tile [80 value | 80 gate] (the tile kernels' act 2 contract)
        M, half, K = 1000, 320, 1280
        xg, Wg, bg = _mk(g, M, K), _mk(g, 2 * half, K, scale=0.03), torch.randn(2 * half, generator=g)
        perm = torch.tensor([(r // 160) * 80 + r % 160 if r % 160 < 80 else half + (r // 160) * 80 + (r % 160 - 80) for r in range(2 * half)])
        yg = torch.full((M, half), float("nan"), dtype=torch.bfloat16, device="cuda")
        hip.gemm(xg.cuda(), Wg[perm].contiguous().cuda(), yg, bias=bg[perm].contiguous().cuda(), act=hip.ACT_GEGLU, N=2 * half)
    finally:
        L.cl_gemm_force_config(-1)
    assert rel_l2(y.cpu().double(), want) < 4e-3
    full = xg.double() @ Wg.double().t() + bg.double()
    wantg = full[:, :half] * torch.nn.functional.gelu(full[:, half:])
    assert rel_l2(yg.cpu().double(), wantg) < 4e-3


@pytest.mark.parametrize("B,H,W,C,N,cfg,what", [
    (20, 64, 64, 320, 320, 47, "persistent halo: 640 tiles over 256 workgroups (ragged tile count), images handed over between tiles"),
    (34, 32, 32, 640, 320, 47, "persistent halo, 32x32 level"),
    (24, 64, 64, 128, 128, 48, "persistent halo, 128-column tiles"),
    (20, 64, 64, 64, 160, 47, "persistent halo, one chunk per tile"),
    (5, 72, 56, 128, 320, 47, "persistent per-tap DMA (ragged geometry: tiles across images -> the row-bias is read from memory)"),
])
def test_w4_persistent_conv_rowbias_residual_vs_fp64(B, H, W, C, N, cfg, what):
    """The persistent form (one workgroup per CU walks its tiles; configurations 47 / 48) with the whole ResBlock epilogue:
    bias + time-embedding row-bias (openaimodel.py:272) + SiLU, then alpha * (.) + beta * skip (openaimodel.py:274).  Bias / row-bias
    reach the epilogue through LDS (one DMA per tile, double-buffered by tile parity), the residual by look-ahead loads."""
    hip, L = _need_gpu()
    g = torch.Generator().manual_seed(B * 1000 + H + C + N + cfg)
    M = B * H * W
    x, w = _mk(g, M, C), _mk(g, N, 9 * C, scale=0.03)
    bias, emb, res = torch.randn(N, generator=g), _mk(g, B, N), _mk(g, M, N)
    want = torch.nn.functional.conv2d(x.double().view(B, H, W, C).permute(0, 3, 1, 2), w.double().view(N, 3, 3, C).permute(0, 3, 1, 2),
                                      bias.double(), padding=1) + emb.double()[:, :, None, None]
    want = torch.nn.functional.silu(want).permute(0, 2, 3, 1).reshape(M, N)
    want = 0.75 * want + 1.5 * res.double()
    outs = []
    try:
        for c in (cfg, cfg, 16 if cfg == 47 else 17):
            L.cl_gemm_force_config(c)
            y = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
            hip.gemm(x.cuda(), w.cuda(), y, bias=bias.cuda(), rowbias=emb.cuda(), rows_per_batch=H * W, act=hip.ACT_SILU,
                     residual=res.cuda(), alpha=0.75, beta=1.5, mode=hip.CONV_S1, conv=(B, H, W, H, W), k1=C, out_f32=True,
                     dtype=hip.BF16)
            outs.append(y.cpu())
    finally:
        L.cl_gemm_force_config(-1)
    assert rel_l2(outs[0].double(), want) < 2e-5, what
    assert torch.equal(outs[0], outs[1]), what                 # bitwise repeatable
    assert rel_l2(outs[0].double(), outs[2].double()) < 2e-6   # the ping-pong tile kernel on the same call


def test_w4_persistent_linear_and_geglu_vs_fp64():
    hip, L = _need_gpu()
    g = torch.Generator().manual_seed(4748)
    M, N, K, r = 70000, 320, 1280, 128                    # 274 x 2 tiles: two to three per workgroup, a ragged last row tile
    x, W, t, Bup = _mk(g, M, K), _mk(g, N, K, scale=0.03), _mk(g, M, r), _mk(g, N, r, scale=0.05)
    bias, res = torch.randn(N, generator=g), _mk(g, M, N)
    want = 0.5 * (x.double() @ W.double().t() + bias.double() + t.double() @ Bup.double().t()) + 2.0 * res.double()
    half, Kg = 1280, 320
    xg, Wg, bg = _mk(g, 33000, Kg), _mk(g, 2 * half, Kg, scale=0.05), torch.randn(2 * half, generator=g)
    perm = torch.tensor([(q // 160) * 80 + q % 160 if q % 160 < 80 else half + (q // 160) * 80 + (q % 160 - 80) for q in range(2 * half)])
    try:
        L.cl_gemm_force_config(47)
        y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        hip.gemm(x.cuda(), W.cuda(), y, a2=t.cuda(), w2=Bup.cuda(), bias=bias.cuda(), residual=res.cuda(), alpha=0.5, beta=2.0)
        yg = torch.full((33000, half), float("nan"), dtype=torch.bfloat16, device="cuda")
        hip.gemm(xg.cuda(), Wg[perm].contiguous().cuda(), yg, bias=bg[perm].contiguous().cuda(), act=hip.ACT_GEGLU, N=2 * half)
    finally:
        L.cl_gemm_force_config(-1)
    assert rel_l2(y.cpu().double(), want) < 2.5e-3
    full = xg.double() @ Wg.double().t() + bg.double()
    assert rel_l2(yg.cpu().double(), full[:, :half] * torch.nn.functional.gelu(full[:, half:])) < 4e-3
