"""GPU parity of the x-stationary streaming product (csrc/gemm_xs.hip, launch configuration 34) through the C ABI (cl_gemm):
the wide-N / short-K linears of FeedForward / CrossAttention (ldm/modules/attention.py:49-76,163-171) with their rank-r LoRA as
the second K segment (cldm/lora.py:285-291).  Reference: the same product in fp64 on the CPU from the bf16-rounded operands;
tolerance 2.5e-3 rel-L2 = one bf16 rounding of the output (measured 1.6e-3 .. 1.7e-3)."""
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


def _ref(x, W, t=None, Bup=None, bias=None, groups=1):
    y = x.double() @ W.double().t()
    if t is not None:
        N, r = W.shape[0], Bup.shape[1]
        for gi in range(groups):
            cols = slice(gi * N // groups, (gi + 1) * N // groups)
            y[:, cols] += t.double()[:, gi * r:(gi + 1) * r] @ Bup.double()[cols].t()
    if bias is not None:
        y += bias.double()
    return y


@pytest.mark.parametrize("M,N,K,r,groups,alpha,alpha_n,nsplit", [
    (1000, 960, 320, 128, 3, 0.31, 320, 0),      # grouped q | k | v with the pre-scaled-Q alpha on the q columns
    (4096, 2560, 320, 128, 1, 1.0, 0, 2),        # GEGLU projection (training form: full width)
    (515, 1280, 320, 0, 1, 1.0, 0, 0),           # ragged M: rows past M repeat the last row
    (2048, 5120, 640, 128, 1, 1.0, 0, 4),
    (700, 1920, 640, 0, 1, 0.5, 640, 1),
    (256, 3200, 320, 0, 1, 1.0, 0, 1),           # a run longer than the bias image: the launcher splits it
])
def test_xs_plain_grouped_alpha_vs_fp64_and_tile_kernels(M, N, K, r, groups, alpha, alpha_n, nsplit):
    hip, L = _need_gpu()
    g = torch.Generator().manual_seed(M + N + K)
    x, W = _mk(g, M, K), _mk(g, N, K, scale=0.05)
    t = _mk(g, M, r * groups) if r else None
    Bup = _mk(g, N, r, scale=0.05) if r else None
    bias = torch.randn(N, generator=g)
    want = _ref(x, W, t, Bup, bias, groups)
    want[:, :alpha_n or N] *= alpha
    cu = lambda v: None if v is None else v.cuda()
    kw = dict(a2=cu(t), w2=cu(Bup), bias=cu(bias), alpha=alpha, alpha_n=alpha_n, a2_group_n=N // groups if groups > 1 else 0)
    outs = []
    try:
        for cfg, sk in ((34, nsplit), (34, nsplit), (-1, 0)):
            L.cl_gemm_force_config(cfg); L.cl_gemm_force_splitk(sk)
            y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            hip.gemm(cu(x), cu(W), y, **kw)
            outs.append(y.cpu())
    finally:
        L.cl_gemm_force_config(-1); L.cl_gemm_force_splitk(0)
    assert rel_l2(outs[0].double(), want) < 2.5e-3
    assert torch.equal(outs[0], outs[1])                       # bitwise repeatable
    assert rel_l2(outs[0].double(), outs[2].double()) < 2.5e-3  # and the tile kernels agree


@pytest.mark.parametrize("M,N,K,r,beta", [(1000, 320, 320, 128, 1.0), (4100, 320, 320, 0, 1.0), (2048, 640, 640, 128, -0.5),
                                           (300, 1280, 640, 0, 2.0)])
def test_xs_residual_epilogue_vs_fp64(M, N, K, r, beta):
    hip, L = _need_gpu()
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x, W = _mk(g, M, K), _mk(g, N, K, scale=0.05)
    t = _mk(g, M, r) if r else None
    Bup = _mk(g, N, r, scale=0.05) if r else None
    bias, res = torch.randn(N, generator=g), _mk(g, M, N)
    want = _ref(x, W, t, Bup, bias) + beta * res.double()
    cu = lambda v: None if v is None else v.cuda()
    try:
        L.cl_gemm_force_config(34)
        outs = []
        for _ in range(2):
            y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            hip.gemm(cu(x), cu(W), y, a2=cu(t), w2=cu(Bup), bias=cu(bias), residual=cu(res), beta=beta)
            outs.append(y.cpu())
        # in place (out IS the residual), as the transformer block calls it
        y2 = cu(res).clone()
        hip.gemm(cu(x), cu(W), y2, a2=cu(t), w2=cu(Bup), bias=cu(bias), residual=y2, beta=beta)
    finally:
        L.cl_gemm_force_config(-1)
    assert rel_l2(outs[0].double(), want) < 2.5e-3
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], y2.cpu())


@pytest.mark.parametrize("M,half,K,r", [(1000, 1280, 320, 128), (4096, 1280, 320, 0), (515, 2560, 640, 0), (384, 96, 640, 128)])
def test_xs_fused_geglu_natural_row_order_vs_fp64(M, half, K, r):
    """act = ACT_GEGLU_SPLIT: W rows [value | gate] as nn.Linear(dim, 2 * inner) holds them (attention.py:52-56)."""
    hip, L = _need_gpu()
    g = torch.Generator().manual_seed(M + half + K)
    N = 2 * half
    x, W = _mk(g, M, K), _mk(g, N, K, scale=0.05)
    t = _mk(g, M, r) if r else None
    Bup = _mk(g, N, r, scale=0.05) if r else None
    bias = torch.randn(N, generator=g)
    h = _ref(x, W, t, Bup, bias)
    want = h[:, :half] * torch.nn.functional.gelu(h[:, half:])
    cu = lambda v: None if v is None else v.cuda()
    outs = []
    for _ in range(2):
        y = torch.full((M, half), float("nan"), dtype=torch.bfloat16, device="cuda")
        hip.gemm(cu(x), cu(W), y, a2=cu(t), w2=cu(Bup), bias=cu(bias), act=hip.ACT_GEGLU_SPLIT, N=N)
        outs.append(y.cpu())
    assert rel_l2(outs[0].double(), want) < 2.5e-3
    assert torch.equal(outs[0], outs[1])
    # what the kernel does not cover is refused, not approximated: fp32 storage
    with pytest.raises(hip.HipError):
        hip.gemm(cu(x).float(), cu(W).float(), torch.empty(M, half, device="cuda"), bias=cu(bias), act=hip.ACT_GEGLU_SPLIT, N=N)


def test_xs_configuration_falls_back_to_the_tile_kernels_for_what_it_does_not_cover():
    """A table entry naming configuration 34 must cost speed, never correctness: rowbias + SiLU (ResBlock emb path) and a
    K the kernel has no instance for run through the built-in rules."""
    hip, L = _need_gpu()
    g = torch.Generator().manual_seed(5)
    M, N, K = 512, 320, 320
    x, W, rb = _mk(g, M, K), _mk(g, N, K, scale=0.05), _mk(g, 4, N)
    want = torch.nn.functional.silu(x.double() @ W.double().t() + rb.double().repeat_interleave(128, 0))
    x2, W2 = _mk(g, M, 448), _mk(g, N, 448, scale=0.05)
    try:
        L.cl_gemm_force_config(34)
        y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        hip.gemm(x.cuda(), W.cuda(), y, rowbias=rb.cuda(), rows_per_batch=128, act=hip.ACT_SILU)
        y2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        hip.gemm(x2.cuda(), W2.cuda(), y2)
    finally:
        L.cl_gemm_force_config(-1)
    assert rel_l2(y.cpu().double(), want) < 2.5e-3
    assert rel_l2(y2.cpu().double(), x2.double() @ W2.double().t()) < 2.5e-3


@pytest.mark.parametrize("M,N,K,geglu", [(1000, 960, 320, False), (4100, 320, 320, False), (2048, 1920, 640, False),
                                         (1000, 2560, 320, True), (515, 5120, 640, True)])
def test_xs_layernorm_prologue_vs_fp64_and_vs_the_layernorm_kernel(M, N, K, geglu):
    """hip.gemm(ln=...): LayerNorm(x) . W^T in one launch (BasicTransformerBlock norm1/2/3 -> the product that reads them,
    ldm/modules/attention.py:271-275).  Against fp64 (LayerNorm in double, rounded to bf16 where the stand-alone kernel
    rounds, then the product), against cl_layernorm_fwd + the same product, and the (mean, rstd) it leaves for the backward."""
    hip, L = _need_gpu()
    g = torch.Generator().manual_seed(M + N + K + 7)
    x = ((torch.randn(M, K, generator=g) * 1.7 + 0.6)).to(torch.bfloat16)
    W = _mk(g, N, K, scale=0.05)
    gamma, beta, bias = 1 + 0.3 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g), torch.randn(N, generator=g)
    xn = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5).to(torch.bfloat16)
    h = xn.double() @ W.double().t() + bias.double()
    want = h[:, :N // 2] * torch.nn.functional.gelu(h[:, N // 2:]) if geglu else h
    NO = N // 2 if geglu else N
    act = hip.ACT_GEGLU_SPLIT if geglu else hip.ACT_NONE
    xd, Wd, gd, bd, biasd = x.cuda(), W.cuda(), gamma.cuda(), beta.cuda(), bias.cuda()
    stats = torch.full((M, 2), float("nan"), device="cuda")
    outs = []
    for _ in range(2):
        y = torch.full((M, NO), float("nan"), dtype=torch.bfloat16, device="cuda")
        hip.gemm(xd, Wd, y, bias=biasd, act=act, N=N, ln=(gd, bd, 1e-5, stats))
        outs.append(y.cpu())
    assert rel_l2(outs[0].double(), want) < 2.5e-3
    assert torch.equal(outs[0], outs[1])
    # the two-launch form it replaces
    n2, st2 = torch.empty_like(xd), torch.empty(M, 2, device="cuda")
    hip.layernorm_fwd(xd, n2, gd, bd, 1e-5, st2)
    y2 = torch.empty(M, NO, dtype=torch.bfloat16, device="cuda")
    hip.gemm(n2, Wd, y2, bias=biasd, act=act, N=N)
    assert rel_l2(outs[0].double(), y2.cpu().double()) < 1.5e-3       # (a different summation order flips a few bf16 roundings)
    assert torch.allclose(stats.cpu(), st2.cpu(), rtol=2e-6, atol=1e-7)
    mu = x.double().mean(1)
    assert torch.allclose(stats[:, 0].cpu().double(), mu, rtol=1e-5, atol=1e-6)
    # refused where the kernel has no instance: a second K segment, fp32
    with pytest.raises(hip.HipError):
        hip.gemm(xd, Wd, torch.empty(M, NO, dtype=torch.bfloat16, device="cuda"), a2=xd[:, :128].contiguous(),
                 w2=Wd[:, :128].contiguous(), bias=biasd, act=act, N=N, ln=(gd, bd, 1e-5, None))
