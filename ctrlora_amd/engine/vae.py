"""AutoencoderKL encoder / decoder (the frozen SD first stage) on the HIP kernels -- SURVEY.md row 8(f1).

Where the reference calls it on the hot path: the condition image is VAE-encoded inside every apply_model
(cldm/cldm_ctrlora_finetune.py:76-77) and the target image once per step (ldm/models/diffusion/ddpm.py:773) --
1 117 GFLOP per image, more than half of the end-to-end training FLOPs; sampling ends with one decode
(ddpm.py:820-828).  Modules restated (behaviour, not code; ldm/modules/diffusionmodules/model.py):

  ResnetBlock :97-149 (temb_channels = 0: GroupNorm(32, eps 1e-6) -> swish -> conv3x3, twice; nin_shortcut 1x1)
  AttnBlock   :152-202 (single head over all C = 512 channels, softmax(q k^T C^-0.5) v, proj_out, residual)
  Downsample  :80-84  (pad (0,1,0,1) then 3x3 stride 2: conv mode CONV_S2A)      Upsample :61-69 (nearest x2 + 3x3)
  Encoder     :452-543, Decoder :546-650; AutoencoderKL.encode/decode (ldm/models/autoencoder.py:70-84) incl. the
  1x1 quant_conv (folded into conv_out: a 1x1 after a 3x3 composes exactly) and post_quant_conv.

Forward only (the first stage is frozen: ddpm.py:615-636), activations token-major [B*H*W, C] in the engine dtype,
fp32 GroupNorm statistics and fp32 attention scores / softmax.  The 4096 x 4096 x 512 single-head attention runs as
two MFMA GEMMs around a row-softmax kernel per image (34 GFLOP per image against 1 083 in the convolutions).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .. import hip
from .blocks import Ctx, GroupNormOp, conv3_fwd, linear_fwd
from .packing import Conv3W, LinearW, NormW, rup


class _VB:
    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, dtype, device):
        self.sd, self.prefix, self.dtype, self.device = sd, prefix, dtype, device

    def g(self, name):
        return self.sd[self.prefix + name]

    def has(self, name):
        return (self.prefix + name) in self.sd

    def conv3(self, name):
        return Conv3W(self.g(name + ".weight"), self.g(name + ".bias"), self.dtype, self.device, need_bwd=False)

    def lin(self, name):
        return LinearW(self.g(name + ".weight"), self.g(name + ".bias"), self.dtype, self.device, need_bwd=False)

    def norm(self, name):
        return GroupNormOp(NormW(self.g(name + ".weight"), self.g(name + ".bias"), self.device), 1e-6, True)


class _VRes:
    def __init__(self, b: _VB, p: str):
        self.n1, self.c1 = b.norm(p + ".norm1"), b.conv3(p + ".conv1")
        self.n2, self.c2 = b.norm(p + ".norm2"), b.conv3(p + ".conv2")
        self.nin = b.lin(p + ".nin_shortcut") if b.has(p + ".nin_shortcut.weight") else None

    def fwd(self, ctx: Ctx, x, B, H, W):
        h, _ = self.n1.fwd(ctx, x, B, H * W)
        h = conv3_fwd(ctx, self.c1, h, B, H, W)
        h2, _ = self.n2.fwd(ctx, h, B, H * W)
        del h
        if self.nin is not None:
            sk, _ = linear_fwd(ctx, self.nin, x)
            return conv3_fwd(ctx, self.c2, h2, B, H, W, out=sk, residual=sk)
        return conv3_fwd(ctx, self.c2, h2, B, H, W, residual=x)


class _VAttn:
    def __init__(self, b: _VB, p: str):
        nw = NormW(b.g(p + ".norm.weight"), b.g(p + ".norm.bias"), b.device)
        self.norm = GroupNormOp(nw, 1e-6, False)
        C = b.g(p + ".q.weight").shape[0]
        W = torch.cat([b.g(f"{p}.{n}.weight").reshape(C, C) for n in ("q", "k", "v")], 0)
        bias = torch.cat([b.g(f"{p}.{n}.bias") for n in ("q", "k", "v")], 0)
        self.qkv = LinearW(W, bias, b.dtype, b.device, need_bwd=False)
        self.proj = b.lin(p + ".proj_out")
        self.C = C

    def fwd(self, ctx: Ctx, x, B, H, W):
        N, C = H * W, self.C
        hn, _ = self.norm.fwd(ctx, x, B, N)
        qkv, _ = linear_fwd(ctx, self.qkv, hn)
        del hn
        a = ctx.new(B * N, C)
        npad = rup(N, 32)
        S = torch.empty((N, N), dtype=torch.float32, device=ctx.device)
        P = ctx.new(N, N)
        vt = torch.empty((1, C, npad), dtype=ctx.dtype, device=ctx.device)
        for bi in range(B):                      # single head over all channels: two GEMMs around a row softmax
            rows = slice(bi * N, (bi + 1) * N)
            q, k, v = qkv[rows, :C], qkv[rows, C:2 * C], qkv[rows, 2 * C:]
            hip.gemm(q, k, S, out_f32=True)                                  # S = q k^T (fp32)
            hip.softmax_rows(S, P, float(C) ** -0.5)
            hip.transpose(v, vt, 1, N, C, npad, ldi=v.stride(0))
            hip.gemm(P, vt[0][:, :N] if npad == N else vt[0], a[rows], k1=N)  # a = P v
        out, _ = linear_fwd(ctx, self.proj, a, residual=x)
        return out


class VAEEncoderE:
    """AutoencoderKL.encode up to the posterior moments: x (B,3,H,W) fp32 in [-1,1] -> (B, 2*embed, H/8, W/8) fp32."""

    def __init__(self, sd, ddconfig: dict, dtype, device, prefix: str = ""):
        hip.lib()
        self.dtype, self.device = dtype, torch.device(device)
        b = _VB(sd, prefix + "encoder.", dtype, self.device)
        ch_mult, nrb = tuple(ddconfig["ch_mult"]), ddconfig["num_res_blocks"]
        assert not ddconfig.get("attn_resolutions"), "the SD first stage has attention in the middle block only"
        self.conv_in = b.conv3("conv_in")
        self.levels: List[tuple] = []
        for i in range(len(ch_mult)):
            blocks = [_VRes(b, f"down.{i}.block.{j}") for j in range(nrb)]
            down = b.conv3(f"down.{i}.downsample.conv") if i != len(ch_mult) - 1 else None
            self.levels.append((blocks, down))
        self.mid1, self.attn, self.mid2 = _VRes(b, "mid.block_1"), _VAttn(b, "mid.attn_1"), _VRes(b, "mid.block_2")
        self.norm_out = b.norm("norm_out")
        # conv_out (3x3) followed by quant_conv (1x1): one 3x3 conv with composed weights (exact: the 1x1 is pointwise)
        Wc, bc = b.g("conv_out.weight").float(), b.g("conv_out.bias").float()
        Q = sd[prefix + "quant_conv.weight"].float().reshape(sd[prefix + "quant_conv.weight"].shape[0], -1)
        qb = sd[prefix + "quant_conv.bias"].float()
        Wf = torch.einsum("po,oikl->pikl", Q, Wc)
        self.conv_out = Conv3W(Wf, Q @ bc + qb, dtype, self.device, need_bwd=False)
        self.out_ch = Q.shape[0]

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        B, Cin, H, W = x.shape
        assert H % 8 == 0 and W % 8 == 0
        ctx = Ctx(self.dtype, self.device, False)
        tok = torch.empty((B * H * W, rup(Cin, 32)), dtype=self.dtype, device=self.device)
        hip.nchw_to_tok(x.float().contiguous(), tok)
        h = conv3_fwd(ctx, self.conv_in, tok, B, H, W)
        del tok
        for blocks, down in self.levels:
            for blk in blocks:
                h = blk.fwd(ctx, h, B, H, W)
            if down is not None:
                h = conv3_fwd(ctx, down, h, B, H, W, mode=hip.CONV_S2A)
                H, W = H // 2, W // 2
        h = self.mid2.fwd(ctx, self.attn.fwd(ctx, self.mid1.fwd(ctx, h, B, H, W), B, H, W), B, H, W)
        hn, _ = self.norm_out.fwd(ctx, h, B, H * W)
        mom_tok = conv3_fwd(ctx, self.conv_out, hn, B, H, W, out_f32=True)
        out = torch.empty((B, self.out_ch, H, W), dtype=torch.float32, device=self.device)
        return hip.tok_to_nchw(mom_tok, out)


class VAEDecoderE:
    """AutoencoderKL.decode: z (B, embed, h, w) fp32 (already divided by scale_factor) -> image (B, 3, 8h, 8w) fp32."""

    def __init__(self, sd, ddconfig: dict, dtype, device, prefix: str = ""):
        hip.lib()
        self.dtype, self.device = dtype, torch.device(device)
        b = _VB(sd, prefix + "decoder.", dtype, self.device)
        ch_mult, nrb = tuple(ddconfig["ch_mult"]), ddconfig["num_res_blocks"]
        # post_quant_conv (1x1, embed -> z_channels) as a padded linear: a 1x1 BEFORE a zero-padded 3x3 does not fold
        pw = sd[prefix + "post_quant_conv.weight"].float()
        zc, emb = pw.shape[0], pw.shape[1]
        Wp = torch.zeros(32, 32); Wp[:zc, :emb] = pw.reshape(zc, emb)
        bp = torch.zeros(32); bp[:zc] = sd[prefix + "post_quant_conv.bias"].float()
        self.post_quant = LinearW(Wp, bp, dtype, self.device, need_bwd=False)
        self.conv_in = b.conv3("conv_in")
        self.mid1, self.attn, self.mid2 = _VRes(b, "mid.block_1"), _VAttn(b, "mid.attn_1"), _VRes(b, "mid.block_2")
        self.levels: List[tuple] = []
        for i in reversed(range(len(ch_mult))):
            blocks = [_VRes(b, f"up.{i}.block.{j}") for j in range(nrb + 1)]
            up = b.conv3(f"up.{i}.upsample.conv") if i != 0 else None
            self.levels.append((blocks, up))
        self.norm_out = b.norm("norm_out")
        self.conv_out = b.conv3("conv_out")
        self.out_ch = b.g("conv_out.weight").shape[0]

    @torch.no_grad()
    def __call__(self, z: torch.Tensor) -> torch.Tensor:
        B, C, H, W = z.shape
        ctx = Ctx(self.dtype, self.device, False)
        tok = torch.empty((B * H * W, 32), dtype=self.dtype, device=self.device)
        hip.nchw_to_tok(z.float().contiguous(), tok)
        zq, _ = linear_fwd(ctx, self.post_quant, tok)
        h = conv3_fwd(ctx, self.conv_in, zq, B, H, W)
        h = self.mid2.fwd(ctx, self.attn.fwd(ctx, self.mid1.fwd(ctx, h, B, H, W), B, H, W), B, H, W)
        for blocks, up in self.levels:
            for blk in blocks:
                h = blk.fwd(ctx, h, B, H, W)
            if up is not None:
                h = conv3_fwd(ctx, up, h, B, H, W, mode=hip.CONV_UP2)
                H, W = 2 * H, 2 * W
        hn, _ = self.norm_out.fwd(ctx, h, B, H * W)
        del h
        img_tok = conv3_fwd(ctx, self.conv_out, hn, B, H, W, out_f32=True)
        out = torch.empty((B, self.out_ch, H, W), dtype=torch.float32, device=self.device)
        return hip.tok_to_nchw(img_tok, out)
