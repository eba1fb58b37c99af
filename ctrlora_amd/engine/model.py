"""apply_model-level executor: ControlNet(+LoRA) -> residual injection -> frozen SD UNet -> eps,
and the matching hand-written backward.  This is what the cldm.* mirror classes call.

Reference seam restated: Control{Finetune,Pretrain,Inference}LDM.apply_model
(cldm/cldm_ctrlora_finetune.py:67-82, cldm_ctrlora_pretrain.py:95-111, cldm_ctrlora_inference.py:156-178)
after the hint has been encoded to a 4-channel latent.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Sequence

import torch

from .. import hip
from .blocks import Ctx
from .nets import ControlNetE, NetCfg, UNetE
from .packing import rup


class CtrLoRAEngine:
    # ControlNet trunk || UNet encoder on two HIP streams (forward); CTRLORA_OVERLAP_STREAMS=0: one stream (A/B switch)
    overlap_streams = os.environ.get("CTRLORA_OVERLAP_STREAMS", "1") != "0"
    overlap_wgrad = os.environ.get("CTRLORA_OVERLAP_WGRAD", "1") != "0"   # weight gradients of stage i || data gradients of stage i+1

    def __init__(self, sd_unet: Dict[str, torch.Tensor], sd_controls: Sequence[Dict[str, torch.Tensor]], cfg: NetCfg,
                 dtype: torch.dtype = torch.bfloat16, device="cuda", need_bwd: bool = True,
                 unet_prefix: str = "", control_prefix: str = ""):
        hip.lib()   # fail loudly, before anything else, if the HIP library is missing
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.unet = UNetE(sd_unet, cfg, dtype, self.device, prefix=unet_prefix, need_bwd=need_bwd)
        self.controls: List[ControlNetE] = [
            ControlNetE(sd, cfg, dtype, self.device, prefix=control_prefix, need_bwd=need_bwd) for sd in sd_controls]
        self._rec = None
        self.cache_context_kv = False
        self._kv: Optional[dict] = None
        self._side = None

    @classmethod
    def from_executors(cls, unet: UNetE, controls: Sequence[ControlNetE]) -> "CtrLoRAEngine":
        """Compose already-built executors (the cldm.* modules build and own theirs lazily)."""
        hip.lib()
        self = cls.__new__(cls)
        self.cfg, self.dtype, self.device = unet.cfg, unet.dtype, unet.device
        self.unet, self.controls = unet, list(controls)
        self._rec, self.cache_context_kv, self._kv = None, False, None
        self._side = None
        return self

    @torch.no_grad()
    def forward_external_control(self, x_noisy, t, context, control: Optional[list], only_mid_control=False):
        """ControlledUnetModel.forward with caller-supplied NCHW residuals (cldm/cldm.py:22-45); like the
        reference it consumes `control` with pop()."""
        B, _, H, W = x_noisy.shape
        ctx = Ctx(self.dtype, self.device, False)
        t = t.to(device=self.device, dtype=torch.long).contiguous()
        c = self._ctx_in(context)
        semb, hs, dims, h_mid = self.unet.encode(ctx, self._tok_in(x_noisy), t, c, B, H, W)
        bufs = self.unet.alloc_decoder_inputs(ctx, B, dims)
        self.unet.fill_without_control(ctx, bufs, hs, h_mid)
        if control is not None:
            sinks = self.unet.control_sinks(bufs, hs, h_mid)
            mid = control.pop()
            hip.axpby(self._tok_in(mid)[:, :sinks[-1][0].shape[1]], sinks[-1][0], 1.0, 1.0)
            for k in range(len(sinks) - 2, -1, -1):
                if only_mid_control:
                    break
                ck = control.pop()
                hip.axpby(self._tok_in(ck)[:, :sinks[k][0].shape[1]], sinks[k][0], 1.0, 1.0)
        eps_tok, _ = self.unet.decode(ctx, bufs, semb, c, B, dims[-1])
        eps = torch.empty((B, self.cfg.out_channels, H, W), dtype=torch.float32, device=self.device)
        return hip.tok_to_nchw(eps_tok, eps)

    # ---------------------------------------------------------------- boundary conversions
    def _tok_in(self, x_nchw: torch.Tensor) -> torch.Tensor:
        B, C, H, W = x_nchw.shape
        out = torch.empty((B * H * W, rup(C, 32)), dtype=self.dtype, device=self.device)
        return hip.nchw_to_tok(x_nchw.float(), out)

    def _ctx_in(self, c: torch.Tensor) -> torch.Tensor:
        B, L, D = c.shape
        out = torch.empty((B * L, D), dtype=self.dtype, device=self.device)
        return hip.pack2d(c.float().reshape(B * L, D), out)

    def reset_context_cache(self):
        self._kv = None

    def detach_context_cache(self):
        """Hand the cached context K / V products to the caller (a kept hipGraph reads them by address) and forget them."""
        kv, self._kv = self._kv, None
        return kv

    # ---------------------------------------------------------------- forward
    @torch.no_grad()
    def forward(self, x_noisy, t, context, hints: Optional[Sequence[torch.Tensor]], control_scales=None,
                lora_weights=None, record: bool = False, only_mid_control: bool = False) -> torch.Tensor:
        """x_noisy (B,4,H,W) fp32, t (B,) int64, context (B,L,D), hints: one latent (B,4,H,W) per
        ControlNet bank or None (plain UNet).  Returns eps (B,out_channels,H,W) fp32."""
        B, _, H, W = x_noisy.shape
        ctx = Ctx(self.dtype, self.device, record)
        t = t.to(device=self.device, dtype=torch.long).contiguous()
        c = self._ctx_in(context)
        kvs = None
        if self.cache_context_kv and not record:
            if self._kv is None:
                self._kv = {"unet": {}, "cn": [dict() for _ in self.controls]}
            kvs = self._kv
        # The ControlNet trunk (everything but its zero convs) does not depend on the UNet encoder: run it on a
        # second stream so that the under-filled 32x32 / 16x16 / 8x8 levels of the two networks overlap.
        # Separate Ctx (GroupNorm scratch) and per-stream split-K workspace keep the streams from sharing state.
        overlap = self.overlap_streams and hints is not None and not only_mid_control
        trunks = []
        if overlap:
            assert len(hints) == len(self.controls)
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = hip.side_stream(self.device)
            hint_toks = [self._tok_in(h) for h in hints]
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                ctx_cn = Ctx(self.dtype, self.device, record)
                for i, (cn, ht) in enumerate(zip(self.controls, hint_toks)):
                    trunks.append(cn.fwd_trunk(ctx_cn, ht, t, c, B, H, W, kv=None if kvs is None else kvs["cn"][i]))
        semb, hs, dims, h_mid = self.unet.encode(ctx, self._tok_in(x_noisy), t, c, B, H, W,
                                                 kv=None if kvs is None else kvs["unet"])
        if overlap:
            torch.cuda.current_stream().wait_stream(self._side)
        bufs = self.unet.alloc_decoder_inputs(ctx, B, dims)
        cn_recs = []
        scales = list(control_scales) if control_scales is not None else [1.0] * (len(dims) + 1)
        if hints is None:
            self.unet.fill_without_control(ctx, bufs, hs, h_mid)
        elif overlap:
            weights = list(lora_weights) if lora_weights is not None else [1.0] * len(hints)
            sinks = self.unet.control_sinks(bufs, hs, h_mid)
            for i, (cn, (rec, cn_hs)) in enumerate(zip(self.controls, trunks)):
                if i > 0:   # accumulate the next LoRA's weighted residuals in place
                    sinks = [(o, o) for o, _ in sinks]
                cn.fwd_zero(cn_hs, sinks, scales, weights[i])
                cn_recs.append((rec, weights[i]))
            del trunks
        else:
            assert len(hints) == len(self.controls)
            weights = list(lora_weights) if lora_weights is not None else [1.0] * len(hints)
            sinks = self.unet.control_sinks(bufs, hs, h_mid)
            if only_mid_control:
                # reference: torch.cat([h, hs.pop()]) without the control term, mid residual still added
                self.unet.fill_without_control(ctx, bufs, hs, h_mid)
                sinks = [(torch.empty_like(o), None) for o, _ in sinks[:-1]] + [sinks[-1]]
            for i, (cn, hint) in enumerate(zip(self.controls, hints)):
                if i > 0:   # accumulate the next LoRA's weighted residuals in place
                    sinks = [(o, o) for o, _ in sinks]
                rec = cn.fwd(ctx, self._tok_in(hint), t, c, B, H, W, sinks, scales, weights[i],
                             kv=None if kvs is None else kvs["cn"][i])
                cn_recs.append((rec, weights[i]))
        del hs
        eps_tok, dec_rec = self.unet.decode(ctx, bufs, semb, c, B, dims[-1], kv=None if kvs is None else kvs["unet"])
        eps = torch.empty((B, self.cfg.out_channels, H, W), dtype=torch.float32, device=self.device)
        hip.tok_to_nchw(eps_tok, eps)
        if record:
            self._rec = (ctx, cn_recs, dec_rec, scales, B, H, W)
        return eps

    @torch.no_grad()
    def control_outputs(self, hint, t, context, bank: int = 0) -> List[torch.Tensor]:
        """ControlNet*.forward as a stand-alone module: the 13 residuals in NCHW fp32."""
        B, _, H, W = hint.shape
        ctx = Ctx(self.dtype, self.device, False)
        cn = self.controls[bank]
        t = t.to(device=self.device, dtype=torch.long).contiguous()
        c = self._ctx_in(context)
        # output grids: follow the encoder's down-sampling
        dims, hh, ww = [], H, W
        for layers in cn.blocks:
            for l in layers:
                if getattr(l, "mode", None) == hip.CONV_S2:
                    hh, ww = hh // 2, ww // 2
            dims.append((hh, ww))
        dims.append((hh, ww))
        chans = cn.chans + [cn.chans[-1]]
        sinks = [(ctx.new(B * h_ * w_, ch), None) for (h_, w_), ch in zip(dims, chans)]
        cn.fwd(ctx, self._tok_in(hint), t, c, B, H, W, sinks, [1.0] * len(sinks), 1.0)
        outs = []
        for (o, _), (h_, w_), ch in zip(sinks, dims, chans):
            y = torch.empty((B, ch, h_, w_), dtype=torch.float32, device=self.device)
            outs.append(hip.tok_to_nchw(o, y))
        return outs

    # ---------------------------------------------------------------- backward
    @torch.no_grad()
    def backward(self, d_eps: torch.Tensor):
        """Accumulate d loss / d trainables into every bank's flat fp32 gradient buffer."""
        assert self._rec is not None, "forward(record=True) must precede backward()"
        ctx, cn_recs, dec_rec, scales, B, H, W = self._rec
        self._rec = None
        d_tok = torch.empty((B * H * W, 32), dtype=self.dtype, device=self.device)
        hip.nchw_to_tok(d_eps.float().contiguous(), d_tok)
        dbufs = self.unet.decode_bwd(ctx, d_tok, dec_rec, B)
        dsinks = self.unet.control_grad_sinks(dbufs)
        if self.overlap_wgrad and self.dtype == torch.bfloat16:
            if self._side is None:
                self._side = hip.side_stream(self.device)
            ctx.wstream = self._side        # idle during backward; has its own split scratch
        for cn, (rec, w) in zip(self.controls, cn_recs):
            cn.bwd(ctx, rec, dsinks, scales, w, B)
        ctx.wstream = None

    # ---------------------------------------------------------------- trainables
    def zero_grad(self):
        for cn in self.controls:
            hip.zero_(cn.tr.flat_grad)

    def repack(self):
        for cn in self.controls:
            cn.repack()
