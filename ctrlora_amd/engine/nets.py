"""ControlNet (CtrLoRA variant) and ControlledUnetModel executors on the HIP kernels.

Built from a state dict with the reference's parameter names, so the same files load:
  ControlNetFinetune   cldm/cldm_ctrlora_finetune.py:10-54  (over ControlNet.__init__, cldm/cldm.py:48-282)
  ControlledUnetModel  cldm/cldm.py:22-45                   (over UNetModel.__init__, openaimodel.py:412-736)

Dataflow differences from the reference that do not change the arithmetic:
  * activations are NHWC / token-major, so SpatialTransformer's two rearranges vanish and
    every 1x1 conv is a plain GEMM;
  * the UNet decoder's torch.cat([h, hs.pop() + control.pop()]) is never materialised by a
    copy: each zero conv writes  (conv(h)+b)*scale + skip  straight into the right half of the
    decoder block's input buffer, and the previous decoder block writes its output into the
    left half (row-stride addressing in every kernel);
  * the backward pass is hand-written: data gradients everywhere they are needed, weight
    gradients only for the optimizer's subset, nothing recomputed.
"""
from __future__ import annotations

import os

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import hip
from .blocks import (AttnE, Ctx, ResBlockE, SpatialTransformerE, base_bwd_weight, conv3_bwd_data, conv3_bwd_weight,
                     conv3_fwd, dense_bwd_weight, group_fwd, linear_bwd_data, linear_bwd_lora, linear_fwd)
from .packing import Conv3W, LinearW, LoraGroup, NormW, TrainableSet, rup


# Bumped whenever packed weights change (re-pack, frozen reload, bank switch, a replayed optimizer graph): captured graphs /
# caches that baked the old copies in (DDIMSampler.reuse_graph) compare it.
WEIGHTS_GENERATION = [0]


@dataclass(frozen=True)
class NetCfg:
    """The architecture knobs of configs/*.yaml (control_stage_config / unet_config params)."""
    in_channels: int = 4
    out_channels: int = 4
    model_channels: int = 320
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    num_heads: int = 8
    context_dim: int = 768

    @property
    def time_embed_dim(self) -> int:
        return 4 * self.model_channels

    @staticmethod
    def from_params(p: dict) -> "NetCfg":
        return NetCfg(in_channels=p.get("in_channels", 4), out_channels=p.get("out_channels", 4),
                      model_channels=p["model_channels"], channel_mult=tuple(p["channel_mult"]),
                      num_res_blocks=p["num_res_blocks"], attention_resolutions=tuple(p["attention_resolutions"]),
                      num_heads=p["num_heads"], context_dim=p["context_dim"])


# A/B switch: CTRLORA_GROUP_LORA=0 keeps one launch per LoRA linear
GROUP_LORA = os.environ.get("CTRLORA_GROUP_LORA", "1") != "0"
# A/B switch: CTRLORA_HOIST_EMB_BWD=0 keeps the emb_layers backward inside every ResBlock
HOIST_EMB_BWD = os.environ.get("CTRLORA_HOIST_EMB_BWD", "1") != "0"


def is_trainable_name(n: str) -> bool:
    """Name filter of ControlFinetuneLDM.configure_optimizers (cldm_ctrlora_finetune.py:92-101)."""
    return ("lora_layer" in n) or ("zero_convs" in n) or ("middle_block_out" in n) or ("norm" in n)


class _Builder:
    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, dtype, device, need_bwd: bool,
                 trainables: Optional[TrainableSet], lora_set: Optional[TrainableSet] = None, train_all: bool = False):
        self.sd, self.prefix, self.dtype, self.device = sd, prefix, dtype, device
        self.need_bwd, self.tr = need_bwd, trainables
        # train_all (Base-ControlNet pre-training, cldm_ctrlora_pretrain.py:174-182): every weight / bias / norm of
        # the network is declared in `trainables`; the LoRA factors go to `lora_set` (one set per task bank)
        self.tr_lora = lora_set if lora_set is not None else trainables
        self.train_all = train_all
        self.fold_lora = False       # set by ControlNetE before building: LoRA linears keep their fp32 base weight
        self.groups: List[LoraGroup] = []
        self.linears: List[LinearW] = []
        self.norms: List[NormW] = []
        self.convs: List[Conv3W] = []
        self.frozen: list = []        # (packed object, loader(sd)) for reload_frozen()

    def reload_frozen(self, sd: Dict[str, torch.Tensor]):
        """Re-pack every frozen tensor from `sd` (same key names) into the existing packed buffers."""
        self.sd = sd
        for fn in self.frozen:
            fn()
        WEIGHTS_GENERATION[0] += 1

    def _g(self, name):
        return self.sd[self.prefix + name]

    def _has(self, name):
        return (self.prefix + name) in self.sd

    def linear(self, name: str) -> LinearW:
        dn = name + ".lora_layer.down.weight"
        L = LinearW(self._g(name + ".weight"), self._g(name + ".bias") if self._has(name + ".bias") else None,
                    self.dtype, self.device, self.need_bwd,
                    keep_f32=self.fold_lora and self._has(dn) and self.dtype != torch.float32)
        if self.train_all:
            tW = self.tr.declare(self.prefix + name + ".weight", (L.N, L.K))
            tb = self.tr.declare(self.prefix + name + ".bias", (L.N,)) if self._has(name + ".bias") else None
            L.attach_trainable_weight(tW, tb)
        if self._has(dn):
            tA = self.tr_lora.declare(self.prefix + dn, self._g(dn).shape)
            un = name + ".lora_layer.up.weight"
            tB = self.tr_lora.declare(self.prefix + un, self._g(un).shape)
            L.attach_lora(tA, tB, self.device)
        self.linears.append(L)
        if not self.train_all:
            self.frozen.append(lambda: L.load(self._g(name + ".weight"),
                                              self._g(name + ".bias") if self._has(name + ".bias") else None))
        return L

    def fused(self, names: Sequence[str]) -> LinearW:
        W = torch.cat([self._g(n + ".weight") for n in names], dim=0)
        L = LinearW(W, None, self.dtype, self.device, self.need_bwd)
        self.frozen.append(lambda: L.load(torch.cat([self._g(n + ".weight") for n in names], dim=0), None))
        return L

    def zero_conv(self, name: str) -> LinearW:
        L = LinearW(self._g(name + ".weight"), self._g(name + ".bias"), self.dtype, self.device, self.need_bwd)
        if self.tr is not None:
            tW = self.tr.declare(self.prefix + name + ".weight", self._g(name + ".weight").shape)
            tb = self.tr.declare(self.prefix + name + ".bias", self._g(name + ".bias").shape)
            L.attach_trainable_weight(tW, tb)
        else:
            self.frozen.append(lambda: L.load(self._g(name + ".weight"), self._g(name + ".bias")))
        self.linears.append(L)
        return L

    def conv3(self, name: str) -> Conv3W:
        cw = Conv3W(self._g(name + ".weight"), self._g(name + ".bias"), self.dtype, self.device, self.need_bwd)
        if self.train_all:
            tW = self.tr.declare(self.prefix + name + ".weight", (cw.O, 9 * cw.Ip), conv=(cw.O, cw.I, cw.Ip))
            tb = self.tr.declare(self.prefix + name + ".bias", (cw.O,))
            cw.attach_trainable(tW, tb)
            self.convs.append(cw)
        else:
            self.frozen.append(lambda: cw.load(self._g(name + ".weight"), self._g(name + ".bias")))
        return cw

    def norm(self, name: str) -> NormW:
        w = NormW(self._g(name + ".weight"), self._g(name + ".bias"), self.device)
        if self.tr is not None and (self.train_all or is_trainable_name(name)):
            w.attach(self.tr.declare(self.prefix + name + ".weight", self._g(name + ".weight").shape),
                     self.tr.declare(self.prefix + name + ".bias", self._g(name + ".bias").shape))
        self.norms.append(w)
        self.frozen.append(lambda: w.load(self._g(name + ".weight"), self._g(name + ".bias")))
        return w

    def res(self, p: str, cin: int, cout: int) -> ResBlockE:
        skip = self.linear(p + ".skip_connection") if cin != cout else None
        return ResBlockE(self.norm(p + ".in_layers.0"), self.conv3(p + ".in_layers.2"),
                         self.linear(p + ".emb_layers.1"), self.norm(p + ".out_layers.0"),
                         self.conv3(p + ".out_layers.3"), skip)

    def st(self, p: str, ch: int, heads: int, lora: bool) -> SpatialTransformerE:
        tb = p + ".transformer_blocks.0"
        norm = self.norm(p + ".norm")
        proj_in = self.linear(p + ".proj_in")
        ln1 = self.norm(tb + ".norm1")
        a1 = [self.linear(f"{tb}.attn1.{n}") for n in ("to_q", "to_k", "to_v", "to_out.0")]
        ln2 = self.norm(tb + ".norm2")
        a2 = [self.linear(f"{tb}.attn2.{n}") for n in ("to_q", "to_k", "to_v", "to_out.0")]
        ln3 = self.norm(tb + ".norm3")
        ff_proj = self.linear(tb + ".ff.net.0.proj")
        ff_out = self.linear(tb + ".ff.net.2")
        proj_out = self.linear(p + ".proj_out")
        fq = fkv = None
        if not lora:   # frozen UNet: one GEMM for q|k|v and for the context's k|v
            fq = self.fused([f"{tb}.attn1.to_q", f"{tb}.attn1.to_k", f"{tb}.attn1.to_v"])
            fkv = self.fused([f"{tb}.attn2.to_k", f"{tb}.attn2.to_v"])
        g1 = g2 = None
        # (the grouped product's tile width -- 64, 128 or 160 -- has to divide the group width = the inner dimension;
        # any other width keeps one launch per linear, as emb_groups does)
        if (lora and GROUP_LORA and self.dtype != torch.float32 and all(L.r for L in a1[:3] + a2[1:3])
                and a1[0].N % 64 == 0):
            # q | k | v (and the context's k | v) share their input: grouped launches (packing.LoraGroup)
            g1, g2 = LoraGroup(a1[:3]), LoraGroup(a2[1:3])
            if self.fold_lora:
                g1.enable_merge(); g2.enable_merge()
            self.groups += [g1, g2]
        attn1 = AttnE(a1[0], a1[1], a1[2], a1[3], heads, True, fused_qkv=fq, group=g1)
        attn2 = AttnE(a2[0], a2[1], a2[2], a2[3], heads, False, fused_kv=fkv, need_kv_grad=lora, group=g2)
        return SpatialTransformerE(norm, proj_in, ln1, attn1, ln2, attn2, ln3, ff_proj, ff_out, proj_out)


# ------------------------------------------------------------------------------ layer wrappers

class _Env:
    """Mutable per-pass state threaded through the layers."""
    __slots__ = ("B", "H", "W", "semb", "c", "Nkv", "dsemb", "emb_grads", "kv", "emb_all", "kv_all", "emb_pre", "de_all")

    def __init__(self, B, H, W, semb, c, Nkv):
        self.B, self.H, self.W, self.semb, self.c, self.Nkv = B, H, W, semb, c, Nkv
        self.dsemb = None
        self.emb_grads = False
        self.kv = None
        self.emb_all = None      # frozen UNet: every ResBlock's emb_layers output, one product [B, sum cout]
        self.kv_all = None       # frozen UNet: every cross-attention's K / V of the context, one product
        self.emb_pre = None      # ControlNet: {id(_Res): (emb_layers output, x A^T)} formed by grouped launches up front
        self.de_all = None       # ControlNet backward: fp32 [B, sum cout] -- every grouped ResBlock's d emb_out lands in its slice


class _Res:
    def __init__(self, blk: ResBlockE):
        self.blk = blk
        self.cout = blk.cout
        self.emb_off = None      # column offset into _Env.emb_all (frozen UNet only)
        self.de_off = None       # column offset into _Env.de_all (ControlNet, member of an emb_layers group)

    def fwd(self, ctx, x, env, out=None):
        pre = None
        if env.emb_pre is not None:
            pre = env.emb_pre.get(id(self))
        elif env.emb_all is not None and self.emb_off is not None:
            pre = env.emb_all[:, self.emb_off:self.emb_off + self.cout]
        return self.blk.fwd(ctx, x, env.semb, env.B, env.H, env.W, out=out, e_pre=pre)

    def bwd(self, ctx, dy, saved, env, out=None):
        slot = None
        if env.de_all is not None and self.de_off is not None:      # hoisted emb_layers backward (ControlNetE.bwd)
            slot = env.de_all[:, self.de_off:self.de_off + self.cout]
        return self.blk.bwd(ctx, dy, saved, env.B, env.H, env.W, dsemb=env.dsemb, need_emb_grads=env.emb_grads,
                            out=out, de_slot=slot)


class _ST:
    def __init__(self, blk: SpatialTransformerE, key: str):
        self.blk, self.key = blk, key
        self.cout = blk.C
        self.kv_off = None       # column offset into _Env.kv_all (frozen UNet only)

    def fwd(self, ctx, x, env, out=None):
        cache = None
        if env.kv is not None:
            cache = env.kv.get(self.key)
            if cache is None:
                cache = self.blk.attn2.project_context(ctx, env.c)
                env.kv[self.key] = cache
        elif env.kv_all is not None and self.kv_off is not None:
            inner = self.blk.attn2.inner
            cache = (env.kv_all[:, self.kv_off:self.kv_off + inner],
                     env.kv_all[:, self.kv_off + inner:self.kv_off + 2 * inner], None, None)
        return self.blk.fwd(ctx, x, env.c, env.B, env.H, env.W, env.Nkv, out=out, kv_cache=cache)

    def bwd(self, ctx, dy, saved, env, out=None):
        return self.blk.bwd(ctx, dy, saved, env.B, env.H, env.W, env.Nkv, out=out)


class _Conv:
    """Plain 3x3 conv layer: input conv, Downsample.op (stride 2), Upsample.conv (nearest x2 fused)."""

    def __init__(self, cw: Conv3W, mode: int):
        self.cw, self.mode = cw, mode
        self.cout = cw.O
        # the source-grid forms of Upsample / Downsample (blocks._phase_ok) use weights derived from the packed ones: built here,
        # not on first use, so that a first call inside a hipGraph capture does not record the packing kernels into the graph
        # (a conv that is made trainable afterwards never uses them)
        from .blocks import CONV_PHASE
        if CONV_PHASE and cw.tW is None:
            if mode == hip.CONV_UP2:
                cw.phase_weights("up2")
                if cw.Wd is not None:
                    cw.phase_weights("up2d")
            elif mode == hip.CONV_S2 and cw.Wd is not None:
                cw.phase_weights("t2")

    def fwd(self, ctx, x, env, out=None):
        y = conv3_fwd(ctx, self.cw, x, env.B, env.H, env.W, mode=self.mode, out=out)
        saved = (x, env.H, env.W) if (ctx.record and self.cw.tW is not None) else ()
        if self.mode == hip.CONV_S2:
            env.H //= 2; env.W //= 2
        elif self.mode == hip.CONV_UP2:
            env.H *= 2; env.W *= 2
        return y, saved

    def bwd(self, ctx, dy, saved, env, out=None):
        # env.H/W are the spatial dims of dy (the conv's output grid)
        if saved:
            conv3_bwd_weight(ctx, self.cw, saved[0], dy, env.B, saved[1], saved[2], mode=self.mode)
        dx = conv3_bwd_data(ctx, self.cw, dy, env.B, env.H, env.W, fwd_mode=self.mode, out=out)
        if self.mode == hip.CONV_S2:
            env.H *= 2; env.W *= 2
        elif self.mode == hip.CONV_UP2:
            env.H //= 2; env.W //= 2
        return dx


def _run_fwd(ctx, layers, x, env, out=None):
    saved = []
    for i, l in enumerate(layers):
        x, s = l.fwd(ctx, x, env, out=out if i == len(layers) - 1 else None)
        saved.append(s)
    return x, saved


def _run_bwd(ctx, layers, dy, saved, env, out=None):
    for i in range(len(layers) - 1, -1, -1):
        dy = layers[i].bwd(ctx, dy, saved[i], env, out=out if i == 0 else None)
    return dy


class _TimeEmbed:
    """timestep_embedding -> Linear -> SiLU -> Linear (-> SiLU for the ResBlocks' emb_layers[0])."""

    def __init__(self, b: _Builder, cfg: NetCfg):
        self.l0, self.l2 = b.linear("time_embed.0"), b.linear("time_embed.2")
        half = cfg.model_channels // 2
        # exactly the reference's fp32 table (util.py:165-167), built on the host once
        freqs = torch.exp(-math.log(10000.0) * torch.arange(0, half, dtype=torch.float32) / half)
        self.freqs = freqs.to(b.device)
        self.mc, self.ted = cfg.model_channels, cfg.time_embed_dim

    def fwd(self, ctx: Ctx, t: torch.Tensor):
        B = t.shape[0]
        temb = ctx.new(B, self.mc)
        hip.timestep_embedding(t, self.freqs, temb)
        te0, t0 = linear_fwd(ctx, self.l0, temb)
        h = ctx.new(B, self.ted); hip.silu_fwd(te0, h)
        emb, t2 = linear_fwd(ctx, self.l2, h)
        semb = ctx.new(B, self.ted); hip.silu_fwd(emb, semb)
        return semb, (temb, te0, t0, h, emb, t2)

    def bwd(self, ctx: Ctx, dsemb, saved):
        temb, te0, t0, h, emb, t2 = saved
        demb = ctx.new(*emb.shape); hip.silu_bwd(emb, dsemb, demb)
        dh, u2 = linear_bwd_data(ctx, self.l2, demb)
        linear_bwd_lora(ctx, self.l2, h, t2, demb, u2)
        base_bwd_weight(ctx, self.l2, h, demb)
        dte0 = ctx.new(*te0.shape); hip.silu_bwd(te0, dh, dte0)
        if self.l0.r:
            u0 = ctx.new(dte0.shape[0], self.l0.r); hip.gemm(dte0, self.l0.Bt, u0)
            linear_bwd_lora(ctx, self.l0, temb, t0, dte0, u0)
        base_bwd_weight(ctx, self.l0, temb, dte0)
        ctx.drop_transposes()


def _encoder_layers(b: _Builder, cfg: NetCfg, lora: bool, after_block=None):
    """input_blocks (openaimodel.py:542-605 / cldm.py:139-237): list of layer lists + channel list.
    `after_block(k)` is called once block k (and finally the middle block, k = n) has been built, so
    the ControlNet can declare zero conv k right behind it (keeps every backward stage's trainables
    contiguous in the flat gradient buffer)."""
    mc = cfg.model_channels
    done = (lambda k: None) if after_block is None else after_block
    blocks = [[_Conv(b.conv3("input_blocks.0.0"), hip.CONV_S1)]]
    chans = [mc]
    done(0)
    ch, ds, idx = mc, 1, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [_Res(b.res(f"input_blocks.{idx}.0", ch, mult * mc))]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                layers.append(_ST(b.st(f"input_blocks.{idx}.1", ch, cfg.num_heads, lora), f"in{idx}"))
            blocks.append(layers)
            chans.append(ch)
            done(idx)
            idx += 1
        if level != len(cfg.channel_mult) - 1:
            blocks.append([_Conv(b.conv3(f"input_blocks.{idx}.0.op"), hip.CONV_S2)])
            chans.append(ch)
            done(idx)
            ds *= 2
            idx += 1
    mid = [_Res(b.res("middle_block.0", ch, ch)), _ST(b.st("middle_block.1", ch, cfg.num_heads, lora), "mid"),
           _Res(b.res("middle_block.2", ch, ch))]
    done(idx)
    return blocks, chans, mid, ds


# ------------------------------------------------------------------------------ ControlNet

class ControlNetE:
    def __init__(self, sd, cfg: NetCfg, dtype, device, prefix: str = "", need_bwd: bool = True,
                 trainables: Optional[TrainableSet] = None, layout_only: bool = False, train_all: bool = False,
                 lora_set: Optional[TrainableSet] = None, merge_lora: Optional[bool] = None):
        """merge_lora (default: on for executors built without a backward): fold W + B A into one packed weight per
        LoRA linear at every repack, as the reference's _fuse_lora does for inference (cldm/lora.py:297-318).
        layout_only: build the flat trainable layout (offsets, backward-ordered stage spans, the stage-completion
        hook) without packing anything for the kernels -- what the data-parallel exchange needs; usable without a
        GPU (the multi-process gloo tests).  Such an executor cannot run: fwd / bwd raise."""
        self.cfg, self.dtype, self.device = cfg, dtype, device
        self.layout_only = layout_only
        self.merge_lora = (not need_bwd) if merge_lora is None else bool(merge_lora)
        if os.environ.get("CTRLORA_MERGE_LORA", "1") == "0":
            self.merge_lora = False
        self.tr = trainables if trainables is not None else TrainableSet()
        self.train_all = train_all
        # pre-training: base weights in self.tr, the active task's LoRA bank in self.tr_lora (switch_bank swaps it)
        self.tr_lora = (lora_set if lora_set is not None else TrainableSet()) if train_all else self.tr
        b = _Builder(sd, prefix, dtype, device, need_bwd, self.tr, self.tr_lora, train_all)
        b.fold_lora = self.merge_lora
        self.lora = (prefix + "time_embed.0.lora_layer.down.weight") in sd
        self.time = _TimeEmbed(b, cfg)
        marks = [len(self.tr.items)]          # stage boundaries in declaration (= forward) order
        self.zero: List[LinearW] = []

        def after_block(k):
            name = f"zero_convs.{k}.0" if (prefix + f"zero_convs.{k}.0.weight") in sd else "middle_block_out.0"
            self.zero.append(b.zero_conv(name))
            marks.append(len(self.tr.items))

        self.blocks, self.chans, self.mid, _ = _encoder_layers(b, cfg, lora=True, after_block=after_block)
        # The ResBlocks' emb_layers (LoRA'd Linear on silu(emb), M = batch) depend on nothing but the time embedding: the ones
        # of equal width run as ONE grouped LoRA product at the start of the trunk (2 launches per width instead of 2 per block)
        self.emb_groups = []
        if self.lora and GROUP_LORA and dtype != torch.float32 and not train_all:   # (trainable biases alias the flat masters)
            by_n: Dict[int, list] = {}
            for l in [l for blk in list(self.blocks) + [self.mid] for l in blk if isinstance(l, _Res)]:
                if l.blk.emb.r:
                    by_n.setdefault(l.blk.emb.N, []).append(l)
            for n_, ls in sorted(by_n.items()):
                if len(ls) >= 2 and n_ % 64 == 0:
                    grp = LoraGroup([l.blk.emb for l in ls])
                    if self.merge_lora:
                        grp.enable_merge()
                    self.emb_groups.append((grp, ls))
        stage_items = [self.tr.items[marks[i]:marks[i + 1]] for i in range(len(marks) - 1)]
        time_items = self.tr.items[:marks[0]]
        # flat buffer in backward-completion order: middle stage first, time_embed last
        self.tr.items.reverse()
        # The grouped emb_layers also run their BACKWARD as grouped launches, once every block's d emb_out exists -- i.e. after
        # the last encoder stage: their LoRA gradients are final only then, so their trainables move from their blocks' stages
        # to the tail of the flat buffer, next to time_embed (the data-parallel hook reports a span when it is final).
        self.emb_sum = 0
        # order of rounds 1-3 (no hoist): optimizer states of those builds were saved as ONE flat tensor in this order
        self.legacy_item_names = [t.name for t in self.tr.items]
        if self.emb_groups and HOIST_EMB_BWD:
            hoisted = set()
            for grp, ls in self.emb_groups:
                for l in ls:
                    l.de_off = self.emb_sum
                    self.emb_sum += l.cout
                    hoisted.update(id(t) for t in (l.blk.emb.tA, l.blk.emb.tB) if t is not None)
            emb_items = [t for t in self.tr.items if id(t) in hoisted]
            time_ids = set(id(t) for t in time_items)
            stage_items = [[t for t in it if id(t) not in hoisted] for it in stage_items]
            self.tr.items = ([t for t in self.tr.items if id(t) not in hoisted and id(t) not in time_ids] + emb_items +
                             [t for t in self.tr.items if id(t) in time_ids])
            time_items = emb_items + time_items
        self.tr.materialize({k: v for k, v in sd.items()}, device)
        if train_all and self.tr_lora.flat is None:
            self.tr_lora.materialize({k: v for k, v in sd.items()}, device)

        def span(items):
            if not items:
                return (0, 0)
            lo = min(t.offset for t in items)
            hi = max(t.offset + rup(t.master.numel(), 64) for t in items)
            return (lo, hi)

        self.stage_spans = [span(it) for it in stage_items]   # index k = encoder stage k (last = middle)
        self.time_span = span(time_items)
        self.on_stage_done = None    # callable(start, end): that slice of flat_grad is final (DP overlap hook)
        self._b = b
        if not layout_only:
            self.repack()

    def backward_stage_order(self):
        """Spans of flat_grad in the order the backward pass finalises them (what `_done` reports)."""
        nb = len(self.blocks)
        return [self.stage_spans[nb]] + [self.stage_spans[k] for k in range(nb - 1, -1, -1)] + [self.time_span]

    def reload_frozen(self, sd):
        """module.load_state_dict() happened after this executor was built: refresh the packed frozen weights in
        place (the trainables are views of the flat masters the Parameters write through) and re-pack."""
        self._b.reload_frozen({k: v for k, v in sd.items()})
        self.repack()

    def switch_bank(self, lora_set: TrainableSet, repack: bool = True, force: bool = False):
        """Pre-training: make `lora_set` (same names / shapes, its own flat master + gradient buffers) the active LoRA
        bank (ControlNetPretrain.switch_lora, cldm_ctrlora_pretrain.py:68-76) and re-pack the LoRA copies from it.
        repack=False: only the host-side pointers move (a replayed hipGraph has already re-packed that bank);
        force=True: re-pack even if the bank is already the active one (first kernel of a captured per-task step)."""
        assert self.train_all
        if lora_set is self.tr_lora and not force:
            return
        for L in self._b.linears:
            if L.tA is not None:
                L.tA, L.tB = lora_set.by_name[L.tA.name], lora_set.by_name[L.tB.name]
        self.tr_lora = lora_set
        WEIGHTS_GENERATION[0] += 1
        if repack:
            self.repack(only=lora_set)

    def _repack_table(self, ts: TrainableSet):
        key = id(ts)
        tabs = self.__dict__.setdefault("_repack_tabs", {})
        tab = tabs.get(key)
        if tab is None:
            rows, prefix = [], [0]

            def add(row):
                R, C = row[1] >> 32, row[1] & 0xffffffff
                rows.append(row)
                prefix.append(prefix[-1] + ((R + 31) // 32) * ((C + 31) // 32))

            def mat(t, R, C, dst, dstT):
                # explicit row strides: members of a LoraGroup are views of the group's buffers
                add([t.offset, (R << 32) | C, 0 if dst is None else dst.data_ptr(), 0 if dstT is None else dstT.data_ptr(),
                     0, 0 if dst is None else dst.stride(0), 0 if dstT is None else dstT.stride(0), 0])

            for L in self._b.linears:
                if L.tA is not None and ts.by_name.get(L.tA.name) is L.tA:
                    mat(L.tA, L.r, L.K, L.A, L.At)
                    mat(L.tB, L.N, L.r, L.B, L.Bt)
                if L.tW is not None and ts.by_name.get(L.tW.name) is L.tW:
                    mat(L.tW, L.N, L.K, L.W, L.Wt)
            for cw in self._b.convs:
                if ts.by_name.get(cw.tW.name) is cw.tW:
                    for row in cw.repack_rows():
                        add(row)
            tab = (torch.tensor(rows, dtype=torch.int64, device=self.device).reshape(-1, 8) if rows else None,
                   torch.tensor(prefix, dtype=torch.int32, device=self.device), len(rows), prefix[-1])
            tabs[key] = tab
        return tab

    def repack(self, only: Optional[TrainableSet] = None):
        """Refresh the packed (storage dtype, both orientations) copies of every trainable matrix from the flat fp32
        masters: ONE kernel per flat buffer over a device-resident descriptor table (built on first use)."""
        WEIGHTS_GENERATION[0] += 1
        for L in self._b.linears:
            if L.tW is not None and L.tb is not None:
                L.bias = L.tb.master
        for cw in self._b.convs:
            cw.bias = cw.tb.master
        for n in self._b.norms:
            n.repack()      # views of the masters: nothing to copy
        sets = [only] if only is not None else ([self.tr] if self.tr_lora is self.tr else [self.tr, self.tr_lora])
        for ts in sets:
            desc, prefix, n, tiles = self._repack_table(ts)
            if n:
                hip.repack(self.dtype, ts.flat, desc, prefix, n, tiles)
        for L in self._b.linears:
            if self.merge_lora:
                L.merge_lora()
            L.invalidate_geglu()     # permuted (GEGLU-fused) copies are rebuilt lazily from the fresh B

    def fwd(self, ctx: Ctx, hint_tok, t, c, B, H, W, sinks, scales, weight=1.0, kv=None):
        """sinks[k] = (out_view, residual_view or None); out = (zero_conv_k(h_k)) * scale_k * weight + residual."""
        rec, hs = self.fwd_trunk(ctx, hint_tok, t, c, B, H, W, kv=kv)
        self.fwd_zero(hs, sinks, scales, weight)
        return rec

    def _runnable(self):
        if self.layout_only:
            raise RuntimeError("layout-only ControlNetE (no packed weights): it describes the trainable buffer, it cannot run")

    def fwd_trunk(self, ctx: Ctx, hint_tok, t, c, B, H, W, kv=None):
        """Encoder + middle block WITHOUT the zero convs (which need the UNet's skip tensors as residuals):
        this part is independent of the UNet encoder and may run concurrently with it on another stream.
        Returns (record or None, [h_k]) -- the 13 stage outputs the zero convs consume."""
        self._runnable()
        semb, tsv = self.time.fwd(ctx, t)
        env = _Env(B, H, W, semb, c, c.shape[0] // B)
        env.kv = kv
        emb_tt = []
        if self.emb_groups:
            env.emb_pre = {}
            for grp, ls in self.emb_groups:
                y, tt = group_fwd(ctx, grp, semb)
                emb_tt.append(tt)
                for i, l in enumerate(ls):
                    env.emb_pre[id(l)] = (y[:, i * grp.N:(i + 1) * grp.N],
                                          None if tt is None else tt[:, i * grp.r:(i + 1) * grp.r])
        h = hint_tok
        saved, hs, dims = [], [], []
        for k, layers in enumerate(self.blocks):
            h, sv = _run_fwd(ctx, layers, h, env)
            saved.append(sv); dims.append((env.H, env.W))
            hs.append(h)
        h, sv = _run_fwd(ctx, self.mid, h, env)
        saved.append(sv); dims.append((env.H, env.W))
        hs.append(h)
        return ((tsv, semb, saved, hs, dims, c, hint_tok, emb_tt) if ctx.record else None), hs

    def fwd_zero(self, hs, sinks, scales, weight=1.0):
        for k, h in enumerate(hs):
            self._zero_fwd(k, h, sinks[k], scales[k] * weight)

    def _zero_fwd(self, k, h, sink, alpha):
        out, res = sink
        z = self.zero[k]
        hip.gemm(h, z.W, out, bias=z.bias, alpha=alpha, residual=res, beta=1.0 if res is not None else 0.0)

    def bwd(self, ctx: Ctx, record, dsinks, scales, weight, B):
        tsv, semb, saved, hs, dims, c, hint_tok, emb_tt = record
        env = _Env(B, 0, 0, semb, c, c.shape[0] // B)
        env.emb_grads = True
        env.dsemb = ctx.zeros(B, self.cfg.time_embed_dim)
        if self.emb_sum:
            env.de_all = ctx.zeros(B, self.emb_sum, torch.float32)
        nb = len(self.blocks)
        # middle_block_out + middle block
        env.H, env.W = dims[nb]
        dh = self._zero_bwd(ctx, nb, hs[nb], dsinks[nb], scales[nb] * weight, None, B, env.H * env.W)
        dh = _run_bwd(ctx, self.mid, dh, saved[nb], env)
        self._done(ctx, self.stage_spans[nb])
        for k in range(nb - 1, -1, -1):
            env.H, env.W = dims[k]
            # stage 0: the input conv is frozen and the hint needs no gradient -> weight grads only
            dh = self._zero_bwd(ctx, k, hs[k], dsinks[k], scales[k] * weight, dh, B, env.H * env.W,
                                need_dx=k > 0 or self.train_all)
            if k > 0:
                dh = _run_bwd(ctx, self.blocks[k], dh, saved[k], env)
            elif self.train_all:      # pre-training also trains the input conv: weight gradient only (the hint needs none)
                conv3_bwd_weight(ctx, self.blocks[0][0].cw, hint_tok, dh, B, env.H, env.W)
            self._done(ctx, self.stage_spans[k])
        if env.de_all is not None:
            self._emb_bwd(ctx, env, semb, emb_tt, B)
        self.time.bwd(ctx, env.dsemb, tsv)
        self._done(ctx, self.time_span)
        ctx.retire_wgrad()      # every weight gradient of this network is ordered before what follows

    def _emb_bwd(self, ctx, env, semb, emb_tt, B):
        """Backward of every grouped emb_layers linear at once (openaimodel.py:254-274: emb_out enters h as a per-sample row
        bias, so d emb_out[b] is the column sum of dh over the sample's pixels -- the ResBlocks left those in env.de_all):
        per width ONE u = de B launch and ONE dsemb += [de | u] [W | A] launch instead of two (+ a clear, a pack and the
        split-K reduces) per block; the LoRA factor gradients dB += de^T t, dA += u^T silu(emb) join the weight-gradient queue."""
        de = ctx.new(B, self.emb_sum)
        hip.pack2d(env.de_all, de)
        for (grp, ls), tt in zip(self.emb_groups, emb_tt):
            o0 = ls[0].de_off
            de_g = de[:, o0:o0 + grp.G * grp.N]
            u = ctx.new(B, grp.G * grp.r)
            if grp.r % 64 == 0:
                hip.gemm(de_g, grp.Bt, u, k1=grp.N, a1_group_n=grp.r)
            else:                                 # rank below the narrowest tile: one small product per member (as AttnE._group_bwd)
                for i, l in enumerate(ls):
                    hip.gemm(de_g[:, i * grp.N:(i + 1) * grp.N], l.blk.emb.Bt, u[:, i * grp.r:(i + 1) * grp.r])
            hip.gemm(de_g, grp.Wt, env.dsemb, a2=u, w2=grp.At, residual=env.dsemb, beta=1.0)
            for i, l in enumerate(ls):
                linear_bwd_lora(ctx, l.blk.emb, semb, tt[:, i * grp.r:(i + 1) * grp.r], de_g[:, i * grp.N:(i + 1) * grp.N],
                                u[:, i * grp.r:(i + 1) * grp.r])

    def _done(self, ctx, span):
        """End of a backward stage: its queued weight gradients go out as one grouped launch, and once they are
        ordered before the main stream's next work (immediately, or one stage later when they ride the side stream)
        the stage's slice of the flat gradient buffer is reported final (data-parallel overlap hook)."""
        def report():
            if self.on_stage_done is not None and span[1] > span[0]:
                self.on_stage_done(span[0], span[1])
        ctx.flush_wgrad(after=report)

    def _zero_bwd(self, ctx, k, h, dz, alpha, dh_in, B, HW, need_dx=True):
        z = self.zero[k]
        dense_bwd_weight(ctx, z, h, dz, B, HW, alpha)
        ctx.drop_transposes()
        if not need_dx:
            return None
        out = ctx.new(h.shape[0], z.K)
        hip.gemm(dz, z.Wt, out, alpha=alpha, residual=dh_in, beta=1.0 if dh_in is not None else 0.0)
        return out


# ------------------------------------------------------------------------------ UNet

class UNetE:
    def __init__(self, sd, cfg: NetCfg, dtype, device, prefix: str = "", need_bwd: bool = True):
        self.cfg, self.dtype, self.device = cfg, dtype, device
        b = _Builder(sd, prefix, dtype, device, need_bwd, None)
        self.time = _TimeEmbed(b, cfg)
        self.blocks, self.chans, self.mid, ds = _encoder_layers(b, cfg, lora=False)
        mc = cfg.model_channels
        chans = list(self.chans)
        ch = mc * cfg.channel_mult[-1]
        self.dec: List[list] = []
        self.dec_c1: List[int] = []
        self.dec_c2: List[int] = []
        idx = 0
        for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
            for i in range(cfg.num_res_blocks + 1):
                ich = chans.pop()
                self.dec_c1.append(ch); self.dec_c2.append(ich)
                layers = [_Res(b.res(f"output_blocks.{idx}.0", ch + ich, mc * mult))]
                ch = mc * mult
                if ds in cfg.attention_resolutions:
                    layers.append(_ST(b.st(f"output_blocks.{idx}.{len(layers)}", ch, cfg.num_heads, False), f"out{idx}"))
                if level and i == cfg.num_res_blocks:
                    layers.append(_Conv(b.conv3(f"output_blocks.{idx}.{len(layers)}.conv"), hip.CONV_UP2))
                    ds //= 2
                self.dec.append(layers)
                idx += 1
        self.out_norm = b.norm("out.0")
        self.out_conv = b.conv3("out.2")
        from .blocks import GroupNormOp
        self.out_gn = GroupNormOp(self.out_norm, 1e-5, True)
        # The 22 emb_layers products (M = batch) and the 16 context K / V products (M = 77 * batch) of the frozen UNet
        # depend on nothing but silu(emb) / the text context: ONE product each at the start of the pass instead of 38
        # launches that cost their ~8 us launch floor apiece.  Row slices of the concatenated weights are the layers'.
        layers = [l for blk in list(self.blocks) + [self.mid] + list(self.dec) for l in blk]
        self._res = [l for l in layers if isinstance(l, _Res)]
        self._sts = [l for l in layers if isinstance(l, _ST) and l.blk.attn2.fused_kv is not None]
        self.emb_all = self.kv_all_w = None
        if os.environ.get("CTRLORA_BATCH_EMB", "1") != "0" and self._res:
            off = 0
            for l in self._res:
                l.emb_off = off; off += l.cout
            self.emb_all = LinearW(torch.cat([l.blk.emb.W.float() for l in self._res], 0),
                                   torch.cat([l.blk.emb.bias for l in self._res], 0), dtype, device, need_bwd=False)
            if self._sts:
                off = 0
                for l in self._sts:
                    l.kv_off = off; off += 2 * l.blk.attn2.inner
                fk = [l.blk.attn2.fused_kv for l in self._sts]
                bias = None if fk[0].bias is None else torch.cat([f.bias for f in fk], 0)
                self.kv_all_w = LinearW(torch.cat([f.W.float() for f in fk], 0), bias, dtype, device, need_bwd=False)
        self._pre = None         # (semb, emb_all output, c, kv_all output) of the running pass: encode -> decode

    def _precompute(self, ctx: Ctx, env: _Env, with_kv: bool):
        if self.emb_all is not None:
            env.emb_all, _ = linear_fwd(ctx, self.emb_all, env.semb)
        if with_kv and self.kv_all_w is not None:
            env.kv_all, _ = linear_fwd(ctx, self.kv_all_w, env.c)
        self._pre = (env.semb, env.emb_all, env.c, env.kv_all)

    # -- encoder + middle (never needs gradients: cldm/cldm.py:25-32 runs it under no_grad)
    def encode(self, ctx: Ctx, x_tok, t, c, B, H, W, kv=None):
        rec = ctx.record
        ctx.record = False
        semb, _ = self.time.fwd(ctx, t)
        env = _Env(B, H, W, semb, c, c.shape[0] // B)
        env.kv = kv
        self._precompute(ctx, env, with_kv=kv is None)
        hs, dims = [], []
        h = x_tok
        for layers in self.blocks:
            h, _ = _run_fwd(ctx, layers, h, env)
            hs.append(h); dims.append((env.H, env.W))
        h, _ = _run_fwd(ctx, self.mid, h, env)
        ctx.record = rec
        return semb, hs, dims, h

    def alloc_decoder_inputs(self, ctx: Ctx, B, dims):
        """One buffer per decoder block = its concatenated input [h | skip + control]."""
        bufs = []
        nd = len(self.dec)
        for i in range(nd):
            Hh, Ww = dims[nd - 1 - i]
            bufs.append(ctx.new(B * Hh * Ww, self.dec_c1[i] + self.dec_c2[i]))
        return bufs

    def control_sinks(self, bufs, hs, h_mid):
        """Where ControlNet zero-conv k must write, and what it must add (UNet skip / middle output)."""
        nd = len(self.dec)
        sinks = []
        for k in range(nd):
            i = nd - 1 - k
            sinks.append((bufs[i][:, self.dec_c1[i]:], hs[k]))
        sinks.append((bufs[0][:, :self.dec_c1[0]], h_mid))
        return sinks

    def fill_without_control(self, ctx, bufs, hs, h_mid, only_mid: Optional[torch.Tensor] = None):
        nd = len(self.dec)
        for k in range(nd):
            i = nd - 1 - k
            hip.axpby(hs[k], bufs[i][:, self.dec_c1[i]:], 1.0, 0.0)
        hip.axpby(h_mid, bufs[0][:, :self.dec_c1[0]], 1.0, 0.0)

    def decode(self, ctx: Ctx, bufs, semb, c, B, dims_mid, kv=None):
        env = _Env(B, dims_mid[0], dims_mid[1], semb, c, c.shape[0] // B)
        env.kv = kv
        pre = self._pre
        if pre is not None and pre[0] is semb and pre[2] is c:      # the batched products of THIS pass (encode made them)
            env.emb_all, env.kv_all = pre[1], pre[3]
        self._pre = None
        saved = []
        nd = len(self.dec)
        h = None
        for i, layers in enumerate(self.dec):
            out = bufs[i + 1][:, :self.dec_c1[i + 1]] if i + 1 < nd else None
            h, sv = _run_fwd(ctx, layers, bufs[i], env, out=out)
            saved.append((sv, (env.H, env.W)))
        M = h.shape[0]
        hn, st = self.out_gn.fwd(ctx, h, B, env.H * env.W)
        eps_tok = conv3_fwd(ctx, self.out_conv, hn, B, env.H, env.W, out_f32=True)      # [M, 32] fp32, 4 real
        rec = (saved, h, st, semb, c, (env.H, env.W)) if ctx.record else None
        return eps_tok, rec

    def decode_bwd(self, ctx: Ctx, d_eps_tok, rec, B):
        """d_eps_tok [M, 32] (engine dtype, channels >= out_channels zero).  Returns per-decoder-block
        gradients of the concatenated inputs (left half: previous block / middle, right half: skip+control)."""
        saved, h_last, st, semb, c, (H, W) = rec
        dhn = conv3_bwd_data(ctx, self.out_conv, d_eps_tok, B, H, W)
        dh = self.out_gn.bwd(ctx, h_last, dhn, st, B, H * W)
        nd = len(self.dec)
        dbufs = [None] * nd
        env = _Env(B, H, W, semb, c, c.shape[0] // B)
        for i in range(nd - 1, -1, -1):
            sv, (Ho, Wo) = saved[i]
            env.H, env.W = Ho, Wo
            dbufs[i] = _run_bwd(ctx, self.dec[i], dh, sv, env)
            dh = dbufs[i][:, :self.dec_c1[i]]
        return dbufs

    def control_grad_sinks(self, dbufs):
        nd = len(self.dec)
        ds = [dbufs[nd - 1 - k][:, self.dec_c1[nd - 1 - k]:] for k in range(nd)]
        ds.append(dbufs[0][:, :self.dec_c1[0]])
        return ds
