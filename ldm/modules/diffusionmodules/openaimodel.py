"""SD1.5 UNet module tree (API of the reference's ldm/modules/diffusionmodules/openaimodel.py).

`UNetModel(**unet_config.params)` builds the same parameter tree / state-dict keys as the reference
constructor (openaimodel.py:412-736: time_embed, input_blocks, middle_block, output_blocks, out);
`forward` runs on the HIP engine (ctrlora_amd.engine.UNetE).  ResBlock / Downsample / Upsample are
parameter containers (reference :90-274).
"""
import os

import torch
import torch.nn as nn

from ldm.modules.attention import SpatialTransformer, _EngineExecuted
from ldm.modules.diffusionmodules.util import conv_nd, linear, normalization, zero_module


class TimestepBlock(nn.Module):
    pass


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def forward(self, *a, **k):
        raise RuntimeError("TimestepEmbedSequential holds parameters only; run the enclosing network")


class Upsample(_EngineExecuted):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=padding)


class Downsample(_EngineExecuted):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if not use_conv:
            raise ValueError("CtrLoRA configs use conv_resample=True")
        self.channels, self.out_channels = channels, out_channels or channels
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)


class ResBlock(TimestepBlock, _EngineExecuted):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        if use_scale_shift_norm or up or down or use_conv:
            raise ValueError("option not used by the CtrLoRA configs")
        self.channels, self.emb_channels, self.out_channels = channels, emb_channels, out_channels or channels
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        self.skip_connection = (nn.Identity() if self.out_channels == channels
                                else conv_nd(dims, channels, self.out_channels, 1))


class EngineHost:
    """Mixin for the network roots (UNetModel, ControlNet): lazily builds the HIP executor from the
    module's own parameters."""
    engine_dtype = None   # torch.bfloat16 (default) / torch.float32 (parity mode)

    def _engine_dtype(self):
        if self.engine_dtype is not None:
            return self.engine_dtype
        env = os.environ.get("CTRLORA_ENGINE_DTYPE", "bf16").lower()
        return torch.float32 if env in ("f32", "fp32", "float32") else torch.bfloat16

    def net_cfg(self):
        from ctrlora_amd.engine import NetCfg
        return NetCfg(in_channels=self.in_channels, out_channels=getattr(self, "out_channels", 4),
                      model_channels=self.model_channels, channel_mult=tuple(self.channel_mult),
                      num_res_blocks=self.num_res_blocks[0], attention_resolutions=tuple(self.attention_resolutions),
                      num_heads=self.num_heads, context_dim=self.context_dim)

    def invalidate_engine(self):
        self.__dict__.pop("_exec", None)

    def _watch_state_loads(self):
        """The executor snapshots and packs the module's weights on first use; `load_state_dict` on this module or
        on any parent (the gradio app swaps sd / cn checkpoints at run time, trainers resume) must not leave it on
        the old weights."""
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._on_state_loaded())

    def _on_state_loaded(self):
        self.invalidate_engine()

    def _device(self):
        return next(self.parameters()).device


def _check_supported(num_res_blocks, channel_mult, num_heads, num_head_channels, use_spatial_transformer, legacy,
                     transformer_depth, resblock_updown, use_scale_shift_norm, dims):
    if isinstance(num_res_blocks, int):
        num_res_blocks = len(channel_mult) * [num_res_blocks]
    if len(set(num_res_blocks)) != 1:
        raise ValueError("per-level num_res_blocks is not used by the CtrLoRA configs")
    if (not use_spatial_transformer or legacy or transformer_depth != 1 or resblock_updown or use_scale_shift_norm
            or dims != 2 or num_heads == -1 or num_head_channels != -1):
        raise ValueError("unsupported UNet option for the CtrLoRA SD1.5 path "
                         "(needs use_spatial_transformer, legacy=False, transformer_depth=1, num_heads set)")
    return list(num_res_blocks)


def build_encoder(net, in_channels, model_channels, num_res_blocks, attention_resolutions, dropout, channel_mult,
                  conv_resample, dims, use_checkpoint, num_heads, context_dim, on_block=None):
    """time_embed + input_blocks + middle_block, shared by UNetModel and ControlNet
    (openaimodel.py:526-657, cldm/cldm.py:131-277).  `on_block(ch)` is called after every input block."""
    ted = model_channels * 4
    net.time_embed = nn.Sequential(linear(model_channels, ted), nn.SiLU(), linear(ted, ted))
    net.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
    chans, ch, ds = [model_channels], model_channels, 1
    if on_block:
        on_block(ch)
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks[level]):
            layers = [ResBlock(ch, ted, dropout, out_channels=mult * model_channels, dims=dims,
                               use_checkpoint=use_checkpoint)]
            ch = mult * model_channels
            if ds in attention_resolutions:
                layers.append(SpatialTransformer(ch, num_heads, ch // num_heads, depth=1, context_dim=context_dim,
                                                 use_checkpoint=use_checkpoint))
            net.input_blocks.append(TimestepEmbedSequential(*layers))
            chans.append(ch)
            if on_block:
                on_block(ch)
        if level != len(channel_mult) - 1:
            net.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
            chans.append(ch)
            if on_block:
                on_block(ch)
            ds *= 2
    net.middle_block = TimestepEmbedSequential(
        ResBlock(ch, ted, dropout, dims=dims, use_checkpoint=use_checkpoint),
        SpatialTransformer(ch, num_heads, ch // num_heads, depth=1, context_dim=context_dim,
                           use_checkpoint=use_checkpoint),
        ResBlock(ch, ted, dropout, dims=dims, use_checkpoint=use_checkpoint))
    return chans, ch, ds


class UNetModel(nn.Module, EngineHost):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True,
                 disable_self_attentions=None, num_attention_blocks=None, disable_middle_self_attn=False,
                 use_linear_in_transformer=False):
        super().__init__()
        if num_classes is not None or n_embed is not None or use_linear_in_transformer:
            raise ValueError("option not used by the CtrLoRA configs")
        if isinstance(context_dim, (list, tuple)):
            context_dim = context_dim[0]
        self.num_res_blocks = _check_supported(num_res_blocks, channel_mult, num_heads, num_head_channels,
                                               use_spatial_transformer, legacy, transformer_depth, resblock_updown,
                                               use_scale_shift_norm, dims)
        self.image_size, self.in_channels, self.model_channels, self.out_channels = (
            image_size, in_channels, model_channels, out_channels)
        self.attention_resolutions, self.dropout, self.channel_mult = list(attention_resolutions), dropout, list(channel_mult)
        self.conv_resample, self.use_checkpoint, self.num_heads, self.context_dim = (
            conv_resample, use_checkpoint, num_heads, context_dim)
        self.dtype = torch.float16 if use_fp16 else torch.float32
        chans, ch, ds = build_encoder(self, in_channels, model_channels, self.num_res_blocks, self.attention_resolutions,
                                      dropout, self.channel_mult, conv_resample, dims, use_checkpoint, num_heads,
                                      context_dim)
        ted = model_channels * 4
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(self.num_res_blocks[level] + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, out_channels=model_channels * mult, dims=dims,
                                   use_checkpoint=use_checkpoint)]
                ch = model_channels * mult
                if ds in self.attention_resolutions:
                    layers.append(SpatialTransformer(ch, num_heads, ch // num_heads, depth=1, context_dim=context_dim,
                                                     use_checkpoint=use_checkpoint))
                if level and i == self.num_res_blocks[level]:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(),
                                 zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))
        self._watch_state_loads()

    # ---- execution
    def executor(self):
        ex = self.__dict__.get("_exec")
        if ex is None:
            from ctrlora_amd.engine import UNetE
            sd = {k: v for k, v in self.state_dict().items()}
            ex = UNetE(sd, self.net_cfg(), self._engine_dtype(), self._device(), need_bwd=True)
            self.__dict__["_exec"] = ex
        return ex

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        from ctrlora_amd.engine import CtrLoRAEngine
        eng = CtrLoRAEngine.from_executors(self.executor(), [])
        return eng.forward(x, timesteps, context, None)
