"""Parameter containers for the transformer part of the SD1.5 UNet / ControlNet.

Same module tree and parameter names as the reference's ldm/modules/attention.py
(CrossAttention :145-162, GEGLU :49-53, FeedForward :59-73, BasicTransformerBlock :246-267,
SpatialTransformer :278-319) so its state dicts load unchanged.  These modules hold weights only:
the arithmetic runs in the HIP engine (ctrlora_amd/engine/blocks.py), which reads the parameters by
name from the enclosing network.  Calling an inner block on its own is not part of the hot path.
"""
import torch.nn as nn


class _EngineExecuted(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} holds parameters only; run the enclosing ControlNet / UNet "
                           "(executed by ctrlora_amd.engine on the GPU)")


class GEGLU(_EngineExecuted):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(_EngineExecuted):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        if not glu:
            raise ValueError("CtrLoRA configs use gated_ff=True (GEGLU)")
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim))


class CrossAttention(_EngineExecuted):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))


class BasicTransformerBlock(_EngineExecuted):
    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False):
        super().__init__()
        if disable_self_attn:
            raise ValueError("disable_self_attn is not used by the CtrLoRA configs")
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)


class SpatialTransformer(_EngineExecuted):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, disable_self_attn=False,
                 use_linear=False, use_checkpoint=True):
        super().__init__()
        if use_linear or depth != 1:
            raise ValueError("SD1.5 CtrLoRA configs use conv proj_in/out and transformer_depth=1")
        if isinstance(context_dim, (list, tuple)):
            context_dim = context_dim[0]
        inner = n_heads * d_head
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, kernel_size=1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, kernel_size=1)
        for p in self.proj_out.parameters():      # zero_module (attention.py:312)
            p.detach().zero_()
