"""AutoencoderKL (API of ldm/models/autoencoder.py:11-119 + ldm/modules/diffusionmodules/model.py).

Same module tree / state-dict keys as the reference so the SD checkpoint loads.  On a GPU, under no_grad (the first
stage is frozen), encode / decode run on the HIP engine (ctrlora_amd/engine/vae.py: the implicit-GEMM convs, GroupNorm
+ swish, the single-head 512-channel attention as two MFMA GEMMs around a row softmax) -- SURVEY.md row 8(f1); the
plain torch modules below remain the CPU path and the definition of the parameter tree.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ldm.modules.distributions.distributions import DiagonalGaussianDistribution


def _norm(c):
    return nn.GroupNorm(num_groups=32, num_channels=c, eps=1e-6, affine=True)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = _norm(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = _norm(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(self.dropout(F.silu(self.norm2(h))))
        if self.in_channels != self.out_channels:
            x = self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.norm = _norm(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    def forward(self, x):
        b, c, h, w = x.shape
        hn = self.norm(x)
        q, k, v = self.q(hn).reshape(b, c, h * w), self.k(hn).reshape(b, c, h * w), self.v(hn).reshape(b, c, h * w)
        a = F.scaled_dot_product_attention(q.transpose(1, 2)[:, None], k.transpose(1, 2)[:, None],
                                           v.transpose(1, 2)[:, None])[:, 0]
        return x + self.proj_out(a.transpose(1, 2).reshape(b, c, h, w))


class _Down(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 2, 0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))       # asymmetric pad (model.py:80-84)


class _Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 1, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, **ignore):
        super().__init__()
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        cur = resolution
        for i in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            bin_, bout = ch * in_mult[i], ch * ch_mult[i]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=bin_, out_channels=bout, dropout=dropout, temb_channels=0))
                bin_ = bout
                if cur in attn_resolutions:
                    attn.append(AttnBlock(bin_))
            d = nn.Module()
            d.block, d.attn = block, attn
            if i != self.num_resolutions - 1:
                d.downsample = _Down(bin_)
                cur //= 2
            self.down.append(d)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=bin_, out_channels=bin_, dropout=dropout, temb_channels=0)
        self.mid.attn_1 = AttnBlock(bin_)
        self.mid.block_2 = ResnetBlock(in_channels=bin_, out_channels=bin_, dropout=dropout, temb_channels=0)
        self.norm_out = _norm(bin_)
        self.conv_out = nn.Conv2d(bin_, 2 * z_channels if double_z else z_channels, 3, 1, 1)

    def forward(self, x):
        h = self.conv_in(x)
        for i, d in enumerate(self.down):
            for j, blk in enumerate(d.block):
                h = blk(h)
                if len(d.attn) > 0:
                    h = d.attn[j](h)
            if i != self.num_resolutions - 1:
                h = d.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.conv_out(F.silu(self.norm_out(h)))


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, **ignore):
        super().__init__()
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        bin_ = ch * ch_mult[-1]
        cur = resolution // 2 ** (self.num_resolutions - 1)
        self.conv_in = nn.Conv2d(z_channels, bin_, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=bin_, out_channels=bin_, dropout=dropout, temb_channels=0)
        self.mid.attn_1 = AttnBlock(bin_)
        self.mid.block_2 = ResnetBlock(in_channels=bin_, out_channels=bin_, dropout=dropout, temb_channels=0)
        self.up = nn.ModuleList()
        for i in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            bout = ch * ch_mult[i]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=bin_, out_channels=bout, dropout=dropout, temb_channels=0))
                bin_ = bout
                if cur in attn_resolutions:
                    attn.append(AttnBlock(bin_))
            u = nn.Module()
            u.block, u.attn = block, attn
            if i != 0:
                u.upsample = _Up(bin_)
                cur *= 2
            self.up.insert(0, u)
        self.norm_out = _norm(bin_)
        self.conv_out = nn.Conv2d(bin_, out_ch, 3, 1, 1)

    def forward(self, z):
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(self.conv_in(z))))
        for i in reversed(range(self.num_resolutions)):
            for j, blk in enumerate(self.up[i].block):
                h = blk(h)
                if len(self.up[i].attn) > 0:
                    h = self.up[i].attn[j](h)
            if i != 0:
                h = self.up[i].upsample(h)
        return self.conv_out(F.silu(self.norm_out(h)))


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, ema_decay=None, learn_logvar=False):
        super().__init__()
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        assert ddconfig["double_z"]
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        self.ddconfig = dict(ddconfig)
        self.engine_dtype = None          # torch.bfloat16 (default) / torch.float32 (parity mode); set by ControlLDM
        self.use_engine = True            # False: the plain torch modules below (CPU, or A/B against MIOpen)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_engine())

    # ---- HIP engine (ctrlora_amd/engine/vae.py): packed once from this module's own (frozen) parameters
    def invalidate_engine(self):
        self.__dict__.pop("_enc", None)
        self.__dict__.pop("_dec", None)

    def _engine(self, which: str):
        ex = self.__dict__.get("_" + which)
        if ex is None:
            import os
            from ctrlora_amd.engine.vae import VAEDecoderE, VAEEncoderE
            dtype = self.engine_dtype
            if dtype is None:
                env = os.environ.get("CTRLORA_ENGINE_DTYPE", "bf16").lower()
                dtype = torch.float32 if env in ("f32", "fp32", "float32") else torch.bfloat16
            dev = next(self.parameters()).device
            cls = VAEEncoderE if which == "enc" else VAEDecoderE
            ex = cls({k: v for k, v in self.state_dict().items()}, self.ddconfig, dtype, dev)
            self.__dict__["_" + which] = ex
        return ex

    def _on_engine(self, x):
        return (self.use_engine and x.is_cuda and not torch.is_grad_enabled() and not self.ddconfig.get("attn_resolutions")
                and x.shape[-1] % 8 == 0 and x.shape[-2] % 8 == 0)

    # the middle AttnBlock runs on N = (H / 8) * (W / 8) latent positions and cl_softmax_rows holds a row of at most 8192
    # scores: larger images (768 x 768 -> N = 9216) take the torch modules, as the reference does, instead of raising
    ENGINE_MAX_ATTN_TOKENS = 8192

    def encode(self, x):
        if (self._on_engine(x) and (x.shape[-1] * x.shape[-2]) % 2048 == 0
                and (x.shape[-1] // 8) * (x.shape[-2] // 8) <= self.ENGINE_MAX_ATTN_TOKENS):
            return DiagonalGaussianDistribution(self._engine("enc")(x))
        return DiagonalGaussianDistribution(self.quant_conv(self.encoder(x)))

    def decode(self, z):
        if (self._on_engine(z) and (z.shape[-1] * z.shape[-2]) % 32 == 0
                and z.shape[-1] * z.shape[-2] <= self.ENGINE_MAX_ATTN_TOKENS):
            return self._engine("dec")(z)
        return self.decoder(self.post_quant_conv(z))

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        return self.decode(posterior.sample() if sample_posterior else posterior.mode()), posterior
