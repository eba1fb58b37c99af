"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the CtrLoRA hot path.

A functional (state-dict driven, no nn.Module) plain-PyTorch restatement of the
reference's arithmetic for SURVEY.md section 8(a) rows a1-a17.  It is the
checker for the HIP engine: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it; the product never does.

Pinning: tests/golden/*.pt hold outputs of the *unmodified reference modules*
(imported from /root/reference in the build container by
tests/golden/make_golden.py) on the same key-addressed weights and inputs;
tests/test_oracle_golden.py checks this restatement against them.

Every function cites the reference lines it follows (paths relative to the
reference repo root).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .arch import ArchCfg, decoder_specs, encoder_specs, middle_spec

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- primitives

def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """ldm/modules/diffusionmodules/util.py:154-174 (repeat_only=False branch)."""
    half = dim // 2
    # built on the host and moved, as the reference does (util.py:166-168): the table is bit-identical on any device
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def linear(sd: SD, name: str, x: torch.Tensor) -> torch.Tensor:
    """nn.Linear, or LoRACompatibleLinear.forward when `<name>.lora_layer.*` exists:
    y = x W^T + b + 1.0 * up(down(x))   (cldm/lora.py:70-80,285-291; network_alpha=None)."""
    y = F.linear(x, sd[f"{name}.weight"], sd.get(f"{name}.bias"))
    dk = f"{name}.lora_layer.down.weight"
    if dk in sd:
        y = y + F.linear(F.linear(x, sd[dk]), sd[f"{name}.lora_layer.up.weight"])
    return y


def group_norm(sd: SD, name: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    """GroupNorm32(32, C) computed in fp32 (util.py:217-219); eps 1e-5 for ResBlocks
    (nn.GroupNorm default), 1e-6 for SpatialTransformer.norm (attention.py:88-89)."""
    return F.group_norm(x.float(), 32, sd[f"{name}.weight"], sd[f"{name}.bias"], eps)


def resblock(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """ResBlock._forward, non-updown / non-scale-shift path (openaimodel.py:254-274)."""
    h = F.conv2d(F.silu(group_norm(sd, f"{p}.in_layers.0", x, 1e-5)),
                 sd[f"{p}.in_layers.2.weight"], sd[f"{p}.in_layers.2.bias"], padding=1)
    emb_out = linear(sd, f"{p}.emb_layers.1", F.silu(emb))
    h = h + emb_out[:, :, None, None]
    h = F.conv2d(F.silu(group_norm(sd, f"{p}.out_layers.0", h, 1e-5)),
                 sd[f"{p}.out_layers.3.weight"], sd[f"{p}.out_layers.3.bias"], padding=1)
    if f"{p}.skip_connection.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.skip_connection.weight"], sd[f"{p}.skip_connection.bias"])
    return x + h


def cross_attention(sd: SD, p: str, x: torch.Tensor, ctx: Optional[torch.Tensor], heads: int) -> torch.Tensor:
    """CrossAttention.forward (attention.py:163-194): fp32 QK^T * d^-0.5, softmax, PV, to_out."""
    ctx = x if ctx is None else ctx
    q, k, v = linear(sd, f"{p}.to_q", x), linear(sd, f"{p}.to_k", ctx), linear(sd, f"{p}.to_v", ctx)
    b, n, inner = q.shape
    d = inner // heads

    def split(t):
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bhid,bhjd->bhij", q.float(), k.float()) * (d ** -0.5)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
    out = out.permute(0, 2, 1, 3).reshape(b, n, inner)
    return linear(sd, f"{p}.to_out.0", out)


def transformer_block(sd: SD, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int) -> torch.Tensor:
    """BasicTransformerBlock._forward (attention.py:271-275) + GEGLU FF (attention.py:49-76)."""
    def ln(n, t):
        return F.layer_norm(t, (t.shape[-1],), sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1e-5)

    x = cross_attention(sd, f"{p}.attn1", ln("norm1", x), None, heads) + x
    x = cross_attention(sd, f"{p}.attn2", ln("norm2", x), ctx, heads) + x
    h = linear(sd, f"{p}.ff.net.0.proj", ln("norm3", x))
    a, gate = h.chunk(2, dim=-1)
    x = linear(sd, f"{p}.ff.net.2", a * F.gelu(gate)) + x
    return x


def spatial_transformer(sd: SD, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int) -> torch.Tensor:
    """SpatialTransformer.forward, use_linear=False (attention.py:321-340)."""
    b, c, h, w = x.shape
    x_in = x
    x = group_norm(sd, f"{p}.norm", x, 1e-6)
    x = F.conv2d(x, sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"])
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    x = transformer_block(sd, f"{p}.transformer_blocks.0", x, ctx, heads)
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    x = F.conv2d(x, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    return x + x_in


def run_block(sd: SD, root: str, spec, x: torch.Tensor, emb: torch.Tensor, ctx: torch.Tensor) -> torch.Tensor:
    """TimestepEmbedSequential.forward dispatch (openaimodel.py:73-87)."""
    for kind, prefix, m in spec.layers:
        p = root + prefix
        if kind == "conv3":
            x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], padding=1)
        elif kind == "res":
            x = resblock(sd, p, x, emb)
        elif kind == "st":
            x = spatial_transformer(sd, p, x, ctx, m["heads"])
        elif kind == "down":      # Downsample (openaimodel.py:133-159): 3x3 stride 2 pad 1
            x = F.conv2d(x, sd[f"{p}.op.weight"], sd[f"{p}.op.bias"], stride=2, padding=1)
        elif kind == "up":        # Upsample (openaimodel.py:108-118): nearest x2, then 3x3
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = F.conv2d(x, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"], padding=1)
        else:
            raise ValueError(kind)
    return x


def time_embed(sd: SD, root: str, cfg: ArchCfg, t: torch.Tensor) -> torch.Tensor:
    """time_embed = Linear, SiLU, Linear (cldm.py:131-136, openaimodel.py:526-531)."""
    e = timestep_embedding(t, cfg.model_channels)
    return linear(sd, f"{root}time_embed.2", F.silu(linear(sd, f"{root}time_embed.0", e)))


# ----------------------------------------------------------------------------- networks

def controlnet_forward(sd: SD, cfg: ArchCfg, hint_z: torch.Tensor, t: torch.Tensor, ctx: torch.Tensor,
                       root: str = "") -> List[torch.Tensor]:
    """ControlNetFinetune.forward (cldm/cldm_ctrlora_finetune.py:40-54): the 4-channel
    *latent* hint goes where vanilla ControlNet feeds x_noisy; 13 zero-conv outputs."""
    emb = time_embed(sd, root, cfg, t)
    enc, _ = encoder_specs(cfg)
    outs = []
    h = hint_z.float()
    for k, spec in enumerate(enc):
        h = run_block(sd, root, spec, h, emb, ctx)
        outs.append(F.conv2d(h, sd[f"{root}zero_convs.{k}.0.weight"], sd[f"{root}zero_convs.{k}.0.bias"]))
    h = run_block(sd, root, middle_spec(cfg), h, emb, ctx)
    outs.append(F.conv2d(h, sd[f"{root}middle_block_out.0.weight"], sd[f"{root}middle_block_out.0.bias"]))
    return outs


def unet_forward(sd: SD, cfg: ArchCfg, x: torch.Tensor, t: torch.Tensor, ctx: torch.Tensor,
                 control: Optional[List[torch.Tensor]] = None, only_mid_control: bool = False,
                 root: str = "") -> torch.Tensor:
    """ControlledUnetModel.forward (cldm/cldm.py:22-45).  Encoder + middle run under
    no_grad in the reference; `control` is consumed back to front by pop()."""
    control = list(control) if control is not None else None
    enc, _ = encoder_specs(cfg)
    hs = []
    with torch.no_grad():
        emb = time_embed(sd, root, cfg, t)
        h = x.float()
        for spec in enc:
            h = run_block(sd, root, spec, h, emb, ctx)
            hs.append(h)
        h = run_block(sd, root, middle_spec(cfg), h, emb, ctx)
    if control is not None:
        h = h + control.pop()
    for spec in decoder_specs(cfg):
        if only_mid_control or control is None:
            h = torch.cat([h, hs.pop()], dim=1)
        else:
            h = torch.cat([h, hs.pop() + control.pop()], dim=1)
        h = run_block(sd, root, spec, h, emb, ctx)
    h = F.silu(group_norm(sd, f"{root}out.0", h, 1e-5))
    return F.conv2d(h, sd[f"{root}out.2.weight"], sd[f"{root}out.2.bias"], padding=1)


def apply_model(sd_cn: SD, sd_unet: SD, cfg: ArchCfg, x_noisy, t, ctx, hint_z,
                control_scales: Optional[Sequence[float]] = None) -> torch.Tensor:
    """ControlFinetuneLDM.apply_model after the hint has been VAE-encoded
    (cldm/cldm_ctrlora_finetune.py:67-82; `hint_z` = 0.18215 * posterior sample)."""
    control = controlnet_forward(sd_cn, cfg, hint_z, t, ctx)
    scales = control_scales if control_scales is not None else [1.0] * len(control)
    control = [c * s for c, s in zip(control, scales)]
    return unet_forward(sd_unet, cfg, x_noisy, t, ctx, control)


def apply_model_multi(sd_cns: Sequence[SD], weights: Sequence[float], sd_unet: SD, cfg: ArchCfg,
                      x_noisy, t, ctx, hint_zs, control_scales=None) -> torch.Tensor:
    """ControlInferenceLDM.apply_model (cldm/cldm_ctrlora_inference.py:156-178):
    weighted sum of the residual lists of several LoRA banks."""
    total = None
    for sd_cn, w, hz in zip(sd_cns, weights, hint_zs):
        c = controlnet_forward(sd_cn, cfg, hz, t, ctx)
        scales = control_scales if control_scales is not None else [1.0] * len(c)
        c = [ci * s * w for ci, s in zip(c, scales)]
        total = c if total is None else [a + b for a, b in zip(total, c)]
    return unet_forward(sd_unet, cfg, x_noisy, t, ctx, total)


# ----------------------------------------------------------------------------- schedules / losses

def make_schedule(timesteps: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.0120):
    """make_beta_schedule('linear') + DDPM.register_schedule
    (util.py:21-43, ldm/models/diffusion/ddpm.py:138-165): fp64 numpy -> fp32 buffers."""
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    return dict(betas=f32(betas), alphas_cumprod=f32(ac), alphas_cumprod_prev=f32(ac_prev),
                sqrt_alphas_cumprod=f32(np.sqrt(ac)), sqrt_one_minus_alphas_cumprod=f32(np.sqrt(1.0 - ac)))


def q_sample(sched, x_start, t, noise):
    """DDPM.q_sample (ddpm.py:356-359) with extract_into_tensor (util.py:96-99)."""
    a = sched["sqrt_alphas_cumprod"].gather(-1, t).reshape(-1, 1, 1, 1)
    b = sched["sqrt_one_minus_alphas_cumprod"].gather(-1, t).reshape(-1, 1, 1, 1)
    return a * x_start + b * noise


def p_losses(sd_cn, sd_unet, cfg, sched, x_start, t, ctx, hint_z, noise, control_scales=None):
    """LatentDiffusion.p_losses, eps-parameterisation, logvar == 0, l_simple_weight 1,
    original_elbo_weight 0 (ddpm.py:885-920)  ->  loss = mean((eps_hat - eps)^2)."""
    x_noisy = q_sample(sched, x_start, t, noise)
    eps = apply_model(sd_cn, sd_unet, cfg, x_noisy, t, ctx, hint_z, control_scales)
    loss_simple = ((eps - noise) ** 2).mean(dim=[1, 2, 3])
    return loss_simple.mean(), eps


def make_ddim_schedule(sched, S: int, eta: float = 0.0, num_ddpm: int = 1000):
    """DDIMSampler.make_schedule, 'uniform' (cldm/ddim_hacked.py:23-52; util.py:46-74).
    ddim_timesteps are +1 shifted; alphas is an fp32 tensor gather, alphas_prev is fp64 numpy."""
    c = num_ddpm // S
    ts = np.asarray(list(range(0, num_ddpm, c))) + 1
    ac = sched["alphas_cumprod"]
    alphas = ac[ts]                                                   # torch fp32
    alphas_prev = np.asarray([ac[0]] + ac[ts[:-1]].tolist())          # numpy fp64 (of fp32 values)
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return dict(timesteps=ts, alphas=alphas, alphas_prev=alphas_prev, sigmas=sigmas,
                sqrt_one_minus_alphas=np.sqrt(1.0 - alphas))


def ddim_step(x, e_cond, e_uncond, scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise=None):
    """DDIMSampler.p_sample_ddim arithmetic (ddim_hacked.py:192,203-231), eps-parameterisation.
    The four coefficients enter as fp32 scalars (torch.full(..., device) of a table entry)."""
    f = lambda v: torch.tensor(float(v), dtype=torch.float32)
    a_t, a_prev, sigma_t, s1m = f(a_t), f(a_prev), f(sigma_t), f(sqrt_one_minus_at)
    e_t = e_uncond + scale * (e_cond - e_uncond) if e_uncond is not None else e_cond
    pred_x0 = (x - s1m * e_t) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    nz = sigma_t * noise if noise is not None else 0.0
    return a_prev.sqrt() * pred_x0 + dir_xt + nz, pred_x0


def ddim_sample(eps_fn, sched, S, x_T, scale=1.0, uncond=False, eta=0.0, noises=None, keep=None):
    """DDIMSampler.ddim_sampling loop (ddim_hacked.py:123-178).  eps_fn(x, t_long, cond: bool).  keep: an optional list that
    receives the sample after every step (what the reference logs as intermediates['x_inter'] with log_every_t = 1)."""
    ds = make_ddim_schedule(sched, S, eta)
    img = x_T
    b = x_T.shape[0]
    steps = []
    for i, step in enumerate(np.flip(ds["timesteps"])):
        index = S - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        steps.append((index, int(step)))
        e_c = eps_fn(img, ts, True)
        e_u = eps_fn(img, ts, False) if (uncond and scale != 1.0) else None
        nz = noises[i] if noises is not None else None
        img, _ = ddim_step(img, e_c, e_u, scale, ds["alphas"][index], ds["alphas_prev"][index],
                           ds["sigmas"][index], ds["sqrt_one_minus_alphas"][index], nz)
        if keep is not None:
            keep.append(img)
    return img, steps


def ddim_encode(eps_fn, sched, S, x0, t_enc, scale=1.0, uncond=False, keep=None):
    """DDIMSampler.encode (ddim_hacked.py:234-279), DDIM tables (use_original_steps=False): deterministic inversion.
    The dtypes follow the reference: alphas_next = ddim_alphas (fp32 tensor), alphas = torch.tensor(ddim_alphas_prev) (fp64), so
    the two coefficients are formed in fp64 from 0-dim tensors and meet the fp32 sample at the multiply.
    eps_fn(x, t_long, cond: bool); keep: optional list receiving x_next after every step."""
    ds = make_ddim_schedule(sched, S, 0.0)
    assert t_enc <= ds["timesteps"].shape[0]
    alphas_next = ds["alphas"][:t_enc]
    alphas = torch.tensor(ds["alphas_prev"][:t_enc])
    x_next = x0
    for i in range(t_enc):
        t = torch.full((x0.shape[0],), int(ds["timesteps"][i]), dtype=torch.long)
        e = eps_fn(x_next, t, True)
        if scale != 1.0:
            assert uncond
            e_u = eps_fn(x_next, t, False)
            e = e_u + scale * (e - e_u)
        xw = (alphas_next[i] / alphas[i]).sqrt() * x_next
        we = alphas_next[i].sqrt() * ((1 / alphas_next[i] - 1).sqrt() - (1 / alphas[i] - 1).sqrt()) * e
        x_next = xw + we
        if keep is not None:
            keep.append(x_next)
    return x_next


def ddim_encode_keep_steps(t_enc, return_intermediates):
    """Which step indices DDIMSampler.encode records as intermediates (ddim_hacked.py:267-272)."""
    out = []
    for i in range(t_enc):
        if return_intermediates and i % (t_enc // return_intermediates) == 0 and i < t_enc - 1:
            out.append(i)
        elif return_intermediates and i >= t_enc - 2:
            out.append(i)
    return out


def ddim_stochastic_encode(sched, S, x0, t, noise, use_original_steps=False):
    """DDIMSampler.stochastic_encode (ddim_hacked.py:282-295): q(x_t | x0) with t indexing the DDIM (or DDPM) table."""
    if use_original_steps:
        # the SAMPLER's own tables (ddim_hacked.py:37-38): fp32 square roots of the fp32 alphas_cumprod -- not DDPM's
        # (fp64 square roots rounded to fp32), which differ in the last bit
        ac = sched["alphas_cumprod"]
        sa, s1m = torch.sqrt(ac), torch.sqrt(1.0 - ac)
    else:
        ds = make_ddim_schedule(sched, S, 0.0)
        sa, s1m = torch.sqrt(ds["alphas"]), torch.as_tensor(np.asarray(ds["sqrt_one_minus_alphas"]))
    ex = lambda a: a.gather(-1, t).reshape(x0.shape[0], *((1,) * (x0.dim() - 1)))
    return ex(sa) * x0 + ex(s1m) * noise


def ddim_decode(eps_fn, sched, S, x_latent, t_start, scale=1.0, uncond=False, eta=0.0, noises=None):
    """DDIMSampler.decode (ddim_hacked.py:298-317): p_sample_ddim over the first t_start DDIM timesteps, flipped."""
    ds = make_ddim_schedule(sched, S, eta)
    ts = ds["timesteps"][:t_start]
    x = x_latent
    for i, step in enumerate(np.flip(ts)):
        index = ts.shape[0] - i - 1
        t = torch.full((x.shape[0],), int(step), dtype=torch.long)
        e_c = eps_fn(x, t, True)
        e_u = eps_fn(x, t, False) if (uncond and scale != 1.0) else None
        nz = noises[i] if noises is not None else None
        x, _ = ddim_step(x, e_c, e_u, scale, ds["alphas"][index], ds["alphas_prev"][index], ds["sigmas"][index],
                         ds["sqrt_one_minus_alphas"][index], nz)
    return x


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=1e-2):
    """torch.optim.AdamW defaults as used by configure_optimizers
    (cldm/cldm_ctrlora_finetune.py:105): decoupled weight decay, bias-corrected moments."""
    p = p * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    return p - (lr / bc1) * m / denom, m, v
