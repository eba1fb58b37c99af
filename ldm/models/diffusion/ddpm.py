"""DDPM / LatentDiffusion wrappers without pytorch-lightning (API subset of the reference's
ldm/models/diffusion/ddpm.py that the CtrLoRA hot path touches):

  DDPM.register_schedule  :138-192   q_sample :356-359   get_loss :383-397   training_step :432-454
  LatentDiffusion.forward :839-848   p_losses :885-920   get_first_stage_encoding :655-662
  DiffusionWrapper        :1312-1352

Schedule buffers are computed exactly as the reference does (fp64 numpy -> fp32) so timestep / index
bookkeeping is bit-exact; q_sample and the MSE loss run as HIP kernels when the latents are on the GPU.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from ldm.modules.diffusionmodules.util import extract_into_tensor, make_beta_schedule
from ldm.util import default, exists, instantiate_from_config


class _LightningFree(nn.Module):
    """The handful of LightningModule facilities the reference code relies on."""
    global_step = 0
    current_epoch = 0

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def log(self, *a, **k):
        pass

    def log_dict(self, d, *a, **k):
        self.last_logged = {k_: (float(v) if torch.is_tensor(v) and v.numel() == 1 else v) for k_, v in d.items()}


class DiffusionWrapper(nn.Module):
    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_concat=None, c_crossattn=None, **kw):
        if self.conditioning_key != "crossattn":
            raise NotImplementedError("CtrLoRA uses conditioning_key='crossattn'")
        return self.diffusion_model(x, t, context=torch.cat(c_crossattn, 1))


class DDPM(_LightningFree):
    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", loss_type="l2", ckpt_path=None,
                 ignore_keys=(), load_only_unet=False, monitor="val/loss", use_ema=True, first_stage_key="image",
                 image_size=256, channels=3, log_every_t=100, clip_denoised=True, linear_start=1e-4, linear_end=2e-2,
                 cosine_s=8e-3, given_betas=None, original_elbo_weight=0., v_posterior=0., l_simple_weight=1.,
                 conditioning_key=None, parameterization="eps", scheduler_config=None, use_positional_encodings=False,
                 learn_logvar=False, logvar_init=0., make_it_fit=False, ucg_training=None, reset_ema=False,
                 reset_num_ema_updates=False):
        super().__init__()
        if parameterization != "eps" or use_ema or learn_logvar:
            raise NotImplementedError("CtrLoRA configs: eps-parameterisation, use_ema=False, fixed logvar")
        self.parameterization, self.clip_denoised, self.log_every_t = parameterization, clip_denoised, log_every_t
        self.first_stage_key, self.image_size, self.channels = first_stage_key, image_size, channels
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.use_ema, self.use_scheduler = False, scheduler_config is not None
        self.v_posterior, self.original_elbo_weight, self.l_simple_weight = v_posterior, original_elbo_weight, l_simple_weight
        self.monitor = monitor
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        self.loss_type = loss_type
        self.learn_logvar = False
        self.register_buffer("logvar", torch.full(fill_value=logvar_init, size=(self.num_timesteps,)))

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if exists(given_betas) else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        f32 = partial(torch.tensor, dtype=torch.float32)
        self.register_buffer("betas", f32(betas))
        self.register_buffer("alphas_cumprod", f32(alphas_cumprod))
        self.register_buffer("alphas_cumprod_prev", f32(alphas_cumprod_prev))
        self.register_buffer("sqrt_alphas_cumprod", f32(np.sqrt(alphas_cumprod)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", f32(np.sqrt(1. - alphas_cumprod)))
        self.register_buffer("log_one_minus_alphas_cumprod", f32(np.log(1. - alphas_cumprod)))
        self.register_buffer("sqrt_recip_alphas_cumprod", f32(np.sqrt(1. / alphas_cumprod)))
        self.register_buffer("sqrt_recipm1_alphas_cumprod", f32(np.sqrt(1. / alphas_cumprod - 1)))
        posterior_variance = ((1 - self.v_posterior) * betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod)
                              + self.v_posterior * betas)
        self.register_buffer("posterior_variance", f32(posterior_variance))
        self.register_buffer("posterior_log_variance_clipped", f32(np.log(np.maximum(posterior_variance, 1e-20))))
        self.register_buffer("posterior_mean_coef1", f32(betas * np.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod)))
        self.register_buffer("posterior_mean_coef2",
                             f32((1. - alphas_cumprod_prev) * np.sqrt(alphas) / (1. - alphas_cumprod)))
        lvlb = self.betas ** 2 / (2 * self.posterior_variance * f32(alphas) * (1 - self.alphas_cumprod))
        lvlb[0] = lvlb[1]
        self.register_buffer("lvlb_weights", lvlb, persistent=False)

    def q_sample(self, x_start, t, noise=None):
        noise = default(noise, lambda: torch.randn_like(x_start))
        if x_start.is_cuda and x_start.dtype == torch.float32 and not x_start.requires_grad:
            from ctrlora_amd import hip
            out = torch.empty_like(x_start)
            return hip.qsample(x_start.contiguous(), noise.float().contiguous(), t.long().contiguous(),
                               self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, out)
        return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def get_loss(self, pred, target, mean=True):
        if self.loss_type == "l1":
            loss = (target - pred).abs()
        elif self.loss_type == "l2":
            loss = torch.nn.functional.mse_loss(target, pred, reduction="none")
        else:
            raise NotImplementedError(f"unknown loss type '{self.loss_type}'")
        return loss.mean() if mean else loss

    def training_step(self, batch, batch_idx=0):
        loss, loss_dict = self.shared_step(batch)
        self.log_dict(loss_dict)
        return loss


class LatentDiffusion(DDPM):
    def __init__(self, first_stage_config, cond_stage_config, num_timesteps_cond=None, cond_stage_key="image",
                 cond_stage_trainable=False, concat_mode=True, cond_stage_forward=None, conditioning_key=None,
                 scale_factor=1.0, scale_by_std=False, force_null_conditioning=False, *args, **kwargs):
        self.num_timesteps_cond = default(num_timesteps_cond, 1)
        self.scale_by_std = scale_by_std
        if conditioning_key is None:
            conditioning_key = "concat" if concat_mode else "crossattn"
        kwargs.pop("ckpt_path", None)
        kwargs.pop("ignore_keys", None)
        super().__init__(*args, conditioning_key=conditioning_key, **kwargs)
        self.concat_mode, self.cond_stage_trainable, self.cond_stage_key = concat_mode, cond_stage_trainable, cond_stage_key
        self.scale_factor = scale_factor
        self.shorten_cond_schedule = self.num_timesteps_cond > 1
        self.first_stage_model = instantiate_from_config(first_stage_config).eval()
        for p in self.first_stage_model.parameters():
            p.requires_grad = False
        self.cond_stage_model = instantiate_from_config(cond_stage_config)
        if not cond_stage_trainable and isinstance(self.cond_stage_model, nn.Module):
            self.cond_stage_model.eval()
            for p in self.cond_stage_model.parameters():
                p.requires_grad = False
        self.cond_stage_forward = cond_stage_forward
        self.clip_denoised = False

    # ---- frozen encoders (outside the hand-written path)
    def get_first_stage_encoding(self, encoder_posterior):
        z = encoder_posterior.sample() if hasattr(encoder_posterior, "sample") else encoder_posterior
        return self.scale_factor * z

    @torch.no_grad()
    def encode_first_stage(self, x):
        return self.first_stage_model.encode(x)

    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        return self.first_stage_model.decode(1. / self.scale_factor * z)

    def get_learned_conditioning(self, c):
        m = self.cond_stage_model
        return m.encode(c) if hasattr(m, "encode") and callable(m.encode) else m(c)

    @torch.no_grad()
    def get_input(self, batch, k, return_first_stage_outputs=False, force_c_encode=False, cond_key=None,
                  return_original_cond=False, bs=None, return_x=False):
        x = batch[k]
        if x.dim() == 3:
            x = x[..., None]
        x = x.permute(0, 3, 1, 2).to(memory_format=torch.contiguous_format).float()
        if bs is not None:
            x = x[:bs]
        x = x.to(self.device)
        z = self.get_first_stage_encoding(self.encode_first_stage(x)).detach()
        xc = batch[cond_key or self.cond_stage_key]
        if bs is not None:
            xc = xc[:bs]
        c = self.get_learned_conditioning(xc) if (not self.cond_stage_trainable or force_c_encode) else xc
        return [z, c]

    def shared_step(self, batch, **kwargs):
        x, c = self.get_input(batch, self.first_stage_key)
        return self(x, c)

    def forward(self, x, c, *args, **kwargs):
        t = torch.randint(0, self.num_timesteps, (x.shape[0],), device=self.device).long()
        return self.p_losses(x, c, t, *args, **kwargs)

    def apply_model(self, x_noisy, t, cond, return_ids=False):
        if not isinstance(cond, dict):
            cond = {"c_crossattn": cond if isinstance(cond, list) else [cond]}
        return self.model(x_noisy, t, **cond)

    def p_losses(self, x_start, cond, t, noise=None):
        noise = default(noise, lambda: torch.randn_like(x_start))
        x_noisy = self.q_sample(x_start=x_start, t=t, noise=noise)
        model_output = self.apply_model(x_noisy, t, cond)
        prefix = "train" if self.training else "val"
        target = noise
        if model_output.is_cuda and self.loss_type == "l2" and self.original_elbo_weight == 0.:
            # logvar == 0 and original_elbo_weight == 0  =>  loss = l_simple_weight * mean((eps-target)^2); the three
            # logged scalars come out of one deterministic HIP reduction
            from ctrlora_amd.train import PLossFn
            out = PLossFn.apply(model_output, target, t, self.lvlb_weights, self.l_simple_weight, 0.0)
            loss = out[2]
            return loss, {f"{prefix}/loss_simple": out[0].detach(), f"{prefix}/loss_vlb": out[1].detach(),
                          f"{prefix}/loss": loss.detach()}
        loss_simple = self.get_loss(model_output, target, mean=False).mean([1, 2, 3])
        logvar_t = self.logvar[t].to(self.device)
        loss = self.l_simple_weight * (loss_simple / torch.exp(logvar_t) + logvar_t).mean()
        loss_vlb = (self.lvlb_weights[t] * loss_simple).mean()
        loss = loss + self.original_elbo_weight * loss_vlb
        return loss, {f"{prefix}/loss_simple": loss_simple.mean(), f"{prefix}/loss_vlb": loss_vlb,
                      f"{prefix}/loss": loss}
