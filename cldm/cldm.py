"""ControlledUnetModel / ControlNet / ControlLDM (API of the reference's cldm/cldm.py).

Same constructor kwargs, attribute names and state-dict keys; the arithmetic runs on the HIP
engine.  What the reference does in each place is cited next to the method that replaces it.
"""
import contextlib

import torch
import torch.nn as nn

from ldm.models.diffusion.ddpm import LatentDiffusion
from ldm.modules.diffusionmodules.openaimodel import (EngineHost, TimestepEmbedSequential, UNetModel, _check_supported,
                                                      build_encoder)
from ldm.modules.diffusionmodules.util import conv_nd, zero_module
from ldm.util import instantiate_from_config


class ControlledUnetModel(UNetModel):
    """cldm/cldm.py:21-45 -- frozen SD UNet; encoder + middle without gradients, `control` residuals
    added to the middle output and to every skip connection (consumed back to front with pop())."""

    def forward(self, x, timesteps=None, context=None, control=None, only_mid_control=False, **kwargs):
        from ctrlora_amd.engine import CtrLoRAEngine
        eng = CtrLoRAEngine.from_executors(self.executor(), [])
        return eng.forward_external_control(x, timesteps, context, control, only_mid_control)


class ControlNet(nn.Module, EngineHost):
    """cldm/cldm.py:48-305 -- trainable copy of the UNet encoder + 13 zero convs.  The CtrLoRA variants
    (cldm_ctrlora_*) delete `input_hint_block` and feed a 4-channel VAE latent of the condition image."""

    def __init__(self, image_size, in_channels, model_channels, hint_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, use_checkpoint=False,
                 use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True,
                 disable_self_attentions=None, num_attention_blocks=None, disable_middle_self_attn=False,
                 use_linear_in_transformer=False):
        super().__init__()
        if isinstance(context_dim, (list, tuple)):
            context_dim = context_dim[0]
        self.num_res_blocks = _check_supported(num_res_blocks, channel_mult, num_heads, num_head_channels,
                                               use_spatial_transformer, legacy, transformer_depth, resblock_updown,
                                               use_scale_shift_norm, dims)
        self.dims, self.image_size, self.in_channels, self.model_channels = dims, image_size, in_channels, model_channels
        self.attention_resolutions, self.dropout, self.channel_mult = list(attention_resolutions), dropout, list(channel_mult)
        self.conv_resample, self.use_checkpoint, self.num_heads, self.context_dim = (
            conv_resample, use_checkpoint, num_heads, context_dim)
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.zero_convs = nn.ModuleList([])
        _, ch, _ = build_encoder(self, in_channels, model_channels, self.num_res_blocks, self.attention_resolutions,
                                 dropout, self.channel_mult, conv_resample, dims, use_checkpoint, num_heads,
                                 context_dim, on_block=lambda c: self.zero_convs.append(self.make_zero_conv(c)))
        # vanilla-ControlNet image-hint encoder (cldm.py:147-163); CtrLoRA deletes it right away
        widths = [(hint_channels, 16, 1), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 96, 2), (96, 96, 1), (96, 256, 2)]
        hint_layers = []
        for cin, cout, stride in widths:
            hint_layers += [conv_nd(dims, cin, cout, 3, padding=1, stride=stride), nn.SiLU()]
        hint_layers.append(zero_module(conv_nd(dims, 256, model_channels, 3, padding=1)))
        self.input_hint_block = TimestepEmbedSequential(*hint_layers)
        self.middle_block_out = self.make_zero_conv(ch)
        self._watch_state_loads()

    def _on_state_loaded(self):
        """Keep the executor (an optimizer may hold its flat master / gradient buffers): the bound trainable
        Parameters wrote through to the masters during the load; frozen weights are re-packed in place."""
        ex = self.__dict__.get("_exec")
        if ex is not None:
            ex.reload_frozen(self._executor_state())
            self.__dict__["_bound_version"] = None

    def make_zero_conv(self, channels):
        return TimestepEmbedSequential(zero_module(conv_nd(self.dims, channels, channels, 1, padding=0)))

    # ---- execution (CtrLoRA form: latent hint)
    def _executor_state(self):
        return {k: v for k, v in self.state_dict().items()}

    def executor(self):
        ex = self.__dict__.get("_exec")
        if ex is None:
            from ctrlora_amd.engine import ControlNetE
            from ctrlora_amd.train import bind_trainables
            ex = ControlNetE(self._executor_state(), self.net_cfg(), self._engine_dtype(), self._device(),
                             train_all=bool(getattr(self, "train_all_weights", False)))
            self.__dict__["_exec"] = ex
            self.__dict__["_bound"] = bind_trainables(self, ex)
        return ex

    def _latent_forward(self, hint, timesteps, context):
        from ctrlora_amd.engine import CtrLoRAEngine
        eng = CtrLoRAEngine.__new__(CtrLoRAEngine)
        ex = self.executor()
        eng.cfg, eng.dtype, eng.device, eng.unet, eng.controls = ex.cfg, ex.dtype, ex.device, None, [ex]
        return eng.control_outputs(hint, timesteps, context, 0)

    def forward(self, x, hint, timesteps, context, **kwargs):
        raise NotImplementedError("vanilla ControlNet (image hint through input_hint_block) is a comparison baseline "
                                  "of the paper, not part of the CtrLoRA hot path; use cldm.cldm_ctrlora_*")


class ControlLDM(LatentDiffusion):
    """cldm/cldm.py:308-438.  `control_scales` (13 floats), `only_mid_control`, `learning_rate`,
    `sd_locked` are plain attributes poked by callers, as in the reference."""

    def __init__(self, control_stage_config, control_key, only_mid_control, global_average_pooling=False,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.control_model = instantiate_from_config(control_stage_config)
        self.control_key = control_key
        self.only_mid_control = only_mid_control
        self.control_scales = [1.0] * 13
        self.global_average_pooling = global_average_pooling
        self.learning_rate = 1e-5
        self.sd_locked = True
        self.engine_dtype = None
        self.dp = None   # ctrlora_amd.parallel.GradAllReduce when running data parallel

    # ---- engine plumbing
    def set_engine_dtype(self, dtype):
        self.engine_dtype = dtype
        for m in (self.model.diffusion_model, self.control_model):
            m.engine_dtype = dtype
            m.invalidate_engine()
        if hasattr(self.first_stage_model, "invalidate_engine"):
            self.first_stage_model.engine_dtype = dtype
            self.first_stage_model.invalidate_engine()
        self.__dict__.pop("_engine", None)

    def _control_executors(self):
        return [self.control_model.executor()]

    def engine(self):
        """The composed executor; rebuilt whenever one of the network roots replaced its executor (dtype change,
        load_state_dict on the UNet, LoRA bank reloads)."""
        unet_ex, ctrls = self.model.diffusion_model.executor(), self._control_executors()
        eng = self.__dict__.get("_engine")
        if (eng is None or eng.unet is not unet_ex or len(eng.controls) != len(ctrls)
                or any(a is not b for a, b in zip(eng.controls, ctrls))):
            from ctrlora_amd.engine import CtrLoRAEngine
            eng = CtrLoRAEngine.from_executors(unet_ex, ctrls)
            self.__dict__["_engine"] = eng
        return eng

    def _sync_trainables(self):
        """Re-pack the engine's copies if an optimizer other than FusedAdamW touched the parameters."""
        from ctrlora_amd.train import trainables_version
        for ex, owner in zip(self.engine().controls, self._executor_owners()):
            bound = owner.__dict__.get("_bound")
            if bound:
                if bound[0].grad is None or bound[-1].grad is None:
                    # a foreign optimizer ran zero_grad(set_to_none=True): gradients were cleared -> re-attach
                    ex.tr.flat_grad.zero_()
                    for p, t in zip(bound, ex.tr.items):
                        p.grad = t.param_view(t.grad)
                ver = trainables_version(bound)
                if owner.__dict__.get("_bound_version") != ver:
                    ex.repack()
                    owner.__dict__["_bound_version"] = ver

    def _executor_owners(self):
        return [self.control_model]

    def _hint_latent(self, cond):
        """cldm_ctrlora_finetune.py:76-77: VAE-encode the condition image, sample the posterior, scale.
        A hint that is already a 4-channel latent (synthetic-latent benchmarks, cached encodings) passes through.

        Inside a `hint_cache()` scope (DDIMSampler.sample opens one) the encoder runs ONCE per condition image and
        its posterior (mean / std) is kept; every call still draws its own posterior sample, so the latent has the
        reference's distribution at every denoising step while the 1.1 TFLOP/image VAE encode -- half of the
        reference's DDIM FLOPs -- leaves the loop."""
        cc = cond["c_concat"]
        hint = cc[0] if len(cc) == 1 else torch.cat(cc, 1)
        if hint.shape[1] == self.channels:      # condition images have 3 channels, latents 4
            return hint
        cache = self.__dict__.get("_hint_cache")
        if cache is None:
            return self.get_first_stage_encoding(self.encode_first_stage(hint))
        key = tuple(id(t) for t in cond["c_concat"])
        hit = cache.get(key)
        if hit is None:
            # the entry keeps the source tensors alive, so an id() cannot be recycled within the scope
            hit = cache[key] = (list(cond["c_concat"]), self.encode_first_stage(hint))
        return self.get_first_stage_encoding(hit[1])

    @contextlib.contextmanager
    def hint_cache(self):
        """Scope in which condition images are constant (one sampling run): see `_hint_latent`."""
        outer = self.__dict__.get("_hint_cache")
        self.__dict__["_hint_cache"] = {} if outer is None else outer
        try:
            yield
        finally:
            if outer is None:
                self.__dict__.pop("_hint_cache", None)

    def _run(self, x_noisy, t, cond_txt, hints, weights=None):
        eng = self.engine()
        need_grad = torch.is_grad_enabled() and hints is not None and self.training
        if need_grad:
            self._sync_trainables()
            from ctrlora_amd.train import ApplyModelFn
            anchor = self._executor_owners()[0].__dict__["_bound"][0]
            return ApplyModelFn.apply(anchor, eng, x_noisy, t, cond_txt, hints, list(self.control_scales), weights,
                                      self.only_mid_control, None if self.dp is None else self.dp.on_backward_done)
        if hints is not None:
            self._sync_trainables()
        return eng.forward(x_noisy, t, cond_txt, hints, control_scales=list(self.control_scales), lora_weights=weights,
                           only_mid_control=self.only_mid_control)

    @torch.no_grad()
    def get_input(self, batch, k, bs=None, *args, **kwargs):
        x, c = super().get_input(batch, self.first_stage_key, *args, **kwargs)
        control = batch[self.control_key]
        if bs is not None:
            control = control[:bs]
        control = control.to(self.device).permute(0, 3, 1, 2).to(memory_format=torch.contiguous_format).float()
        return x, dict(c_crossattn=[c], c_concat=[control])

    @torch.no_grad()
    def get_unconditional_conditioning(self, N):
        return self.get_learned_conditioning([""] * N)

    @torch.no_grad()
    def log_images(self, batch, N=4, n_row=2, sample=False, ddim_steps=50, ddim_eta=0.0, unconditional_guidance_scale=9.0,
                   **kwargs):
        """What ImageLogger writes during training (cldm/cldm.py:359-411, cldm_ctrlora_pretrain.py:113-172): the VAE
        reconstruction of the target, the condition image, the caption, and DDIM samples with classifier-free guidance,
        all through the engine (sample_log -> DDIMSampler, decode_first_stage)."""
        from ldm.util import log_txt_as_img
        z, c = self.get_input(batch, self.first_stage_key, bs=N)
        N = min(z.shape[0], N)
        c_cat, c_txt = c["c_concat"][0][:N], c["c_crossattn"][0][:N]
        extra = {k: v for k, v in c.items() if k not in ("c_concat", "c_crossattn")}      # 'task' when pre-training
        log = {"reconstruction": self.decode_first_stage(z[:N]), "control": c_cat * 2.0 - 1.0,
               "conditioning": log_txt_as_img((512, 512), list(batch[self.cond_stage_key])[:N], size=16)}
        cond = dict(c_concat=[c_cat], c_crossattn=[c_txt], **extra)
        if sample:
            samples, _ = self.sample_log(cond=cond, batch_size=N, ddim=True, ddim_steps=ddim_steps, eta=ddim_eta)
            log["samples"] = self.decode_first_stage(samples)
        if unconditional_guidance_scale > 1.0:
            uc = dict(c_concat=[c_cat], c_crossattn=[self.get_unconditional_conditioning(N)], **extra)
            samples, _ = self.sample_log(cond=cond, batch_size=N, ddim=True, ddim_steps=ddim_steps, eta=ddim_eta,
                                         unconditional_guidance_scale=unconditional_guidance_scale,
                                         unconditional_conditioning=uc)
            log[f"samples_cfg_scale_{unconditional_guidance_scale:.2f}"] = self.decode_first_stage(samples)
        return log

    def apply_model(self, x_noisy, t, cond, *args, **kwargs):
        raise NotImplementedError("vanilla ControlLDM (image hint) is a comparison baseline; use the CtrLoRA LDMs")

    def configure_optimizers(self):
        raise NotImplementedError("use ControlFinetuneLDM / ControlPretrainLDM")
