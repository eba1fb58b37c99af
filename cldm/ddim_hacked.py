"""DDIM sampler (API of the reference's cldm/ddim_hacked.py: make_schedule :23-52, sample :55-120,
ddim_sampling :123-178, p_sample_ddim :181-231).

What changes underneath, not in the results:
  * the per-step coefficients live in ONE device table; the CFG combine + x0 / direction / noise update is a
    single fused HIP kernel (the reference issues ~5 elementwise launches and 4 host syncs per step);
  * classifier-free guidance evaluates the conditional and unconditional passes as one batch of 2B through the
    engine when both conditionings have the same structure (samples are independent, so the results are
    those of the reference's two sequential apply_model calls);
  * schedule tables are computed exactly as the reference does (fp64 numpy / fp32 torch on the host) --
    index and timestep bookkeeping is bit-exact.
"""
import contextlib

import numpy as np
import torch

from ldm.modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps, noise_like


def _cat_conds(c, u):
    """Batch two conditioning dicts (or lists of dicts) along dim 0; None if their structure differs."""
    if isinstance(c, dict) and isinstance(u, dict) and c.keys() == u.keys():
        out = {}
        for k in c:
            a, b = c[k], u[k]
            if isinstance(a, list) and isinstance(b, list) and len(a) == len(b) and all(
                    torch.is_tensor(x) and torch.is_tensor(y) and x.shape == y.shape for x, y in zip(a, b)):
                out[k] = [torch.cat([x, y], 0) for x, y in zip(a, b)]
            elif a is None and b is None:
                out[k] = None
            elif isinstance(a, str) and a == b:
                out[k] = a
            else:
                return None
        return out
    if isinstance(c, (list, tuple)) and isinstance(u, (list, tuple)) and len(c) == len(u):
        parts = [_cat_conds(a, b) for a, b in zip(c, u)]
        return None if any(p is None for p in parts) else parts
    return None


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.batch_cfg = True
        # capture one denoise step (both CFG passes + the fused update) as a hipGraph and replay it for the
        # rest of the loop; set False to run every step eagerly
        self.use_graph = True
        # opt-in: keep the captured denoise-step graph (and its static buffers) across sample() calls.  Valid while the call
        # is the same problem -- same conditioning TENSORS (identity and version), batch, shape, S, guidance scale -- and the
        # model's weights are unchanged; anything else re-captures.  A serving loop over one prompt / a benchmark sets it.
        self.reuse_graph = False
        self._graph_state = None
        self.graph_hits = 0
        # True (default): a condition IMAGE is VAE-encoded once per sample() call and only its posterior is re-sampled at
        # every apply_model (ControlLDM.hint_cache).  False: the reference's own schedule -- the encoder runs inside every
        # apply_model call (cldm_ctrlora_inference.py:165-172: 2 x 1.1 TFLOP per image and denoise step), for measurements
        # of the reference-faithful form.
        self.hoist_hint_encode = True

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor) and attr.device != self.model.device:
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        alphas_cumprod = self.model.alphas_cumprod
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        to_torch = lambda x: x.clone().detach().to(torch.float32).to(self.model.device)
        self.register_buffer("betas", to_torch(self.model.betas))
        self.register_buffer("alphas_cumprod", to_torch(alphas_cumprod))
        self.register_buffer("alphas_cumprod_prev", to_torch(self.model.alphas_cumprod_prev))
        ac = alphas_cumprod.cpu()
        self.register_buffer("sqrt_alphas_cumprod", to_torch(np.sqrt(ac)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", to_torch(np.sqrt(1. - ac)))
        ddim_sigmas, ddim_alphas, ddim_alphas_prev = make_ddim_sampling_parameters(
            alphacums=ac, ddim_timesteps=self.ddim_timesteps, eta=ddim_eta, verbose=verbose)
        self.register_buffer("ddim_sigmas", ddim_sigmas)
        self.register_buffer("ddim_alphas", ddim_alphas)
        self.register_buffer("ddim_alphas_prev", ddim_alphas_prev)
        self.register_buffer("ddim_sqrt_one_minus_alphas", np.sqrt(1. - ddim_alphas))
        # device table for the fused update kernel: rows = ddim index, cols = {a_t, a_prev, sigma_t, sqrt(1-a_t)};
        # each entry is the fp32 value torch.full(..., table[index]) produces in the reference (:203-211)
        cols = [torch.as_tensor(np.asarray(v, dtype=np.float64) if not torch.is_tensor(v) else v.double().cpu().numpy())
                for v in (ddim_alphas, ddim_alphas_prev, ddim_sigmas, np.sqrt(1. - ddim_alphas))]
        self.coef_table = torch.stack([c.to(torch.float32) for c in cols], dim=1).contiguous().to(self.model.device)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, dynamic_threshold=None, ucg_schedule=None, **kwargs):
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if verbose:
            print(f"Data shape for DDIM sampling is {size}, eta {eta}")
        # condition images do not change during a run: their VAE encode is done once (ControlLDM.hint_cache)
        scope = getattr(self.model, "hint_cache", None) if self.hoist_hint_encode else None
        with (scope() if callable(scope) else contextlib.nullcontext()):
            return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                      quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                      noise_dropout=noise_dropout, temperature=temperature,
                                      score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                      log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                      unconditional_conditioning=unconditional_conditioning,
                                      dynamic_threshold=dynamic_threshold, ucg_schedule=ucg_schedule)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, dynamic_threshold=None, ucg_schedule=None):
        if ddim_use_original_steps or quantize_denoised or score_corrector is not None or dynamic_threshold is not None:
            raise NotImplementedError("option not used by the CtrLoRA sampling scripts")
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device).float()
        if timesteps is None:
            timesteps = self.ddim_timesteps
        else:
            subset_end = int(min(timesteps / self.ddim_timesteps.shape[0], 1) * self.ddim_timesteps.shape[0]) - 1
            timesteps = self.ddim_timesteps[:subset_end]
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        if (self.use_graph and img.is_cuda and mask is None and ucg_schedule is None and total_steps >= 4
                and noise_dropout == 0. and callable(getattr(self.model, "engine", None))):
            return self._sampling_graphed(cond, img, timesteps, callback, img_callback, log_every_t, temperature,
                                          unconditional_guidance_scale, unconditional_conditioning, intermediates)
        eng = getattr(self.model, "engine", None)
        if callable(eng):
            # text context is constant over the loop: project K/V of every cross-attention once
            self.model.engine().cache_context_kv = True
            self.model.engine().reset_context_cache()
        try:
            for i, step in enumerate(time_range):
                index = total_steps - i - 1
                ts = torch.full((b,), int(step), device=device, dtype=torch.long)
                if mask is not None:
                    assert x0 is not None
                    img = self.model.q_sample(x0, ts) * mask + (1. - mask) * img
                if ucg_schedule is not None:
                    assert len(ucg_schedule) == len(time_range)
                    unconditional_guidance_scale = ucg_schedule[i]
                img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, temperature=temperature,
                                                  noise_dropout=noise_dropout,
                                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                                  unconditional_conditioning=unconditional_conditioning)
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(pred_x0, i)
                if index % log_every_t == 0 or index == total_steps - 1:
                    intermediates["x_inter"].append(img)
                    intermediates["pred_x0"].append(pred_x0)
        finally:
            if callable(eng):
                self.model.engine().cache_context_kv = False
                self.model.engine().reset_context_cache()
        return img, intermediates

    @torch.no_grad()
    def _sampling_graphed(self, cond, img, timesteps, callback, img_callback, log_every_t, temperature, cfg_scale,
                          uncond, intermediates):
        """The S-step loop of ddim_sampling (:157-178) with the loop state on the DEVICE: a cursor i counts
        iterations, `cl_ddim_set_t` derives ts = ddim_timesteps[S-1-i] and `cl_ddim_step_dev` the table row, so
        one step = both CFG passes through the engine + the fused update is a fixed kernel sequence.  Iteration 0
        runs eagerly (fills the context K/V cache, sizes every buffer), iteration 1 is captured, the rest replay."""
        from ctrlora_amd import hip
        model, device = self.model, img.device
        S, b = int(timesteps.shape[0]), img.shape[0]

        def _sig(c):
            if c is None:
                return None
            return tuple((k, tuple((id(t), t._version, tuple(t.shape)) for t in (v if isinstance(v, (list, tuple)) else [v])
                                   if torch.is_tensor(t))) for k, v in sorted(c.items()) if v is not None)

        def _plain(c):     # non-tensor entries of a conditioning dict (cond['task'], ...) are baked into the capture too
            if c is None:
                return None
            return tuple((k, repr(v)) for k, v in sorted(c.items())
                         if v is not None and not torch.is_tensor(v)
                         and not (isinstance(v, (list, tuple)) and any(torch.is_tensor(t) for t in v)))

        from ctrlora_amd.engine.nets import WEIGHTS_GENERATION
        # everything the captured step bakes in by VALUE: the residual scales (the UI's strength slider), the multi-LoRA
        # weights, only_mid_control, the CFG batching mode, and the generation of the packed weights (a re-pack, a reloaded
        # checkpoint, a bank switch or a replayed optimizer step makes the cached context K/V and folded LoRA copies stale)
        key = (id(model.engine()), S, tuple(img.shape), float(cfg_scale), float(temperature), _sig(cond), _sig(uncond),
               np.ascontiguousarray(timesteps).tobytes(),
               tuple(float(v) for v in getattr(model, "control_scales", ()) or ()),
               tuple(float(v) for v in (getattr(model, "lora_weights", None) or ())),
               bool(getattr(model, "only_mid_control", False)), bool(self.batch_cfg), bool(self.hoist_hint_encode), _plain(cond),
               _plain(uncond), WEIGHTS_GENERATION[0])
        st = self._graph_state if self.reuse_graph else None
        # sample() rebuilds the coefficient table on every call: the kept graph reads ITS table by address (held in st), so a
        # hit needs equal contents, not the same tensor
        if (st is not None and st["key"] == key and not (callback or img_callback)
                and st["coef"].shape == self.coef_table.shape and bool(torch.equal(st["coef"], self.coef_table))):
            # replay-only loop: same kernels, same buffers (the context K/V products of iteration 0 are still in st)
            st["x"].copy_(img.float())
            st["cursor"].zero_()
            for i in range(S):
                st["graph"].replay()
                index = S - i - 1
                if index % log_every_t == 0 or index == S - 1:
                    intermediates["x_inter"].append(st["x"].clone())
                    intermediates["pred_x0"].append(st["pred_x0"].clone())
            self.graph_hits += 1
            return st["x"].clone(), intermediates
        self._graph_state = None
        x = img.float().contiguous().clone()
        pred_x0 = torch.empty_like(x)
        ts = torch.zeros(b, dtype=torch.long, device=device)
        cursor = torch.zeros(1, dtype=torch.int32, device=device)
        table = torch.as_tensor(np.ascontiguousarray(timesteps).astype(np.int64)).to(device)
        use_cfg = not (uncond is None or cfg_scale == 1.)
        both = _cat_conds(cond, uncond) if (use_cfg and self.batch_cfg) else None
        any_sigma = bool((self.coef_table[:, 2] != 0).any())

        def body():
            hip.ddim_set_t(table, cursor, S, ts)
            e_u = None
            if not use_cfg:
                e_c = model.apply_model(x, ts, cond)
            elif both is not None:
                e = model.apply_model(torch.cat([x, x], 0), torch.cat([ts, ts], 0), both)
                e_c, e_u = e[:b], e[b:]
            else:
                e_c = model.apply_model(x, ts, cond)
                e_u = model.apply_model(x, ts, uncond)
            # the reference draws the sigma*noise term every step (:227); with eta = 0 every sigma is zero and
            # the draw cannot change the sample, so it is skipped (only the global RNG position differs)
            noise = noise_like(x.shape, device, False) * temperature if any_sigma else None
            hip.ddim_step_dev(x, e_c.float().contiguous(), None if e_u is None else e_u.float().contiguous(), noise,
                              self.coef_table, cursor, S, float(cfg_scale), x, pred_x0)
            hip.tick(cursor)

        eng = model.engine()
        eng.cache_context_kv = True
        eng.reset_context_cache()
        graph = None
        try:
            for i in range(S):
                index = S - i - 1
                if i == 0:
                    body()
                elif i == 1:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        body()
                    graph.replay()      # capture does not execute
                else:
                    graph.replay()
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(pred_x0, i)
                if index % log_every_t == 0 or index == S - 1:
                    intermediates["x_inter"].append(x.clone())
                    intermediates["pred_x0"].append(pred_x0.clone())
        finally:
            eng.cache_context_kv = False
            kv_keep = eng.detach_context_cache() if (self.reuse_graph and graph is not None) else None
            eng.reset_context_cache()
        out = x.clone()
        if self.reuse_graph and graph is not None:
            # the graph reads the cached K/V products and the conditioning tensors by address: hold them
            self._graph_state = dict(key=key, graph=graph, x=x, pred_x0=pred_x0, ts=ts, cursor=cursor, table=table, kv=kv_keep,
                                     conds=(cond, uncond, both), coef=self.coef_table,
                                     eng=eng)     # keeps id(engine) in `key` from being recycled by a rebuilt engine
        del graph
        return out, intermediates

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, dynamic_threshold=None):
        from ctrlora_amd import hip
        b, device = x.shape[0], x.device
        e_u = None
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
            e_c = self.model.apply_model(x, t, c)
        else:
            both = _cat_conds(c, unconditional_conditioning) if self.batch_cfg else None
            if both is not None:
                e = self.model.apply_model(torch.cat([x, x], 0), torch.cat([t, t], 0), both)
                e_c, e_u = e[:b].contiguous(), e[b:].contiguous()
            else:
                e_c = self.model.apply_model(x, t, c)
                e_u = self.model.apply_model(x, t, unconditional_conditioning)
        if self.model.parameterization != "eps":
            raise NotImplementedError("v-parameterisation is not used by the CtrLoRA configs")
        sigma_nonzero = float(self.coef_table[index, 2]) != 0.0 if noise_dropout > 0. else True
        # the reference draws randn for the sigma*noise term every step, also when sigma == 0 (:227)
        noise = noise_like(x.shape, device, repeat_noise) * temperature
        if noise_dropout > 0. and sigma_nonzero:
            noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        x = x.float().contiguous()
        x_prev, pred_x0 = torch.empty_like(x), torch.empty_like(x)
        hip.ddim_step(x, e_c.float().contiguous(), None if e_u is None else e_u.float().contiguous(), noise.contiguous(),
                      self.coef_table, index, float(unconditional_guidance_scale), x_prev, pred_x0)
        return x_prev, pred_x0

    # ------------------------------------------------------------------ inversion / img2img helpers
    def _eps_pair(self, x, t, c, scale, uc):
        """(eps_cond, eps_uncond or None) for one inversion step.  With guidance the reference evaluates ONE batch of 2B
        ordered [unconditional; conditional] (cldm/ddim_hacked.py:257-262); tensors and same-structure dict conditionings
        are batched that way here, anything else runs as two passes (the samples are independent)."""
        if scale == 1.:
            return self.model.apply_model(x, t, c), None
        assert uc is not None
        b = x.shape[0]
        both = torch.cat((uc, c)) if torch.is_tensor(c) and torch.is_tensor(uc) else _cat_conds(uc, c)
        if both is not None:
            e = self.model.apply_model(torch.cat((x, x)), torch.cat((t, t)), both)
            return e[b:].contiguous(), e[:b].contiguous()
        return self.model.apply_model(x, t, c), self.model.apply_model(x, t, uc)

    @torch.no_grad()
    def encode(self, x0, c, t_enc, use_original_steps=False, return_intermediates=None,
               unconditional_guidance_scale=1.0, unconditional_conditioning=None, callback=None):
        """Deterministic DDIM inversion x0 -> x_{t_enc} (reference cldm/ddim_hacked.py:234-279):
            x_next = sqrt(a_next / a) x + sqrt(a_next) (sqrt(1 / a_next - 1) - sqrt(1 / a - 1)) eps.
        This is the sampler's own update with the roles of the two alphas exchanged and sigma = 0
        (sqrt(a_next) (x - sqrt(1 - a) eps) / sqrt(a) + sqrt(1 - a_next) eps), so every step is ONE launch of the fused
        guidance + update kernel on a table {a, a_next, 0, sqrt(1 - a)} instead of the reference's six elementwise
        launches; the two forms differ by fp32 rounding only (tests: <= 2e-6 per step against the oracle's restatement of
        the reference arithmetic).  Returns (x_encoded, {'x_encoded', 'intermediate_steps'[, 'intermediates']})."""
        from ctrlora_amd import hip
        steps_all = np.arange(self.ddpm_num_timesteps) if use_original_steps else self.ddim_timesteps
        assert t_enc <= steps_all.shape[0]
        n = int(t_enc)
        if use_original_steps:
            a_next, a_cur = self.alphas_cumprod[:n], self.alphas_cumprod_prev[:n]
        else:
            a_next, a_cur = self.ddim_alphas[:n], self.ddim_alphas_prev[:n]
        as64 = lambda v: torch.as_tensor(np.asarray(v.detach().cpu() if torch.is_tensor(v) else v, dtype=np.float64))
        a_next, a_cur = as64(a_next), as64(a_cur)
        table = torch.stack([a_cur, a_next, torch.zeros(n, dtype=torch.float64), torch.sqrt(1. - a_cur)], dim=1)
        table = table.to(torch.float32).contiguous().to(x0.device)
        x = x0.float().contiguous()
        b = x.shape[0]
        kept, kept_at = [], []
        every = (n // return_intermediates) if return_intermediates else 0
        for i in range(n):
            t = torch.full((b,), int(steps_all[i]), device=self.model.device, dtype=torch.long)
            e_c, e_u = self._eps_pair(x, t, c, unconditional_guidance_scale, unconditional_conditioning)
            x_new = torch.empty_like(x)
            hip.ddim_step(x, e_c.float().contiguous(), None if e_u is None else e_u.float().contiguous(), None, table, i,
                          float(unconditional_guidance_scale), x_new, None)
            x = x_new
            if return_intermediates and ((i % every == 0 and i < n - 1) or i >= n - 2):
                kept.append(x)
                kept_at.append(i)
            if callback:
                callback(i)
        out = {"x_encoded": x, "intermediate_steps": kept_at}
        if return_intermediates:
            out.update({"intermediates": kept})
        return x, out

    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """q(x_t | x0) on the DDIM (or, use_original_steps, the DDPM) tables (reference cldm/ddim_hacked.py:282-295);
        t indexes the table.  The forward-process kernel of the training path (cl_qsample) does the arithmetic."""
        if use_original_steps:
            sa, s1m = self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod
        else:
            # (square roots on the host: correctly rounded, the values the reference's tables hold on CPU)
            sa, s1m = torch.sqrt(self.ddim_alphas.cpu()), self.ddim_sqrt_one_minus_alphas
        if noise is None:
            noise = torch.randn_like(x0)
        if not x0.is_cuda:
            from ldm.modules.diffusionmodules.util import extract_into_tensor
            as_t = lambda v: v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))
            return (extract_into_tensor(as_t(sa), t, x0.shape) * x0 + extract_into_tensor(as_t(s1m), t, x0.shape) * noise)
        from ctrlora_amd import hip
        dev32 = lambda v: torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).to(x0.device, torch.float32).contiguous()
        out = torch.empty_like(x0, dtype=torch.float32)
        return hip.qsample(x0.float().contiguous(), noise.float().contiguous(), t.long().contiguous().to(x0.device),
                           dev32(sa), dev32(s1m), out)

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, callback=None):
        """The last t_start steps of the sampler from a given latent (reference cldm/ddim_hacked.py:298-317)."""
        if use_original_steps:
            raise NotImplementedError("option not used by the CtrLoRA sampling scripts")
        timesteps = self.ddim_timesteps[:t_start]
        total_steps = timesteps.shape[0]
        print(f"Running DDIM Sampling with {total_steps} timesteps")
        x_dec = x_latent
        for i, step in enumerate(np.flip(timesteps)):
            index = total_steps - i - 1
            ts = torch.full((x_latent.shape[0],), int(step), device=x_latent.device, dtype=torch.long)
            x_dec, _ = self.p_sample_ddim(x_dec, cond, ts, index=index,
                                          unconditional_guidance_scale=unconditional_guidance_scale,
                                          unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
        return x_dec
