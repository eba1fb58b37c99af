"""Generate golden vectors from the UNMODIFIED reference (run in the build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.json, *.pt

/root/reference cannot travel to the GPU box, so its outputs on key-addressed
weights (oracle/arch.py:draw_param) and seeded inputs are committed here as small
fixtures.  Nothing under /root/reference is modified; the four third-party
packages it imports but this image lacks (pytorch_lightning, torchvision,
omegaconf, open_clip) are replaced by inert stubs in sys.modules (SURVEY.md
Appendix C).
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True


def use_reference_packages():
    """Make `import cldm / ldm / datasets` resolve to the REFERENCE and `import oracle` to this repo.

    The reference's cldm/ and ldm/ have no __init__.py (namespace packages); this repo's mirror has regular
    packages of the same names, and a regular package ANYWHERE on sys.path beats a namespace portion found
    earlier.  So the repo root must not be on sys.path at all while the reference is imported: it is removed
    (with '' / cwd entries pointing at it), already-imported mirror modules are purged, and `oracle` -- the only
    repo package the generators need -- is registered by file location."""
    import importlib.util
    root = os.path.realpath(ROOT)
    sys.path[:] = [p for p in sys.path if os.path.realpath(p or os.getcwd()) != root]
    for name in [m for m in sys.modules if m.split(".")[0] in ("cldm", "ldm", "datasets", "scripts", "api")]:
        del sys.modules[name]
    sys.path.insert(0, REF)
    if "oracle" not in sys.modules:
        spec = importlib.util.spec_from_file_location("oracle", os.path.join(ROOT, "oracle", "__init__.py"),
                                                      submodule_search_locations=[os.path.join(ROOT, "oracle")])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["oracle"] = mod
        spec.loader.exec_module(mod)
    import cldm
    import ldm
    for pkg in (cldm, ldm):
        assert all(os.path.realpath(p).startswith(REF) for p in pkg.__path__), (pkg.__name__, list(pkg.__path__))


def install_stubs():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
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

        def log_dict(self, *a, **k):
            pass

    class Callback:
        pass

    pl.LightningModule = LightningModule
    pl.Callback = Callback
    pl.Trainer = object
    cbs = types.ModuleType("pytorch_lightning.callbacks")
    cbs.Callback = Callback
    ut = types.ModuleType("pytorch_lightning.utilities")
    utd = types.ModuleType("pytorch_lightning.utilities.distributed")
    utd.rank_zero_only = lambda f: f
    ut.distributed = utd
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda *a, **k: None
    tv.utils = tvu
    oc = types.ModuleType("omegaconf")

    class ListConfig(list):
        pass

    oc.ListConfig = ListConfig
    oc.OmegaConf = object
    ocl = types.ModuleType("omegaconf.listconfig")
    ocl.ListConfig = ListConfig
    sys.modules.update({
        "pytorch_lightning": pl, "pytorch_lightning.callbacks": cbs, "pytorch_lightning.utilities": ut,
        "pytorch_lightning.utilities.distributed": utd, "torchvision": tv, "torchvision.utils": tvu,
        "omegaconf": oc, "omegaconf.listconfig": ocl, "open_clip": types.ModuleType("open_clip"),
    })


def ref_kwargs(cfg, control: bool):
    kw = dict(image_size=32, in_channels=cfg.in_channels, model_channels=cfg.model_channels,
              attention_resolutions=list(cfg.attention_resolutions), num_res_blocks=cfg.num_res_blocks,
              channel_mult=list(cfg.channel_mult), num_heads=cfg.num_heads, use_spatial_transformer=True,
              transformer_depth=1, context_dim=cfg.context_dim, use_checkpoint=True, legacy=False)
    if control:
        kw.update(hint_channels=3, ft_with_lora=True, lora_rank=cfg.lora_rank, norm_trainable=True)
    else:
        kw.update(out_channels=cfg.out_channels)
    return kw


def build_ldm(cfg):
    """The real ControlFinetuneLDM, with VAE / CLIP swapped for Identity and the VAE-encode of
    the hint bypassed (hint latents are fed directly -- SURVEY.md Appendix B item 18)."""
    from ldm.util import instantiate_from_config
    conf = dict(target="cldm.cldm_ctrlora_finetune.ControlFinetuneLDM", params=dict(
        linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000,
        first_stage_key="jpg", cond_stage_key="txt", control_key="hint", image_size=64, channels=4,
        cond_stage_trainable=False, conditioning_key="crossattn", monitor="val/loss_simple_ema",
        scale_factor=0.18215, use_ema=False, only_mid_control=False,
        control_stage_config=dict(target="cldm.cldm_ctrlora_finetune.ControlNetFinetune", params=ref_kwargs(cfg, True)),
        unet_config=dict(target="cldm.cldm.ControlledUnetModel", params=ref_kwargs(cfg, False)),
        first_stage_config=dict(target="torch.nn.Identity"),
        cond_stage_config=dict(target="torch.nn.Identity")))
    model = instantiate_from_config(conf)
    model.encode_first_stage = lambda x: x
    model.get_first_stage_encoding = lambda z: z
    return model


def digest(t: torch.Tensor, n=16):
    f = t.detach().float().flatten()
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return dict(shape=list(t.shape), sum=float(f.double().sum()), l2=float(f.double().norm()),
                idx=idx.tolist(), vals=f[idx].tolist())


def sampled(t: torch.Tensor, n=256):
    """Fixture form of a big gradient tensor: n evenly spaced entries (as a tensor) + l2 norm + sum."""
    f = t.detach().float().flatten()
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return dict(shape=list(t.shape), sum=float(f.double().sum()), l2=float(f.double().norm()), idx=idx,
                vals=f[idx].clone())


def inputs_for(cfg, B, H, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(
        z=torch.randn(B, 4, H, H, generator=g), hint_z=torch.randn(B, 4, H, H, generator=g) * 0.18215 * 5,
        ctx=torch.randn(B, 77, cfg.context_dim, generator=g), noise=torch.randn(B, 4, H, H, generator=g),
        t=torch.randint(0, 1000, (B,), generator=g).long())


def gen_model_golden(name, cfg, B, H, seed, full_tensors, sampled_grads=False):
    from oracle import arch
    torch.manual_seed(0)
    model = build_ldm(cfg)
    cn_shapes, un_shapes = arch.controlnet_shapes(cfg), arch.unet_shapes(cfg)
    # --- key / shape parity with the real reference modules
    ref_cn = {k: list(v.shape) for k, v in model.control_model.state_dict().items()}
    ref_un = {k: list(v.shape) for k, v in model.model.diffusion_model.state_dict().items()}
    if not sampled_grads:      # (the 64x64 fixture shares keys_sd15.json: same architecture)
        json.dump(dict(controlnet=ref_cn, unet=ref_un), open(f"{HERE}/keys_{name}.json", "w"))
    assert ref_cn == {k: list(v) for k, v in cn_shapes.items()}, "controlnet key/shape mismatch"
    assert ref_un == {k: list(v) for k, v in un_shapes.items()}, "unet key/shape mismatch"
    model.control_model.load_state_dict(arch.make_state(cn_shapes, seed), strict=True)
    model.model.diffusion_model.load_state_dict(arch.make_state(un_shapes, seed), strict=True)
    model.train()
    model.learning_rate = 1e-4
    inp = inputs_for(cfg, B, H, seed)
    cond = dict(c_crossattn=[inp["ctx"]], c_concat=[inp["hint_z"]])
    out = dict(meta=dict(name=name, B=B, H=H, seed=seed, cfg=cfg.__dict__))
    # --- ControlNet residuals and eps through the real apply_model
    with torch.no_grad():
        control = model.control_model(hint=inp["hint_z"], timesteps=inp["t"], context=inp["ctx"])
    out["control_digest"] = [digest(c) for c in control]
    # --- p_losses + backward + the reference's optimizer selection + one AdamW step
    os.makedirs("./tmp", exist_ok=True)
    opt = model.configure_optimizers()
    loss, ld = model.p_losses(inp["z"], cond, inp["t"], noise=inp["noise"])
    with torch.no_grad():
        x_noisy = model.q_sample(inp["z"], inp["t"], inp["noise"])
        eps = model.apply_model(x_noisy, inp["t"], cond)
    out["loss"] = float(loss)
    out["x_noisy"] = x_noisy.clone()
    out["eps"] = eps.clone()
    loss.backward()
    names = {id(p): n for n, p in model.control_model.named_parameters()}
    tr = [names[id(p)] for p in opt.param_groups[0]["params"]]
    out["trainable_names"] = tr
    out["grad_digest"] = {n: digest(dict(model.control_model.named_parameters())[n].grad) for n in tr}
    if sampled_grads:   # the benchmarked shapes: every trainable gradient as 256 sampled entries + norm + sum
        out["grad_sampled"] = {n: sampled(dict(model.control_model.named_parameters())[n].grad) for n in tr}
    opt.step()
    out["adamw_digest"] = {n: digest(dict(model.control_model.named_parameters())[n]) for n in tr[:: max(1, len(tr) // 24)]}
    if full_tensors:
        out["control"] = [c.clone() for c in control]
        pick = [n for n in tr if ("input_blocks.1.1" in n or "middle_block.1" in n or "zero_convs.0." in n
                                  or "zero_convs.5." in n or "time_embed" in n or "input_blocks.4.0.emb" in n)]
        out["grads"] = {n: dict(model.control_model.named_parameters())[n].grad.clone() for n in pick}
    torch.save(out, f"{HERE}/model_{name}.pt")
    print(f"[golden] {name}: loss={out['loss']:.6f}  trainables={len(tr)}  eps_l2={float(eps.norm()):.4f}")
    return model


def gen_schedule_golden():
    from ldm.models.diffusion.ddpm import DDPM
    from cldm.ddim_hacked import DDIMSampler
    m = DDPM.__new__(DDPM)
    nn.Module.__init__(m)
    m.v_posterior = 0.0
    m.parameterization = "eps"
    m.register_schedule(beta_schedule="linear", timesteps=1000, linear_start=0.00085, linear_end=0.0120)
    out = dict(ddpm={k: getattr(m, k).clone() for k in
                     ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                      "sqrt_one_minus_alphas_cumprod")})

    class Stub:
        num_timesteps = 1000
        device = torch.device("cpu")
        parameterization = "eps"

    stub = Stub()
    for k, v in out["ddpm"].items():
        setattr(stub, k, v)
    DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)
    out["ddim"] = {}
    for S, eta in ((50, 0.0), (20, 0.0), (10, 0.5), (4, 1.0)):
        s = DDIMSampler(stub)
        s.make_schedule(S, ddim_eta=eta, verbose=False)
        out["ddim"][f"S{S}_eta{eta}"] = dict(
            timesteps=np.asarray(s.ddim_timesteps).copy(), alphas=s.ddim_alphas.clone(),
            alphas_prev=np.asarray(s.ddim_alphas_prev).copy(), sigmas=np.asarray(s.ddim_sigmas).copy(),
            sqrt_one_minus_alphas=np.asarray(s.ddim_sqrt_one_minus_alphas).copy())
    # sampler arithmetic on an analytic eps model, with and without CFG, eta 0 and > 0
    g = torch.Generator().manual_seed(7)
    x_T = torch.randn(2, 4, 8, 8, generator=g)

    def eps_model(x, t, c):
        return torch.tanh(0.7 * x + 0.001 * t.float().view(-1, 1, 1, 1)) * (1.0 if c == "c" else 0.6)

    stub.apply_model = eps_model
    out["ddim_traj"] = {}
    for S, eta, scale in ((50, 0.0, 7.5), (10, 0.5, 3.0), (5, 0.0, 1.0)):
        torch.manual_seed(123)     # noise_like draws from the default generator each step
        s = DDIMSampler(stub)
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            samples, inter = s.sample(S, 2, (4, 8, 8), "c", verbose=False, eta=eta, x_T=x_T.clone(),
                                      unconditional_guidance_scale=scale, unconditional_conditioning="u")
        out["ddim_traj"][f"S{S}_eta{eta}_cfg{scale}"] = dict(samples=samples.clone(), x_T=x_T.clone())
    torch.save(out, f"{HERE}/schedule.pt")
    print("[golden] schedule.pt written; ac[0]=%.16f ac[999]=%.18f" %
          (float(out["ddpm"]["alphas_cumprod"][0]), float(out["ddpm"]["alphas_cumprod"][999])))


def gen_lora_golden():
    """cldm.lora unit vectors: forward, _fuse_lora equivalence (cldm/lora.py:237-291)."""
    from cldm.lora import LoRALinearLayer, LoRACompatibleLinear
    torch.manual_seed(3)
    lin = LoRACompatibleLinear(96, 64, lora_layer=LoRALinearLayer(96, 64, rank=32))
    nn.init.normal_(lin.lora_layer.up.weight, std=0.05)
    x = torch.randn(5, 7, 96)
    y = lin(x)
    sd = {k: v.clone() for k, v in lin.state_dict().items()}
    lin._fuse_lora()
    yf = lin(x)
    torch.save(dict(state=sd, x=x, y=y.detach(), y_fused=yf.detach(), w_fused=lin.weight.detach().clone()),
               f"{HERE}/lora.pt")
    print("[golden] lora.pt written; fuse rel err %.2e" % float((y - yf).norm() / y.norm()))


if __name__ == "__main__":
    assert os.path.isdir(REF), "run in the build container (needs /root/reference)"
    install_stubs()
    os.chdir("/tmp")
    use_reference_packages()
    from oracle import arch
    if "--only-sd15-64" in sys.argv:
        # BASELINE config-2 shapes (latent 64x64: N = 4096 attention, M = 8192 GEMM tiles, split-K levels); B = 2
        gen_model_golden("sd15_64", arch.SD15, B=2, H=64, seed=7, full_tensors=False, sampled_grads=True)
        sys.exit(0)
    if "--only-sd15-64-r32" in sys.argv:
        # BASELINE configs[0] at its own shape: configs/ctrlora_finetune_sd15_rank32.yaml (rank 32), bs 1, 512x512 -> latent 64x64
        from dataclasses import replace
        gen_model_golden("sd15_64_r32", replace(arch.SD15, lora_rank=32), B=1, H=64, seed=9, full_tensors=False, sampled_grads=True)
        sys.exit(0)
    gen_lora_golden()
    gen_schedule_golden()
    gen_model_golden("tiny", arch.TINY, B=2, H=16, seed=11, full_tensors=True)
    if "--no-sd15" not in sys.argv:
        gen_model_golden("sd15", arch.SD15, B=1, H=16, seed=5, full_tensors=False)
