"""GPU bring-up check: HIP engine vs the CPU oracle (prints errors, asserts nothing)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import arch, ref_model as R
from ctrlora_amd.engine import CtrLoRAEngine, NetCfg

def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

def netcfg(c):
    return NetCfg(c.in_channels, c.out_channels, c.model_channels, c.channel_mult, c.num_res_blocks,
                  c.attention_resolutions, c.num_heads, c.context_dim)

def run(name, cfg, B, H, seed, dtype, do_bwd=True):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(B, 4, H, H, generator=g); hint = torch.randn(B, 4, H, H, generator=g)
    ctx = torch.randn(B, 77, cfg.context_dim, generator=g); noise = torch.randn(B, 4, H, H, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g).long()
    sd_cn = arch.make_state(arch.controlnet_shapes(cfg), seed); sd_un = arch.make_state(arch.unet_shapes(cfg), seed)
    t0 = time.time()
    eng = CtrLoRAEngine(sd_un, [sd_cn], netcfg(cfg), dtype=dtype, device="cuda")
    torch.cuda.synchronize(); print(f"[{name}] engine built in {time.time()-t0:.1f}s")
    dev = lambda x: x.cuda()
    # --- oracle
    tr = [k for k in sd_cn if arch.is_trainable(k)]
    for k in tr: sd_cn[k].requires_grad_(True)
    t0 = time.time()
    ctrl_ref = R.controlnet_forward(sd_cn, cfg, hint, t, ctx)
    eps_ref = R.apply_model(sd_cn, sd_un, cfg, z, t, ctx, hint)
    loss_ref = ((eps_ref - noise) ** 2).mean()
    if do_bwd: loss_ref.backward()
    print(f"[{name}] oracle fwd+bwd {time.time()-t0:.1f}s loss={float(loss_ref):.6f}")
    # --- engine
    outs = eng.control_outputs(dev(hint), dev(t), dev(ctx))
    for k, (o, r) in enumerate(zip(outs, ctrl_ref)):
        print(f"[{name}] control[{k}] rel={rel(o, r):.3e} shape={tuple(o.shape)}")
    eps = eng.forward(dev(z), dev(t), dev(ctx), [dev(hint)], record=do_bwd)
    print(f"[{name}] eps rel={rel(eps, eps_ref):.3e}")
    eps_plain = eng.forward(dev(z), dev(t), dev(ctx), None)
    print(f"[{name}] eps(no control) rel={rel(eps_plain, R.unet_forward(sd_un, cfg, z, t, ctx)):.3e}")
    if do_bwd:
        eng.zero_grad()
        d_eps = 2.0 * (eps - dev(noise)) / eps.numel()
        eng.backward(d_eps)
        torch.cuda.synchronize()
        worst = []
        for tt in eng.controls[0].tr.items:
            e = rel(tt.grad, sd_cn[tt.name].grad)
            worst.append((e, tt.name))
        worst.sort(reverse=True)
        print(f"[{name}] grads: n={len(worst)} max rel={worst[0][0]:.3e} median={worst[len(worst)//2][0]:.3e}")
        for e, n in worst[:12]: print(f"    {e:.3e} {n}")
    del eng; torch.cuda.empty_cache()

if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny32", "tiny16"]
    if "tiny32" in which: run("tiny f32", arch.TINY, 2, 16, 11, torch.float32)
    if "tiny16" in which: run("tiny bf16", arch.TINY, 2, 16, 11, torch.bfloat16)
    if "sd32" in which: run("sd15 f32", arch.SD15, 1, 16, 5, torch.float32)
    if "sd16" in which: run("sd15 bf16", arch.SD15, 1, 16, 5, torch.bfloat16)
