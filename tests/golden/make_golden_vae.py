"""Golden vectors for the first-stage (AutoencoderKL) path -- SURVEY.md 8(f1) -- from the UNMODIFIED reference.

    python tests/golden/make_golden_vae.py        # writes tests/golden/vae.pt  (build container only)

The reference's ldm.models.autoencoder.AutoencoderKL (SD ddconfig: ch 128, ch_mult 1-2-4-4, 2 res blocks, middle
attention) is built with key-addressed weights (oracle.arch.draw_param: every value a function of the state-dict key,
so the test rebuilds them without the reference), fed a deterministic synthetic image (`test_image`) and a seeded
latent; the fixture holds the encoder moments in full (1 x 8 x 32 x 32) and 4096 sampled pixels + norm of the
decoded image.  Image size 256 x 256 keeps the CPU run and the fixture small; the GPU test also runs 512 x 512
against the plain-torch modules of this repo's mirror, which this fixture pins (same keys, same arithmetic).
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

DDCONFIG = dict(attn_resolutions=[], ch=128, ch_mult=[1, 2, 4, 4], double_z=True, dropout=0.0, in_channels=3,
                num_res_blocks=2, out_ch=3, resolution=256, z_channels=4)
SEED = 17


def test_image(B, H, W):
    """Smooth + edgy synthetic image in [-1, 1] (a stand-in for a condition image: flat regions and sharp edges)."""
    y = torch.linspace(-1, 1, H).view(1, 1, H, 1)
    x = torch.linspace(-1, 1, W).view(1, 1, 1, W)
    b = torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1)
    base = torch.cat([torch.sin(3.1 * x + 2.0 * y + b), torch.cos(5.3 * x * y - b), torch.sin(7.0 * (x * x + y * y)).expand(B, -1, -1, -1)], 1)
    edges = ((torch.sin(9.0 * x + b) * torch.cos(11.0 * y) > 0.3).float() * 2 - 1)
    return (0.6 * base + 0.4 * edges).clamp(-1, 1).contiguous()


def vae_state(module, seed=SEED):
    from oracle import arch
    return {k: arch.draw_param("first_stage." + k, tuple(v.shape), seed) for k, v in module.state_dict().items()}


def sampled(t, n=4096):
    f = t.detach().float().flatten()
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return dict(shape=list(t.shape), l2=float(f.double().norm()), sum=float(f.double().sum()), idx=idx, vals=f[idx].clone())


if __name__ == "__main__":
    import make_golden as mg
    assert os.path.isdir(mg.REF)
    mg.install_stubs()
    os.chdir("/tmp")
    mg.use_reference_packages()
    sys.modules.setdefault("xformers", None)
    from ldm.models.autoencoder import AutoencoderKL
    assert AutoencoderKL.__module__ and os.path.realpath(sys.modules[AutoencoderKL.__module__].__file__).startswith(mg.REF)
    torch.manual_seed(0)
    m = AutoencoderKL(ddconfig=DDCONFIG, lossconfig=dict(target="torch.nn.Identity"), embed_dim=4).eval()
    m.load_state_dict(vae_state(m), strict=True)
    keys = {k: list(v.shape) for k, v in m.state_dict().items()}
    x = test_image(1, 256, 256)
    g = torch.Generator().manual_seed(SEED)
    z = torch.randn(1, 4, 32, 32, generator=g)
    with torch.no_grad():
        post = m.encode(x)
        dec = m.decode(z)
    out = dict(meta=dict(seed=SEED, ddconfig=DDCONFIG, H=256), keys=keys, moments=post.parameters.clone(),
               mean_l2=float(post.mean.norm()), decoded=sampled(dec))
    torch.save(out, os.path.join(HERE, "vae.pt"))
    print("[golden] vae.pt: moments l2 %.4f, decoded l2 %.4f, %d keys" % (float(post.parameters.norm()), out["decoded"]["l2"], len(keys)))
