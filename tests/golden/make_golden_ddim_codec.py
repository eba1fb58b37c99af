"""Golden vectors for DDIMSampler.encode / stochastic_encode / decode from the UNMODIFIED reference
(cldm/ddim_hacked.py:234-317; build container only).

    python tests/golden/make_golden_ddim_codec.py        # writes tests/golden/ddim_codec.pt

The sampler is driven by the same analytic eps model as schedule.pt's `ddim_traj` block, with the conditioning a per-sample
scale TENSOR (the reference's guided `encode` batches `torch.cat((unconditional_conditioning, c))`, so its conditionings
must be tensors): eps(x, t, c) = tanh(0.7 x + 0.001 t) * c[:, None, None, None], c = 1.0 (conditional) / 0.6 (unconditional).
The fixture holds inputs and the reference's outputs only.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
if HERE not in sys.path:
    sys.path.insert(0, HERE)
import make_golden as G   # noqa: E402  (helpers only: use_reference_packages / install_stubs)


def eps_model(x, t, c):
    return torch.tanh(0.7 * x + 0.001 * t.float().view(-1, 1, 1, 1)) * c.view(-1, 1, 1, 1)


def main():
    G.install_stubs()
    G.use_reference_packages()
    from ldm.models.diffusion.ddpm import DDPM
    from cldm.ddim_hacked import DDIMSampler
    m = DDPM.__new__(DDPM)
    nn.Module.__init__(m)
    m.v_posterior = 0.0
    m.parameterization = "eps"
    m.register_schedule(beta_schedule="linear", timesteps=1000, linear_start=0.00085, linear_end=0.0120)

    class Stub:
        num_timesteps = 1000
        device = torch.device("cpu")
        parameterization = "eps"

    stub = Stub()
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev"):
        setattr(stub, k, getattr(m, k).clone())
    stub.apply_model = eps_model
    DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)

    g = torch.Generator().manual_seed(11)
    B = 2
    x0 = torch.randn(B, 4, 8, 8, generator=g)
    c, uc = torch.full((B,), 1.0), torch.full((B,), 0.6)
    out = dict(x0=x0.clone(), c=c.clone(), uc=uc.clone(), encode={}, decode={}, stochastic={})
    quiet = lambda: contextlib.ExitStack()

    def run(fn, *a, **k):
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            return fn(*a, **k)

    # encode: (S, t_enc, guidance scale, return_intermediates)
    for S, t_enc, scale, ri in ((50, 50, 1.0, None), (50, 30, 7.5, 5), (20, 12, 3.0, 4), (10, 10, 1.0, 10)):
        s = DDIMSampler(stub)
        s.make_schedule(S, ddim_eta=0.0, verbose=False)
        x_enc, info = run(s.encode, x0.clone(), c, t_enc, return_intermediates=ri, unconditional_guidance_scale=scale,
                          unconditional_conditioning=uc if scale != 1.0 else None)
        rec = dict(x_encoded=x_enc.clone(), intermediate_steps=list(info["intermediate_steps"]))
        if ri:
            rec["intermediates"] = [t.clone() for t in info["intermediates"]]
        out["encode"][f"S{S}_t{t_enc}_cfg{scale}_ri{ri}"] = rec
    # encode on the DDPM tables (use_original_steps)
    s = DDIMSampler(stub)
    s.make_schedule(50, ddim_eta=0.0, verbose=False)
    x_enc, info = run(s.encode, x0.clone(), c, 40, use_original_steps=True)
    out["encode"]["orig_t40_cfg1.0"] = dict(x_encoded=x_enc.clone(), intermediate_steps=list(info["intermediate_steps"]))

    # decode: (S, eta, t_start, scale); noise_like draws from the default generator each step
    for S, eta, t_start, scale in ((50, 0.0, 50, 7.5), (50, 0.0, 20, 1.0), (10, 0.5, 6, 3.0)):
        s = DDIMSampler(stub)
        s.make_schedule(S, ddim_eta=eta, verbose=False)
        torch.manual_seed(321)
        x_dec = run(s.decode, x0.clone(), c, t_start, unconditional_guidance_scale=scale, unconditional_conditioning=uc)
        out["decode"][f"S{S}_eta{eta}_t{t_start}_cfg{scale}"] = dict(x_dec=x_dec.clone())

    # encode then decode (eta 0, no guidance) returns near x0: recorded as a reference-side property
    s = DDIMSampler(stub)
    s.make_schedule(50, ddim_eta=0.0, verbose=False)
    x_enc, _ = run(s.encode, x0.clone(), c, 50)
    torch.manual_seed(321)
    x_back = run(s.decode, x_enc, c, 50)
    out["roundtrip_S50"] = dict(x_back=x_back.clone(), rel=float((x_back - x0).norm() / x0.norm()))

    # stochastic_encode: t indexes the table
    noise = torch.randn(B, 4, 8, 8, generator=g)
    for S, idx, orig in ((50, [0, 49], False), (20, [7, 13], False), (50, [3, 999], True)):
        s = DDIMSampler(stub)
        s.make_schedule(S, ddim_eta=0.0, verbose=False)
        t = torch.tensor(idx, dtype=torch.long)
        y = s.stochastic_encode(x0.clone(), t, use_original_steps=orig, noise=noise.clone())
        out["stochastic"][f"S{S}_orig{int(orig)}"] = dict(t=t.clone(), y=y.clone())
    out["noise"] = noise.clone()
    torch.save(out, f"{HERE}/ddim_codec.pt")
    print("[golden] ddim_codec.pt written; encode keys", list(out["encode"]), "round trip rel", out["roundtrip_S50"]["rel"])


if __name__ == "__main__":
    main()
