"""GPU parity of DDIMSampler.encode / stochastic_encode / decode (cldm/ddim_hacked.py, mirroring the reference's
cldm/ddim_hacked.py:234-317) against outputs of the UNMODIFIED reference sampler (tests/golden/ddim_codec.pt, written by
tests/golden/make_golden_ddim_codec.py) and against the oracle's restatement.  The sampler is driven by the fixture's
analytic eps model, so what is compared is the sampler's own arithmetic: the fused guidance + update kernel
(cl_ddim_step) on the inversion table, the forward-process kernel (cl_qsample) on the DDIM tables, and the index /
timestep bookkeeping.
"""
import contextlib
import io
import os

import pytest
import torch

from tests.util import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu

ENC_TOL = 1e-5     # fp32: the kernel's form of the update (x0-prediction, then direction) vs the reference's two-coefficient
#                    form -- measured 2e-7 ... 6e-7 after 10 - 50 steps; GPU tanh of the analytic model included


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")


class _AnalyticModel:
    """What DDIMSampler touches of a model, with the fixture's eps(x, t, c) = tanh(0.7 x + 0.001 t) * c."""
    num_timesteps = 1000
    parameterization = "eps"

    def __init__(self, dev):
        from oracle import ref_model as R
        s = R.make_schedule()
        self.device = dev
        self.betas = s["betas"].to(dev)
        self.alphas_cumprod = s["alphas_cumprod"].to(dev)
        self.alphas_cumprod_prev = s["alphas_cumprod_prev"].to(dev)
        self.calls = []

    def apply_model(self, x, t, c):
        if isinstance(c, dict):
            c = c["c_crossattn"][0]
        self.calls.append(int(x.shape[0]))
        return torch.tanh(0.7 * x + 0.001 * t.float().view(-1, 1, 1, 1)) * c.view(-1, 1, 1, 1)


def _sampler(S, eta=0.0):
    from cldm.ddim_hacked import DDIMSampler
    dev = torch.device("cuda")
    m = _AnalyticModel(dev)
    s = DDIMSampler(m)
    s.use_graph = False
    s.make_schedule(S, ddim_eta=eta, verbose=False)
    return s, m


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_encode_matches_reference_sampler_and_keeps_the_same_intermediates():
    _need_gpu()
    g = torch.load(os.path.join(GOLDEN, "ddim_codec.pt"), weights_only=False)
    dev = torch.device("cuda")
    x0, c, uc = g["x0"].to(dev), g["c"].to(dev), g["uc"].to(dev)
    worst = 0.0
    for key, ref in g["encode"].items():
        orig = key.startswith("orig")
        if orig:
            S, t_enc, scale, ri = 50, 40, 1.0, None
        else:
            S = int(key.split("_")[0][1:]); t_enc = int(key.split("_")[1][1:])
            scale = float(key.split("cfg")[1].split("_")[0])
            ri = key.split("ri")[1]
            ri = None if ri == "None" else int(ri)
        s, m = _sampler(S)
        seen = []
        x_enc, info = s.encode(x0.clone(), c, t_enc, use_original_steps=orig, return_intermediates=ri,
                               unconditional_guidance_scale=scale, unconditional_conditioning=uc if scale != 1.0 else None,
                               callback=seen.append)
        assert seen == list(range(t_enc))
        assert info["intermediate_steps"] == ref["intermediate_steps"], key
        assert info["x_encoded"] is x_enc
        e = rel_l2(x_enc.cpu(), ref["x_encoded"])
        worst = max(worst, e)
        assert e < ENC_TOL, (key, e)
        for got, want in zip(info.get("intermediates", []), ref.get("intermediates", [])):
            assert rel_l2(got.cpu(), want) < ENC_TOL, key
        # guidance = ONE batch of 2B per step, ordered [unconditional; conditional] as the reference batches it
        assert m.calls == [2 * x0.shape[0] if scale != 1.0 else x0.shape[0]] * t_enc, key
    print(f"[ddim codec] encode worst rel-L2 vs the reference sampler {worst:.2e}")


def test_encode_guidance_with_dict_conditionings_and_two_pass_fallback():
    """ControlLDM conditionings are dicts (the reference's torch.cat of them would raise): same-structure dicts are batched
    [unconditional; conditional]; structurally different ones run as two passes.  All three forms give the tensor result."""
    _need_gpu()
    g = torch.load(os.path.join(GOLDEN, "ddim_codec.pt"), weights_only=False)
    dev = torch.device("cuda")
    x0, c, uc = g["x0"].to(dev), g["c"].to(dev), g["uc"].to(dev)
    ref = g["encode"]["S20_t12_cfg3.0_ri4"]["x_encoded"]
    s, m = _sampler(20)
    a, _ = s.encode(x0.clone(), {"c_crossattn": [c]}, 12, unconditional_guidance_scale=3.0,
                    unconditional_conditioning={"c_crossattn": [uc]})
    assert m.calls == [4] * 12
    assert rel_l2(a.cpu(), ref) < ENC_TOL
    s, m = _sampler(20)
    b, _ = s.encode(x0.clone(), {"c_crossattn": [c]}, 12, unconditional_guidance_scale=3.0, unconditional_conditioning=uc)
    assert m.calls == [2] * 24
    assert rel_l2(b.cpu(), ref) < ENC_TOL
    with pytest.raises(AssertionError):
        s.encode(x0.clone(), c, 12, unconditional_guidance_scale=3.0)            # guidance without an unconditional input
    with pytest.raises(AssertionError):
        s.encode(x0.clone(), c, 21)                                              # more steps than the schedule has


def test_decode_and_round_trip_match_reference_sampler():
    _need_gpu()
    g = torch.load(os.path.join(GOLDEN, "ddim_codec.pt"), weights_only=False)
    dev = torch.device("cuda")
    x0, c, uc = g["x0"].to(dev), g["c"].to(dev), g["uc"].to(dev)
    for key, ref in g["decode"].items():
        S = int(key.split("_")[0][1:]); eta = float(key.split("_")[1][3:]); t_start = int(key.split("_")[2][1:])
        scale = float(key.split("cfg")[1])
        s, _ = _sampler(S, eta)
        if eta == 0.0:
            out = _quiet(s.decode, x0.clone(), c, t_start, unconditional_guidance_scale=scale, unconditional_conditioning=uc)
            assert rel_l2(out.cpu(), ref["x_dec"]) < ENC_TOL, key
        else:
            # eta > 0: noise_like draws the per-step noise from the DEVICE generator (the reference's fixture used the CPU
            # stream), so the same device stream is drawn first and handed to the oracle's restatement of decode
            from oracle import ref_model as R

            def eps(x, t, cond):
                return torch.tanh(0.7 * x + 0.001 * t.float().view(-1, 1, 1, 1)) * (1.0 if cond else 0.6)
            torch.manual_seed(77)
            noises = [torch.randn(x0.shape, device=dev).cpu() for _ in range(t_start)]
            want = R.ddim_decode(eps, R.make_schedule(), S, g["x0"], t_start, scale=scale, uncond=True, eta=eta, noises=noises)
            seen = []
            torch.manual_seed(77)
            out = _quiet(s.decode, x0.clone(), c, t_start, unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                         callback=seen.append)
            assert seen == list(range(t_start))
            assert rel_l2(out.cpu(), want) < ENC_TOL, key
    s, _ = _sampler(50)
    x_enc, _ = s.encode(x0.clone(), c, 50)
    back = _quiet(s.decode, x_enc, c, 50)
    assert rel_l2(back.cpu(), g["roundtrip_S50"]["x_back"]) < 5e-5
    assert abs(float((back - x0).norm() / x0.norm()) - g["roundtrip_S50"]["rel"]) < 1e-4
    with pytest.raises(NotImplementedError):
        s.decode(x_enc, c, 50, use_original_steps=True)


def test_stochastic_encode_is_bit_exact_with_reference_sampler():
    _need_gpu()
    g = torch.load(os.path.join(GOLDEN, "ddim_codec.pt"), weights_only=False)
    dev = torch.device("cuda")
    x0, noise = g["x0"].to(dev), g["noise"].to(dev)
    for key, ref in g["stochastic"].items():
        S = int(key.split("_")[0][1:]); orig = key.endswith("orig1")
        s, _ = _sampler(S)
        y = s.stochastic_encode(x0.clone(), ref["t"].to(dev), use_original_steps=orig, noise=noise.clone())
        assert torch.equal(y.cpu(), ref["y"]), key
    s, _ = _sampler(50)
    torch.manual_seed(5)
    y = s.stochastic_encode(x0, torch.tensor([10, 20], device=dev))              # noise drawn inside
    assert y.shape == x0.shape and bool(torch.isfinite(y).all())
