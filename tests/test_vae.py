"""First stage (AutoencoderKL) -- SURVEY.md 8(f1): the mirror's plain-torch modules are pinned to the UNMODIFIED
reference (tests/golden/vae.pt, made by tests/golden/make_golden_vae.py); the HIP engine (ctrlora_amd/engine/vae.py)
is checked against that fixture and, at 512 x 512, against the pinned torch modules on the same GPU."""
import os

import pytest
import torch

from tests.util import GOLDEN, rel_l2


def _build(device="cpu"):
    from ldm.models.autoencoder import AutoencoderKL
    from tests.golden.make_golden_vae import DDCONFIG, vae_state
    m = AutoencoderKL(ddconfig=DDCONFIG, lossconfig=dict(target="torch.nn.Identity"), embed_dim=4).eval()
    m.load_state_dict(vae_state(m), strict=True)
    return m.to(device)


def _gold():
    return torch.load(os.path.join(GOLDEN, "vae.pt"), weights_only=False)


def test_mirror_autoencoder_matches_the_reference_golden():
    from tests.golden.make_golden_vae import SEED, test_image
    g = _gold()
    m = _build()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == g["keys"]
    with torch.no_grad():
        post = m.encode(test_image(1, 256, 256))
        dec = m.decode(torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(SEED)))
    assert rel_l2(post.parameters, g["moments"]) < 2e-5
    d = g["decoded"]
    assert list(dec.shape) == d["shape"]
    assert rel_l2(dec.flatten()[d["idx"]], d["vals"]) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol_m,tol_d", [(torch.float32, 2e-4, 2e-4), (torch.bfloat16, 3e-2, 3e-2)])
def test_vae_engine_matches_reference_golden(dtype, tol_m, tol_d):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.golden.make_golden_vae import SEED, test_image
    g = _gold()
    m = _build("cuda")
    m.engine_dtype = dtype
    with torch.no_grad():
        post = m.encode(test_image(1, 256, 256).cuda())
        dec = m.decode(torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(SEED)).cuda())
    assert "_enc" in m.__dict__ and "_dec" in m.__dict__          # the HIP engine ran, not the torch modules
    e_m = rel_l2(post.parameters, g["moments"])
    d = g["decoded"]
    e_d = rel_l2(dec.flatten()[d["idx"].cuda()], d["vals"])
    print(f"[vae {dtype}] moments rel-L2 {e_m:.3e}, decoded rel-L2 {e_d:.3e}")
    assert e_m < tol_m and e_d < tol_d, (e_m, e_d)


@pytest.mark.gpu
def test_vae_engine_512_batch_matches_torch_modules_on_gpu():
    """BASELINE geometry: 512 x 512 condition images (latent 64 x 64, N = 4096 tokens in the middle attention), B = 2;
    fp32 engine vs the (reference-pinned) torch modules running on PyTorch-ROCm's own kernels."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.golden.make_golden_vae import test_image
    m = _build("cuda")
    m.engine_dtype = torch.float32
    x = test_image(2, 512, 512).cuda()
    z = torch.randn(2, 4, 64, 64, generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        m.use_engine = False
        ref_m, ref_d = m.encode(x).parameters, m.decode(z)
        m.use_engine = True
        got_m, got_d = m.encode(x).parameters, m.decode(z)
    assert rel_l2(got_m, ref_m) < 2e-4 and rel_l2(got_d, ref_d) < 2e-4
    m.engine_dtype = torch.bfloat16
    m.invalidate_engine()
    with torch.no_grad():
        b_m, b_d = m.encode(x).parameters, m.decode(z)
    e1, e2 = rel_l2(b_m, ref_m), rel_l2(b_d, ref_d)
    print(f"[vae 512 bf16] moments {e1:.3e} decoded {e2:.3e}")
    assert e1 < 3e-2 and e2 < 3e-2
