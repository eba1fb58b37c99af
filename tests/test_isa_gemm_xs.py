"""Static checks of the compiled x-stationary kernels (csrc/gemm_xs.hip), in the spirit of tests/test_isa_mfma_hazards.py: the
kernel's counted `s_waitcnt vmcnt(n)` arithmetic (tests/test_xs_vmcnt_model.py) assumes exact instruction counts per chunk --
TWO 16-byte stores per output block, TWO 16-byte residual loads, K / 64 LDS-DMA instructions per wave and chunk -- so the LISTING
is checked for what the compiler could silently change: a 16-byte store split into narrower ones (or a predicated tail), spills
to scratch inside the counted region, an x fragment load that is not one dwordx4.  hipcc cross-compiles; no GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

from tests.util import ROOT


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("needs hipcc")
    out = tmp_path_factory.mktemp("isa") / "gemm_xs.s"
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "ctrlora_amd", "csrc", "gemm_xs.hip"), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    kernels = {}
    for nm in re.findall(r"\n(_ZN2cl14gemm_xs_kernel\S*): ", text):
        i = text.index("\n" + nm + ": ")
        body = text[i:text.index("s_endpgm", i)]
        ks1, ks2, ring, minw, epi, lnp = map(int, re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)", nm).groups())
        kernels[(ks1, ks2, ring, minw, epi, lnp)] = body
    return kernels


def test_every_instance_the_launcher_names_is_compiled(listing):
    want = {(20, 0, 3, 2), (20, 8, 2, 2), (40, 0, 3, 1), (40, 8, 3, 1)}
    for epi in (0, 1, 2):
        assert {k[:4] for k in listing if k[4] == epi and k[5] == 0} == want, epi
    assert {k[:5] for k in listing if k[5] == 1} == {(20, 0, 3, 2, 0), (20, 0, 3, 2, 2), (40, 0, 3, 1, 0), (40, 0, 3, 1, 2)}
    assert len(listing) == 16


def test_instruction_shapes_the_vmcnt_arithmetic_relies_on(listing):
    for (ks1, ks2, ring, minw, epi, lnp), body in listing.items():
        key = (ks1, ks2, ring, minw, epi, lnp)
        cnt = lambda pat: len(re.findall(pat, body))
        ks, dpc = ks1 + ks2, (ks1 + ks2) // 4
        # no spills: a scratch access inside the loop would be a vector-memory operation the count does not know about
        assert cnt(r"\bscratch_") == 0, key
        # stores: every output store is ONE dwordx4 (pairs: SPB = 2 per block site); the only other store is the LayerNorm
        # prologue's (mean, rstd) pair -- issued before the first DMA, outside the counted region
        st16, st_all = cnt(r"global_store_dwordx4"), cnt(r"global_store_")
        assert st16 >= 4 and st16 % 2 == 0, (key, st16)
        assert st_all - st16 == (1 if lnp else 0), (key, st_all, st16)
        assert cnt(r"buffer_store_|flat_store_") == 0, key
        # LDS-DMA: whole chunks only (K / 64 instructions per wave and chunk), 16 bytes per lane
        dma = cnt(r"global_load_lds_dwordx4")
        assert dma > 0 and dma % dpc == 0 and cnt(r"global_load_lds_dword\b|global_load_lds_ushort|global_load_lds_ubyte") == 0, (key, dma, dpc)
        # x fragments: one dwordx4 per k-step (+ 2 per residual site for XS_RES); nothing narrower on the vector-memory path
        ld16 = cnt(r"global_load_dwordx4")
        assert ld16 >= ks and (ld16 - ks) % 2 == 0 and ((ld16 - ks) > 0) == (epi == 1), (key, ld16)
        assert cnt(r"global_load_dwordx2|global_load_dwordx3|global_load_ushort|global_load_ubyte|buffer_load_|flat_load_") == 0, key
        # the MFMA chains are whole (K / 16 per chunk) and 32x32x16 bf16; the run-time vmcnt switch is there (case 0 .. 32:
        # counts above that are the compiler's own waits on the x-fragment loads of the prologue)
        mf = cnt(r"v_mfma_f32_32x32x16_bf16")
        assert mf > 0 and mf % ks == 0 and cnt(r"v_mfma_") == mf, (key, mf)
        waits = {int(v) for v in re.findall(r"s_waitcnt[^\n]*vmcnt\((\d+)\)", body)}
        assert 0 in waits and (dpc in waits if ring == 3 else 2 in waits), (key, sorted(waits))
        # the T21 store widening is there (two swaps per store), and the register budget of the launch bounds holds
        assert cnt(r"v_permlane32_swap") >= 2 * st16, key
        vg = int(re.search(r"\.vgpr_count:\s+(\d+)", body + "").group(1)) if re.search(r"\.vgpr_count:\s+(\d+)", body) else None
        if vg is not None:
            assert vg <= (256 if minw <= 2 else 168), (key, vg)
