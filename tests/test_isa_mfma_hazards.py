"""Static hazard check of the one kernel that issues its MFMAs from inline asm (csrc/attention_fwd40.hip): the compiler inserts
no wait states in front of readers of an inline-asm MFMA's result, and -- found in round 4 -- it may schedule such a reader
in front of a hand-placed `s_nop` drain that names no register.  tools/isa_mfma_hazards.py replays the listing with a simple
issue-slot model (P + 3 wait states behind a P-pass MFMA; back-to-back MFMAs paced by the matrix pipe) and flags every
non-MFMA instruction that touches an MFMA's destination too early.  No GPU needed (hipcc cross-compiles)."""
import importlib.util
import os
import shutil
import subprocess

import pytest

from tests.util import ROOT

spec = importlib.util.spec_from_file_location("isa_mfma_hazards", os.path.join(ROOT, "tools", "isa_mfma_hazards.py"))
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)


def _scan(text):
    lines = text.split("\n")
    out = []
    for name, lo, hi in lint.kernels(lines):
        out += lint.scan(lines, lo, hi)
    return out


LISTING = """
_Z1kv:
	v_mfma_f32_32x32x16_bf16 v[2:17], v[80:83], v[84:87], v[2:17]
	{between}
	ds_bpermute_b32 v40, v41, v6
	s_endpgm
"""


def test_lint_flags_a_reader_behind_too_few_wait_states_and_accepts_enough():
    assert len(_scan(LISTING.format(between="s_nop 0"))) == 1                      # the round-4 bug: one s_nop 0 behind the MFMA
    assert len(_scan(LISTING.format(between="s_nop 9"))) == 1                      # 10 wait states: one short of 8 + 3
    assert _scan(LISTING.format(between="s_nop 10")) == []
    assert _scan(LISTING.format(between="s_nop 15\n\ts_nop 15")) == []             # the kernel's drain
    # two unrelated MFMAs in between pace the consumer by the matrix pipe (8 slots each): enough without a single s_nop
    two = "v_mfma_f32_32x32x16_bf16 v[20:35], v[80:83], v[84:87], v[20:35]\n\tv_mfma_f32_32x32x16_bf16 v[50:65], v[80:83], v[84:87], v[50:65]"
    assert _scan(LISTING.format(between=two)) == []
    one = "v_mfma_f32_32x32x16_bf16 v[20:35], v[80:83], v[84:87], v[20:35]\n\tv_mov_b32 v90, v91"
    assert len(_scan(LISTING.format(between=one))) == 1                            # one MFMA + one VALU in between is not
    # an MFMA reading another MFMA's result is the matrix pipe's own business (not flagged here)
    assert _scan(LISTING.replace("ds_bpermute_b32 v40, v41, v6", "v_mfma_f32_32x32x16_bf16 v[2:17], v[80:83], v[84:87], v[2:17]")
                 .format(between="s_nop 0")) == []


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
@pytest.mark.parametrize("source,key,min_mfma", [("attention_fwd40.hip", "attn_fwd40", 60),      # inline-asm MFMAs: the case the tool exists for
                                                 ("attention_tr.hip", "attn_", 1000)])            # builtin MFMAs beside inline-asm LDS reads: the
def test_attention_listings_have_no_early_reader_of_an_mfma_result(tmp_path, source, key, min_mfma):     # compiler's own nops must satisfy the model
    src = os.path.join(ROOT, "ctrlora_amd", "csrc", source)
    out = tmp_path / (source + ".s")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-S", "--cuda-device-only",
                        src, "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = out.read_text().split("\n")
    ks = [k for k in lint.kernels(lines) if key in k[0]]
    assert ks, "no kernel found in the listing"
    found = [(name,) + f for name, lo, hi in ks for f in lint.scan(lines, lo, hi)]
    assert found == [], "\n".join(f"{name[:60]} line {ln}: {s}  <- line {mln} ({have} of {need} wait states)"
                                   for name, ln, s, mln, _, have, need in found[:10])
    assert sum(1 for l in lines if "\tv_mfma_f32_" in l) >= min_mfma, "the listing should contain the kernels' MFMA streams"
