"""tools/kernel_resources.py: the register / LDS / scratch table read from hipcc's listings (profiles/r05_final/kernel_resources.txt).
Parsing is checked on a synthetic metadata block; two small sources are compiled for real (hipcc cross-compiles, no GPU) and the
budgets DESIGN quotes for them are asserted: the d_head-40 attention forward fits two waves per SIMD without scratch, the
weight-gradient kernel of the default ring fits the three workgroups per CU the ring was chosen for."""
import os
import shutil
import sys

import pytest

from tests.util import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr                                                             # noqa: E402

META = """
	.amdgpu_metadata
---
amdhsa.kernels:
  - .agpr_count:     32
    .args:
      - .name:           not_the_kernel
        .offset:         0
    .group_segment_fixed_size: 16384
    .max_flat_workgroup_size: 256
    .name:           _ZN2cl14gemm_xs_kernelILi40ELi0ELi3ELi1ELi0ELb1EEEvNS_10GemmParamsEi
    .private_segment_fixed_size: 0
    .sgpr_count:     44
    .sgpr_spill_count: 0
    .uses_dynamic_stack: false
    .vgpr_count:     288
    .vgpr_spill_count: 0
  - .agpr_count:     0
    .group_segment_fixed_size: 0
    .name:           _ZN2cl12_GLOBAL__N_114gn1_fwd_kernelItLb1ELi16EEEvNS_8GnParamsE
    .private_segment_fixed_size: 36
    .sgpr_count:     96
    .sgpr_spill_count: 0
    .uses_dynamic_stack: false
    .vgpr_count:     128
    .vgpr_spill_count: 8
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
"""


def test_metadata_block_is_parsed_and_occupancy_derived():
    ks = [kr.derive(k) for k in kr.parse_metadata(META)]
    assert [k["symbol"][:24] for k in ks] == ["_ZN2cl14gemm_xs_kernelIL", "_ZN2cl12_GLOBAL__N_114gn"]
    a, b = ks
    # .vgpr_count is the unified count on gfx90a+: 256 arch + 32 accumulation registers = 288 -> one wave per SIMD
    assert (a["unified"], a["arch_vgpr"], a["waves_simd"], a["lds_wg_cu"], a["flag"]) == (288, 256, 1, 10, False)
    assert (b["unified"], b["waves_simd"], b["lds_wg_cu"], b["flag"]) == (128, 4, None, True)


def test_names_match_the_form_the_traces_print():
    f = kr.short_name
    assert f("void cl::gemm_xs_kernel<20, 0, 3, 2, 0, false>(cl::GemmParams, int)") == "gemm_xs_kernel<20, 0, 3, 2, 0, false>"
    assert f("void cl::(anonymous namespace)::gemm_fl_kernel<unsigned short, 256, 160, 4, 2, 1, 3, 3>(cl::GemmParams, int)") \
        == "gemm_fl_kernel<unsigned short, 256, 160, 4, 2, 1, 3, 3>"
    assert f("cl::attn_fwd40_kernel(cl::AttnParams, void const*)") == "attn_fwd40_kernel"
    assert f("void cl::k<(ctrlora_dtype)1>(void (*)(int))") == "k<1>"


def test_stats_join_and_flag_lines(tmp_path):
    p = tmp_path / "stats.txt"
    p.write_text("kernels: 10 dispatches\nname      calls   total_ms    avg_us      %\n"
                 "gemm_xs_kernel<40, 0, 3, 1, 0, true>            52      2.157     41.49   1.55\n"
                 "gn1_fwd_kernel<unsigned short, true>           80      0.100      1.25   0.19\n"
                 "Cijk_library_kernel                                4      0.010      2.50   0.01\n")
    st = kr.read_stats(str(p))
    assert st["gemm_xs_kernel<40, 0, 3, 1, 0, true>"] == (52, 2.157, 1.55) and len(st) == 3
    rows = [kr.derive(k) for k in kr.parse_metadata(META)]
    rows[0].update(source="gemm_xs.hip", name="gemm_xs_kernel<40, 0, 3, 1, 0, true>")
    rows[1].update(source="norm.hip", name="gn1_fwd_kernel<unsigned short, true, 16>")
    text = kr.table(rows, st, "t")
    # a trace older than a template parameter still finds its kernel; the library kernel is reported, not dropped silently
    assert "! gn1_fwd_kernel<unsigned short, true, 16>" in text and "Cijk_library_kernel" in text.splitlines()[-2]
    assert "flagged (scratch / spills): 1" in text


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_budgets_of_two_hot_kernels(tmp_path):
    rows = {k["name"]: k for k in kr.collect(sources=["attention_fwd40.hip", "wgrad.hip"], outdir=str(tmp_path))}
    a = rows["attn_fwd40_kernel"]
    assert not a["flag"] and a["agpr_count"] == 0 and a["waves_simd"] == 2, a          # two 4-wave workgroups per CU
    ring = 3                                                                            # g_wgrad_ring default (wgrad.hip)
    w = rows[f"wgrad_tn_kernel<{ring}, 32>"]
    assert not w["flag"] and w["waves_simd"] >= 3, w                                    # 3 workgroups of 4 waves per CU
    assert all(not k["flag"] for n, k in rows.items() if n.startswith("wgrad_"))


def test_committed_table_covers_the_measured_steps():
    path = os.path.join(ROOT, "profiles", "r05_final", "kernel_resources.txt")
    text = open(path).read()
    assert "## kernels of profiles/r05_final/train_kernel_stats_steady.txt" in text
    assert "## kernels of profiles/r05_final/ddim_kernel_stats_steady.txt" in text
    cover = [float(x) for x in __import__("re").findall(r"listed kernels cover ([\d.]+) %", text)]
    assert len(cover) == 2 and min(cover) > 90.0
