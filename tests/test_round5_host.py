"""Host-side logic of round 5 (no GPU): the x-stationary kernel's routing rules and its launch-table overlay, the launch-tag
table's C entry points and the decode tools/prof_shapes.py does with it, fast model construction (ctrlora_amd/fastinit.py)."""
import json
import os
import subprocess
import sys

import torch
import torch.nn as nn

from tests.util import ROOT


def test_xs_routing_rules_are_what_was_measured():
    """hip.xs_geglu_ok / xs_ln_ok: where the fused GEGLU projection and the LayerNorm prologue go to csrc/gemm_xs.hip
    (profiles/r05_gemm_xs/probe_xs_fast_gelu.log); CTRLORA_GEMM_XS / CTRLORA_LN_PROLOGUE switch both off."""
    from ctrlora_amd import hip
    assert hip.XS_ENABLED and hip.LN_PROLOGUE
    assert hip.xs_geglu_ok(32768, 320, 0) and hip.xs_geglu_ok(131072, 320, 128) and hip.xs_geglu_ok(8192, 640, 0)
    assert not hip.xs_geglu_ok(2048, 640, 0) and not hip.xs_geglu_ok(8192, 1280, 0) and not hip.xs_geglu_ok(32768, 320, 32)
    assert hip.xs_ln_ok(32768, 960, 320) and hip.xs_ln_ok(8192, 640, 640) and hip.xs_ln_ok(32768, 2560, 320, hip.ACT_GEGLU_SPLIT)
    assert not hip.xs_ln_ok(2048, 3840, 1280)                       # K = 1280: no instance (x would need 320 VGPRs)
    assert not hip.xs_ln_ok(32768, 950, 320) and not hip.xs_ln_ok(64, 960, 320)
    assert not hip.xs_ln_ok(2048, 5120, 640, hip.ACT_GEGLU_SPLIT)   # that GEGLU stays on the tile kernels: no prologue there
    assert not hip.xs_ln_ok(32768, 960, 320, hip.ACT_SILU)
    keep = hip.XS_ENABLED, hip.LN_PROLOGUE
    try:
        hip.LN_PROLOGUE = False
        assert not hip.xs_ln_ok(32768, 960, 320) and hip.xs_geglu_ok(32768, 320, 0)
        hip.LN_PROLOGUE, hip.XS_ENABLED = True, False
        assert not hip.xs_ln_ok(32768, 960, 320) and not hip.xs_geglu_ok(32768, 320, 0)
    finally:
        hip.XS_ENABLED, hip.LN_PROLOGUE = keep


def test_xs_overlay_table_layers_over_the_base_table():
    from ctrlora_amd import hip
    base = json.load(open(hip.GEMM_TABLE_PATH))["entries"]
    over = json.load(open(hip.GEMM_XS_TABLE_PATH))["entries"]
    assert over and all(r[7] == 34 and r[0] == hip.BF16 and r[1] == 0 and r[4] in (320, 640) and r[5] in (0, 128) and r[3] % 32 == 0
                        and 0 <= r[8] <= 8 for r in over)
    assert not any(r[7] == 34 for r in base)                       # CTRLORA_GEMM_XS=0 must be able to drop every use of it
    L = hip.lib()
    n_base = hip.load_gemm_table(hip.GEMM_TABLE_PATH)
    n_all = hip.load_gemm_table(hip.GEMM_XS_TABLE_PATH, clear=False)
    assert n_all == len({tuple(r[:7]) for r in base} | {tuple(r[:7]) for r in over}) >= n_base
    assert L.cl_gemm_tune_set(0, 0, 64, 64, 64, 0, 0, 48, 0) == 0 and L.cl_gemm_tune_set(0, 0, 64, 64, 64, 0, 0, 49, 0) != 0
    hip.load_gemm_table(hip.GEMM_TABLE_PATH)
    hip.load_gemm_table(hip.GEMM_XS_TABLE_PATH, clear=False)


def test_launch_tag_table_entry_points_and_decode(tmp_path):
    """cl_debug_gemm_tag* (csrc/debug_hooks.h) load and answer without a GPU (no launch happens: the table stays empty), and
    tools/prof_shapes.py attributes trace rows to signatures from (workgroups - real grid): tile kernels + tag, the
    x-stationary kernel (row blocks + tag) x column runs."""
    from ctrlora_amd import hip
    L = hip.lib()
    assert L.cl_debug_gemm_tag(1) == 0 and L.cl_debug_gemm_tag_count() >= 0 and L.cl_debug_gemm_tag(0) == 0
    assert isinstance(hip.gemm_tags(), list)
    tags = [dict(dtype=0, mode=1, M=32768, N=320, K1=320, K2=0, act=0, residual=0, tag=1, workgroups=256, wg_size=512, launches=11),
            dict(dtype=0, mode=0, M=32768, N=2560, K1=320, K2=0, act=0, residual=0, tag=2, workgroups=512, wg_size=256, launches=3),
            dict(dtype=0, mode=0, M=2048, N=1280, K1=1280, K2=0, act=0, residual=1, tag=3, workgroups=512, wg_size=256, launches=21)]
    (tmp_path / "tags.json").write_text(json.dumps(tags))
    rows = ["Kernel_Name,Start_Timestamp,End_Timestamp,Grid_Size,Workgroup_Size",
            f'"void cl::gemm_fl_kernel<unsigned short, 256, 160>(cl::GemmParams)",1000,59000,{(256 + 1) * 512},512',
            f'"void cl::gemm_xs_kernel<20, 0, 3, 2, 0, false>(cl::GemmParams, int)",60000,124000,{(256 + 2) * 2 * 256},256',
            f'"void cl::gemm_fl_kernel<unsigned short, 64, 80>(cl::GemmParams)",130000,149000,{(512 + 3) * 256},256',
            f'"void cl::gemm_fl_kernel<unsigned short, 64, 80>(cl::GemmParams)",150000,170000,{(512 + 7) * 256},256',      # no such tag
            '"void cl::ln_fwd_kernel<unsigned short>()",171000,180000,65536,256']
    (tmp_path / "trace_kernel_trace.csv").write_text("\n".join(rows) + "\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_shapes.py"), str(tmp_path / "trace_kernel_trace.csv"),
                        str(tmp_path / "tags.json")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "3 signatures x kernels" in out and "(1 dispatches" in out
    body = [ln for ln in out.splitlines() if ln[:1] == " " and ln.split()[0].replace(".", "").isdigit()]
    by_shape = {tuple(ln.split()[6:11]): ln.split() for ln in body}
    assert by_shape[("1", "32768", "320", "320", "0")][2] == "58.0"        # avg us of the conv row
    assert by_shape[("0", "32768", "2560", "320", "0")][2] == "64.0" and "gemm_xs_kernel" in " ".join(by_shape[("0", "32768", "2560", "320", "0")])
    assert by_shape[("0", "2048", "1280", "1280", "0")][12] == "1"         # the residual flag travels with the signature


def test_fast_init_keeps_explicit_initialisations_and_the_default_distribution():
    from ctrlora_amd.fastinit import fill_default_init, skip_default_init
    saved = nn.Linear.reset_parameters, nn.modules.conv._ConvNd.reset_parameters

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(400, 300)
            self.conv = nn.Conv2d(16, 32, 3, padding=1)
            self.zero = nn.Conv2d(8, 8, 1)
            for p in self.zero.parameters():
                p.detach().zero_()                                   # zero_module (cldm/cldm.py): explicit, after construction
            self.down = nn.Linear(400, 8, bias=False)
            nn.init.normal_(self.down.weight, std=1 / 8)             # LoRALinearLayer (cldm/lora.py:67-68)
            self.norm = nn.GroupNorm(4, 32)
            self.emb = nn.Embedding(10, 7)

    with skip_default_init():
        net = Net()
    assert (nn.Linear.reset_parameters, nn.modules.conv._ConvNd.reset_parameters) == saved          # restored on exit
    assert torch.isnan(net.lin.weight).all() and torch.isnan(net.conv.bias).all()                  # marked, not drawn
    down = net.down.weight.detach().clone()
    fill_default_init(net, seed=3)
    assert all(torch.isfinite(p).all() for p in net.parameters())
    assert float(net.zero.weight.abs().max()) == 0.0 and float(net.zero.bias.abs().max()) == 0.0
    assert torch.equal(net.down.weight, down) and 0.08 < float(down.std()) < 0.17
    assert torch.equal(net.norm.weight, torch.ones(32)) and float(net.emb.weight.std()) > 0.5
    for m, fan_in in ((net.lin, 400), (net.conv, 16 * 9)):
        b = fan_in ** -0.5
        assert float(m.weight.abs().max()) <= b and float(m.bias.abs().max()) <= b
        assert abs(float(m.weight.std()) - b / 3 ** 0.5) < 0.05 * b  # U(-b, b)
    # the same seed gives the same weights, another one does not
    with skip_default_init():
        n2 = Net()
    fill_default_init(n2, seed=3)
    assert torch.equal(n2.lin.weight, net.lin.weight)
    with skip_default_init():
        n3 = Net()
    fill_default_init(n3, seed=4)
    assert not torch.equal(n3.lin.weight, net.lin.weight)
    # an exception inside the context still restores torch's constructors
    try:
        with skip_default_init():
            raise RuntimeError("boom")
    except RuntimeError:
        pass
    assert (nn.Linear.reset_parameters, nn.modules.conv._ConvNd.reset_parameters) == saved
