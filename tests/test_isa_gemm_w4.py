"""Static checks of the compiled loader / consumer kernel (csrc/gemm_w4.hip), in the spirit of tests/test_isa_gemm_xs.py.  The
consumers' main loop is a hand-ordered stream whose correctness rests on COUNTED waits: LDS returns in order, so "fragment k has
landed" is `s_waitcnt lgkmcnt(reads issued behind it)`, and the counts come from a constexpr replay of the issue order (w4_sched).
This test (a) replays that order independently in Python and (b) reads the LISTING for what the compiler could silently change:
the order and number of reads / MFMAs / barriers per stage, a compiler v_mov / scratch access / extra wait inside the stream, the
loaders' DMA counts that their counted vmcnt assumes.  hipcc cross-compiles; no GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

from tests.util import ROOT


def sched(NW):
    """The issue order of one consumer stage, from the kernel's comments (not from its code): behind group g go out W[g + R];
    X(k-half 1, g) for g < 4; two X(next stage, k-half 0) behind groups GB + 1 and GB + 2.  Returns the lgkmcnt each group may run at."""
    G2, R = 2 * NW, NW // 2
    GB = G2 - R - 1
    seq = []
    for S in range(3):
        for g in range(G2):
            seq.append(("grp", S, g))
            k = g + R
            seq.append(("W", S, k) if k < G2 else ("W", S + 1, k - G2))
            if g < 4:
                seq.append(("X1", S, g))
            if g == GB + 1:
                seq += [("X0", S + 1, 0), ("X0", S + 1, 1)]
            if g == GB + 2:
                seq += [("X0", S + 1, 2), ("X0", S + 1, 3)]
    allow = []
    for g in range(G2):
        pg = seq.index(("grp", 1, g))
        need = [("W", 1, g)]
        if g == 0:
            need += [("X0", 1, i) for i in range(4)]
        if g == NW:
            need += [("X1", 1, i) for i in range(4)]
        last = max(seq.index(n) for n in need)
        assert last < pg
        allow.append(sum(1 for e in seq[last + 1:pg] if e[0] != "grp"))
    return allow


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("needs hipcc")
    out = tmp_path_factory.mktemp("isa") / "gemm_w4.s"
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "ctrlora_amd", "csrc", "gemm_w4.hip"), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    kernels = {}
    for nm in re.findall(r"\n(_ZN2cl\S*gemm_w4_kernel\S*):", text):
        i = text.index("\n" + nm + ":")
        body = text[i:text.index(".Lfunc_end", i)]
        nw, mode, abl = map(int, re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)E", nm).groups())
        kernels[(nw, mode, abl)] = body
    return kernels


def test_every_instance_the_launcher_names_is_compiled(listing):
    assert set(listing) == {(nw, mode, 0) for nw in (10, 8) for mode in (0, 1, 2, 3)}


def _blocks(body):
    cur, out = [], []
    for ln in body.split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", ln):
            out.append(cur); cur = []
        else:
            cur.append(ln.strip())
    out.append(cur)
    return out


def _stream(block):
    """One stage instance: the block up to its last branch (a fall-through block behind it has no label of its own)."""
    cuts = [k for k, ln in enumerate(block) if ln.startswith("s_cbranch")]
    lines = block[:max(cuts) + 1] if cuts else block
    ins = [ln.split()[0] for ln in lines if ln and not ln.startswith((";", ".", "//"))]
    waits = [int(m.group(1)) for ln in lines for m in [re.match(r"s_waitcnt lgkmcnt\((\d+)\)$", ln.split(";")[0].strip())] if m]
    return lines, ins, waits


def test_consumer_stream_is_what_the_counts_assume(listing):
    for (nw, mode, _), body in listing.items():
        G2, R = 2 * nw, nw // 2
        GB = G2 - R - 1
        inst = [b for b in _blocks(body) if sum("v_mfma" in ln for ln in b) >= 4 * G2]
        # two instances of the stage: the steady one (reads on into the next stage) and a tile's last stage (reads nothing ahead)
        assert len(inst) == 2, ((nw, mode), [sum("v_mfma" in ln for ln in b) for b in _blocks(body) if any("v_mfma" in ln for ln in b)])
        inst.sort(key=lambda b: -sum(ln.startswith("ds_read_b128") for ln in _stream(b)[0]))
        for which, block in enumerate(inst):
            lines, ins, waits = _stream(block)
            cnt = lambda p: sum(1 for i in ins if re.fullmatch(p, i))
            # (the last stage's block runs on into the epilogue: only its first 4 G2 MFMAs and what lies between them are the stage)
            if which == 1:
                k = [j for j, i in enumerate(ins) if i.startswith("v_mfma")][4 * G2 - 1]
                ins = ins[:k + 1]
                nl = [j for j, ln in enumerate(lines) if ln.startswith("v_mfma")][4 * G2 - 1]
                lines, waits = lines[:nl + 1], _stream(lines[:nl + 1])[2]
            assert cnt(r"v_mfma_f32_16x16x32_bf16") == 4 * G2 and cnt(r"v_mfma_.*") == 4 * G2
            assert cnt(r"s_barrier") == 2, (nw, mode, which)
            assert cnt(r"scratch_.*|buffer_.*|global_.*|flat_.*|ds_write.*|v_accvgpr.*") == 0, (nw, mode, which)   # reads and MFMAs only
            assert cnt(r"v_mov_b32.*") == 0, (nw, mode, which)   # a compiler copy of a fragment register would read it before it lands
            if which == 0:
                assert cnt(r"ds_read_b128") == 2 * nw + 8, (nw, mode)          # W fragments of both k-halves + 2 x 4 X fragments
                # the waits, in order, are exactly the independent replay's counts (the compiler adds none: its own would show up here)
                assert waits == sched(nw), ((nw, mode), waits, sched(nw))
            else:
                # nothing of a next stage behind group GB: W fragments up to the stage's own last one, X of k-half 1
                assert cnt(r"ds_read_b128") == (G2 - R) + 4, (nw, mode, cnt(r"ds_read_b128"))
                assert waits[:GB + 1] == sched(nw)[:GB + 1] and all(w == 0 for w in waits[GB + 1:]), ((nw, mode), waits)
            assert sum(1 for ln in lines if ln.startswith("s_waitcnt")) == G2, (nw, mode, which)
            # one group = its wait, then four MFMAs back to back: the first instruction after every wait is an MFMA (or a pad nop)
            seq = [i for i in ins if i.startswith(("s_waitcnt", "v_mfma", "ds_read", "s_barrier"))]
            for k, i in enumerate(seq):
                if i == "s_waitcnt":
                    assert seq[k + 1:k + 5] == ["v_mfma_f32_16x16x32_bf16"] * 4, ((nw, mode, which), k, seq[k:k + 6])


def test_loader_dma_counts_and_register_budget(listing):
    for (nw, mode, _), body in listing.items():
        dma = len(re.findall(r"global_load_lds_dwordx4", body))
        assert dma > 0 and len(re.findall(r"global_load_lds_dword\b", body)) == 0
        # waits of the per-stage DMA chain: vmcnt(G) with G = the wave's DMA instructions per stage (mixed loaders: 8 + BN / 32;
        # halo weight loaders: BN / 16), never a compiler vmcnt(0) inside the loop other than the kernel's own tail / image waits
        waits = {int(v) for v in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)}
        want = nw if mode == 3 else 8 + nw // 2
        assert want in waits, ((nw, mode), sorted(waits))
        m = re.search(r"\.vgpr_count:\s+(\d+)", body)
        if m:
            assert int(m.group(1)) <= 128
        # 512 threads: two waves per SIMD -> 256 registers per lane in all
        # No scratch access anywhere: a scratch RELOAD's vmcnt(0) also waits for every global store before it, i.e. it would
        # drain a finished tile's stores in front of the next tile (the epilogue is written around this: LDS bias, explicit
        # v_accvgpr_read, per-unit coordinates from the lane id)
        assert len(re.findall(r"scratch_", body)) == 0, (nw, mode)
