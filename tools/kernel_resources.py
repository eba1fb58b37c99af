#!/usr/bin/env python3
"""Register / LDS / scratch budget of every compiled kernel, read from hipcc's own listing (no GPU needed).

    python tools/kernel_resources.py [--stats profiles/r05_final/train_kernel_stats_steady.txt ...] [--all] [--out FILE]

For each csrc/*.hip the device listing (`hipcc -S --cuda-device-only`, the flags of ctrlora_amd/build.py) ends in the
`amdhsa.kernels` metadata: VGPRs (on gfx90a+ `.vgpr_count` is the UNIFIED count, accumulation offset + AGPRs), AGPRs, SGPRs,
spill counts, static LDS, scratch bytes, workgroup size.  From them: unified registers per lane = align8(vgpr_count)
(gfx950: 512 per SIMD lane, so waves/SIMD = min(8, 512 // unified)),
and the LDS-limited workgroups per CU for kernels with a static LDS segment (160 KB per CU; kernels that size their LDS at launch
show `dyn`).  With --stats (the per-kernel tables tools/prof_summary.py writes) only the kernels that ran in those steps are
listed, joined with their share of the step, so the table reads as "what the step's time runs at"; --all lists every kernel.
A kernel with scratch or spills is flagged `!`: the hot kernels are expected to have none (tests/test_kernel_resources.py)."""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctrlora_amd import build as _build                                                   # noqa: E402

CXXFILT = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
LDS_PER_CU = 160 * 1024
UNIFIED_REGS = 512
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
          "private_segment_fixed_size", "max_flat_workgroup_size", "uses_dynamic_stack")


def listing(src: str, outdir: str) -> str:
    out = os.path.join(outdir, src.replace(".hip", ".s"))
    spath = os.path.join(_build.CSRC, src)
    newest = max([os.path.getmtime(spath)] + [os.path.getmtime(os.path.join(_build.CSRC, h)) for h in os.listdir(_build.CSRC) if h.endswith(".h")])
    if os.path.exists(out) and os.path.getmtime(out) >= newest:          # a kept listing (--keep DIR) newer than source and headers
        return out
    flags = [f for f in _build.FLAGS if f != "-fPIC"]
    r = subprocess.run(["hipcc", *flags, "-S", "--cuda-device-only", os.path.join(_build.CSRC, src), "-o", out],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc -S failed for {src}:\n{r.stderr[-2000:]}")
    return out


def parse_metadata(text: str) -> list[dict]:
    """The `amdhsa.kernels:` YAML sequence at the end of a listing -> one dict per kernel (only FIELDS + name)."""
    i = text.rfind("amdhsa.kernels:")
    if i < 0:
        return []
    end = text.find("amdhsa.target:", i)
    body = text[i:end if end > 0 else len(text)]
    kernels = []
    for block in re.split(r"\n  - ", body)[1:]:
        k = {}
        m = re.search(r"(?:^|\n)\s{0,4}\.name:\s+(\S+)", block)
        if not m:
            continue
        k["symbol"] = m.group(1)
        for f in FIELDS:
            m = re.search(rf"(?:^|\n)\s{{0,4}}\.{f}:\s+(\S+)", block)
            if m:
                k[f] = int(m.group(1)) if m.group(1).lstrip("-").isdigit() else (1 if m.group(1) == "true" else 0)
        kernels.append(k)
    return kernels


def demangle(symbols: list[str]) -> list[str]:
    if not symbols:
        return []
    exe = CXXFILT if os.path.exists(CXXFILT) else "c++filt"
    r = subprocess.run([exe], input="\n".join(symbols) + "\n", capture_output=True, text=True, check=True)
    return r.stdout.splitlines()


def short_name(demangled: str) -> str:
    """`void cl::(anonymous namespace)::gemm_fl_kernel<unsigned short, 256, ...>(cl::GemmParams, int)` -> `gemm_fl_kernel<...>`
    (the form rocprofv3 / tools/prof_summary.py print)."""
    s = demangled
    depth, cut = 0, len(s)
    for j in range(len(s) - 1, -1, -1):                                                  # drop the trailing parameter list
        if s[j] == ")":
            depth += 1
        elif s[j] == "(":
            depth -= 1
            if depth == 0:
                cut = j
                break
    s = s[:cut] if s.endswith(")") else s
    s = re.sub(r"^void\s+", "", s)
    s = s.replace("(anonymous namespace)::", "")
    s = re.sub(r"^(?:\w+::)+", "", s)
    s = re.sub(r"\(ctrlora_dtype\)(\d+)", r"\1", s)
    return s.strip()


def align(x: int, a: int) -> int:
    return (x + a - 1) // a * a


def derive(k: dict) -> dict:
    v, a = k.get("vgpr_count", 0), k.get("agpr_count", 0)
    uni = align(v, 8)                                               # .vgpr_count already counts the AGPRs behind the accum offset
    k["unified"], k["arch_vgpr"] = uni, v - a
    k["waves_simd"] = min(8, UNIFIED_REGS // uni) if uni else 8
    lds = k.get("group_segment_fixed_size", 0)
    k["lds_wg_cu"] = (LDS_PER_CU // lds) if lds else None
    k["flag"] = bool(k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0) or k.get("sgpr_spill_count", 0)
                     or k.get("uses_dynamic_stack", 0))
    return k


def collect(sources=None, outdir=None) -> list[dict]:
    sources = sources or [s for s in _build.SOURCES if s != "capi.hip"]
    own = outdir is None
    outdir = outdir or tempfile.mkdtemp(prefix="isa_")
    os.makedirs(outdir, exist_ok=True)
    with ThreadPoolExecutor(max_workers=6) as ex:
        paths = list(ex.map(lambda s: listing(s, outdir), sources))
    rows = []
    for src, p in zip(sources, paths):
        with open(p) as f:
            ks = parse_metadata(f.read())
        for k, d in zip(ks, demangle([k["symbol"] for k in ks])):
            k["source"], k["name"] = src, short_name(d)
            rows.append(derive(k))
        if own:
            os.remove(p)
    return rows


def read_stats(path: str) -> dict[str, tuple[int, float, float]]:
    """name -> (calls, total_ms, percent) from a tools/prof_summary.py table."""
    out = {}
    with open(path) as f:
        for line in f:
            m = re.match(r"^(\S.*?)\s{2,}(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
            if m and not line.startswith("name "):
                out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(5)))
    return out


def table(rows: list[dict], stats: dict | None, label: str) -> str:
    hdr = f"{'kernel':<72} {'src':<16} {'vgpr':>4} {'agpr':>4} {'uni':>4} {'w/SIMD':>6} {'sgpr':>4} {'LDS B':>7} {'WG/CU':>5} {'scr B':>5} {'spill':>5}"
    if stats is not None:
        hdr += f" {'calls':>6} {'% step':>6}"
    lines = [label, hdr]
    by_name = {}
    for k in rows:
        by_name.setdefault(k["name"], k)
    if stats is not None:
        for n in stats:                                  # a trace older than a template parameter: `k<4>` is today's `k<4, 32>`
            if n not in by_name and n.endswith(">"):
                later = sorted(m for m in by_name if m.startswith(n[:-1] + ","))
                if later:                                # several: the first (the added parameter's old value was its smallest)
                    by_name[n] = by_name[later[0]]
        order = sorted(stats, key=lambda n: -stats[n][1])
        missing = [n for n in order if n not in by_name]
        sel = [(by_name[n], stats[n]) for n in order if n in by_name]
    else:
        missing, sel = [], [(k, None) for k in sorted(rows, key=lambda k: (k["source"], k["name"]))]
    for k, st in sel:
        lds = k.get("group_segment_fixed_size", 0)
        line = (f"{('! ' if k['flag'] else '') + k['name']:<72.72} {k['source']:<16} {k['arch_vgpr']:>4} {k.get('agpr_count', 0):>4} "
                f"{k['unified']:>4} {k['waves_simd']:>6} {k.get('sgpr_count', 0):>4} {(str(lds) if lds else 'dyn/0'):>7} "
                f"{(str(k['lds_wg_cu']) if k['lds_wg_cu'] else '-'):>5} {k.get('private_segment_fixed_size', 0):>5} "
                f"{k.get('vgpr_spill_count', 0) + k.get('sgpr_spill_count', 0):>5}")
        if st is not None:
            line += f" {st[0]:>6} {st[2]:>6.2f}"
        lines.append(line)
    if missing:
        lines.append(f"not in the listings (library kernels: rocBLAS / torch / RCCL): {len(missing)}: " + "; ".join(missing[:12]))
    if stats is not None:
        cov = sum(st[2] for _, st in sel)
        lines.append(f"listed kernels cover {cov:.1f} % of the step's GPU time; flagged (scratch / spills): "
                     f"{sum(1 for k, _ in sel if k['flag'])}")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", nargs="*", default=[])
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--keep", default="", help="directory to keep the listings in")
    a = ap.parse_args()
    rows = collect(outdir=a.keep or None)
    parts = [f"# {len(rows)} kernels in {len({k['source'] for k in rows})} sources; hipcc {' '.join(_build.FLAGS)}; "
             f"gfx950: {UNIFIED_REGS} unified registers per SIMD lane, {LDS_PER_CU // 1024} KB LDS per CU; `dyn/0` = LDS sized at launch (or none)"]
    for s in a.stats:
        parts.append(table(rows, read_stats(s), f"\n## kernels of {s}"))
    if a.all or not a.stats:
        parts.append(table(rows, None, "\n## every kernel"))
    flagged = [k for k in rows if k["flag"]]
    parts.append(f"\n## kernels with scratch / spills / dynamic stack: {len(flagged)} of {len(rows)}")
    for k in flagged:
        parts.append(f"  {k['name']}  ({k['source']}): scratch {k.get('private_segment_fixed_size', 0)} B, "
                     f"vgpr spills {k.get('vgpr_spill_count', 0)}, sgpr spills {k.get('sgpr_spill_count', 0)}")
    text = "\n".join(parts) + "\n"
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
