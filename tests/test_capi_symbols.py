"""The C-ABI library builds, loads and exports every symbol include/ctrlora_hip.h declares, and the
ctypes signatures agree with the header (argument counts and pointer/int/float classes).
No kernel is launched here (no GPU needed)."""
import ctypes
import os
import re

import pytest

from tests.util import ROOT


def _header_decls(path=("include", "ctrlora_hip.h")):
    src = open(os.path.join(ROOT, *path)).read()
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|long)\s+(cl_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).split(",")]
        if args == ["void"] or args == [""]:
            args = []
        decls[m.group(2)] = (m.group(1), args)
    return decls


def _cls(arg: str) -> str:
    if "*" in arg:
        return "ptr"
    base = arg.split()[:-1]
    if "float" in base:
        return "float"
    if "long" in base:
        return "long"
    return "int"


@pytest.fixture(scope="module")
def built_lib():
    from ctrlora_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(built_lib):
    decls = _header_decls()
    assert len(decls) >= 30
    L = ctypes.CDLL(built_lib)
    for name in decls:
        assert hasattr(L, name), f"{name} declared in ctrlora_hip.h but not exported"
    assert L.cl_abi_version() == 7 and "#define CL_ABI_VERSION 7" in open(os.path.join(ROOT, "include", "ctrlora_hip.h")).read()


def test_ctypes_signatures_match_header(built_lib):
    from ctrlora_amd import hip
    decls = _header_decls()
    assert set(hip.EXPORTED) == set(decls), set(hip.EXPORTED) ^ set(decls)
    cmap = {ctypes.c_void_p: "ptr", ctypes.c_long: "long", ctypes.c_int: "int", ctypes.c_float: "float"}
    for name, (ret, args) in decls.items():
        sig = hip._SIGS[name]
        assert len(sig) == len(args), (name, len(sig), len(args))
        for a, s in zip(args, sig):
            want = _cls(a)
            got = cmap.get(s, "ptr")
            assert want == got, (name, a, s)


def test_probe_hooks_live_outside_the_boundary_header(built_lib):
    """The A/B switches (csrc/debug_hooks.h) are exported and bound, are NOT declared in include/ctrlora_hip.h, and an
    unknown attention schedule code is refused instead of silently selecting something."""
    from ctrlora_amd import hip
    dbg = _header_decls(("ctrlora_amd", "csrc", "debug_hooks.h"))
    assert set(dbg) == set(hip._DEBUG_SIGS) and not (set(dbg) & set(_header_decls()))
    L = ctypes.CDLL(built_lib)
    for name, (ret, args) in dbg.items():
        assert hasattr(L, name) and len(args) == len(hip._DEBUG_SIGS[name])
    assert L.cl_debug_attention_variant(9999) != 0 and L.cl_debug_attention_variant(0) == 0
    assert not hasattr(L, "cl_attention_force_variant")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ctrlora_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(hip.HipError):
        hip.lib()


def test_measured_gemm_table_is_well_formed_and_registers():
    """ctrlora_amd/gemm_tuned_gfx950.json (tools/gemm_autotune.py): every row is a product signature + a tile
    configuration the launcher knows + a split factor; hip.lib() registers all of them through cl_gemm_tune_set."""
    import json
    from ctrlora_amd import hip
    with open(hip.GEMM_TABLE_PATH) as f:
        tab = json.load(f)
    rows = tab["entries"]
    assert rows and len({tuple(r[:7]) for r in rows}) == len(rows)
    for dtype, mode, M, N, K1, K2, geglu, cfg, sk in rows:
        assert dtype in (hip.BF16, hip.F32) and 0 <= mode <= 5 and M > 0 and N > 0 and K1 > 0 and K2 >= 0
        assert geglu in (0, 1) and 0 <= cfg <= 36 and 0 <= sk <= 64
        assert not (geglu and cfg not in (2, 8, 10, 12, 14, 16, 18, 20, 23, 25, 27, 29))
    L = hip.lib()
    assert hip.load_gemm_table(hip.GEMM_TABLE_PATH) == len(rows) == L.cl_gemm_tune_size()
    assert hip.load_gemm_table("") == 0 == L.cl_gemm_tune_size()          # A/B switch: built-in rules only
    assert L.cl_gemm_tune_set(0, 0, 64, 64, 64, 0, 0, 99, 0) != 0           # unknown configuration refused
    hip.load_gemm_table(hip.GEMM_TABLE_PATH)


def test_ctypes_structs_have_the_layout_the_c_compiler_gives_the_header(tmp_path):
    """cl_gemm_params / cl_wgrad_desc are passed BY POINTER / as a host array across the boundary: the ctypes mirrors in
    ctrlora_amd/hip.py must agree with the header on every field's name, offset and on the total size -- checked against gcc's
    own offsetof over include/ctrlora_hip.h (the header is plain C; a field renamed on one side only fails to compile here)."""
    import shutil
    import subprocess
    from ctrlora_amd import hip
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    pairs = [("cl_gemm_params", hip.GemmParams), ("cl_wgrad_desc", hip.WgradDesc)]
    lines = ["#include <stddef.h>", "#include <stdio.h>", '#include "ctrlora_hip.h"', "int main(void) {"]
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu %zu\\n", offsetof({cname}, {fname}), sizeof((({cname}*)0)->{fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    got = {}
    for ln in out:
        w = ln.split()
        if len(w) == 3 and w[1] == "size":
            got[(w[0], None)] = int(w[2])
        elif len(w) == 4:
            got[(w[0], w[1])] = (int(w[2]), int(w[3]))
    for cname, cls in pairs:
        assert got[(cname, None)] == ctypes.sizeof(cls), (cname, got[(cname, None)], ctypes.sizeof(cls))
        for fname, ftype in cls._fields_:
            f = getattr(cls, fname)
            assert got[(cname, fname)] == (f.offset, ctypes.sizeof(ftype)), (cname, fname, got[(cname, fname)], f.offset)
    # and the header has no field the mirror lacks: same field COUNT per struct
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ctrlora_hip.h")).read(), flags=re.S)
    for cname, cls in pairs:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, flags=re.S).group(1)
        nfields = sum(len(decl.split(",")) for decl in body.split(";") if decl.strip())
        assert nfields == len(cls._fields_), (cname, nfields, len(cls._fields_))
