"""Build ctrlora_amd/libctrlora_hip.so (gfx950) in-tree with hipcc.

    python -m ctrlora_amd.build          # incremental (object files cached under build/obj)
    python -m ctrlora_amd.build --force

The .so stays out of git (see .gitignore) but travels to the GPU box with the
working-tree snapshot, so nothing is JIT-compiled at run time.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(HERE, "libctrlora_hip.so")
SOURCES = ["gemm.hip", "gemm_xs.hip", "gemm_w4.hip", "wgrad.hip", "norm.hip", "norm_coop.hip", "elementwise.hip", "attention_fwd.hip", "attention_bwd.hip", "attention_tr.hip", "attention_fwd40.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _headers_mtime() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "ctrlora_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(spath), _headers_mtime())):
        return obj
    cmd = ["hipcc", *FLAGS, "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[ctrlora_amd.build] linked {LIB}")
    elif verbose:
        print(f"[ctrlora_amd.build] up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
