"""Every Python file of the repo passes the scope-aware undefined-name scan (tests/tools/undefined_names.py): GPU-only
code paths and GPU-only tests cannot be executed here, but a misspelt or out-of-scope name in them can be found here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


def test_no_undefined_names_anywhere(tmp_path):
    import undefined_names
    files = subprocess.run(["git", "ls-files", "*.py"], cwd=ROOT, capture_output=True, text=True).stdout.split()
    if not files:                                   # not a git checkout (the GPU box snapshot): walk the tree
        for d, _, fs in os.walk(ROOT):
            if any(s in d for s in (".git", "gpurun_out", "__pycache__", "/build")):
                continue
            files += [os.path.relpath(os.path.join(d, f), ROOT) for f in fs if f.endswith(".py")]
    assert len(files) > 50
    found = undefined_names.scan([os.path.join(ROOT, f) for f in files])
    assert not found, "\n".join(found)
    # the scanner itself: a closure reading a name of a sibling scope, and a method reading a class attribute bare
    bad = tmp_path / "bad.py"
    bad.write_text("def f():\n    def g():\n        return dtype\n    return g\nclass A:\n    x = 1\n    def m(self):\n        return x\n")
    assert len(undefined_names.scan([str(bad)])) == 2
