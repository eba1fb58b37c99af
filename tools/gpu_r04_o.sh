#!/bin/bash
# Round 4, visit O: after tying the fwd40 drains to the accumulators: kernel repeatability, step determinism, attention tests,
# the graphed-step test that caught it, timing.
mkdir -p gpurun_out/r04_o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_o
timeout 600 python tests/tools/debug_fwd40_determinism.py > $O/fwd40_determinism.log 2>&1
grep "^\[" $O/fwd40_determinism.log | head -30
timeout 280 python tests/tools/debug_determinism.py --tag default > $O/determinism.log 2>&1; grep "^\[" $O/determinism.log
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -q -x -m gpu -k "attention or graphed_two_stream" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 200 python tests/tools/attn_bench.py --variants 14,0p --rounds 5 --shapes "40,4096,4096,8;40,4096,4096,32" --spike > $O/attn_fwd.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r04_o/attn_fwd.log"):
    if l.startswith("{"):
        c=json.loads(l); print(c["shape"], {k:(v.get("fwd_us_median"), v.get("o_err")) for k,v in c.items() if k!="shape"})
PY
