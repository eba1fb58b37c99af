#!/bin/bash
mkdir -p gpurun_out/r04_d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python tests/tools/attn_bench.py --variants 14,0p,22p --rounds 7 --spike --shapes "40,4096,4096,8;40,4096,4096,32;40,1024,1024,8;40,9216,9216,2" --out gpurun_out/r04_d/x_v3.json > gpurun_out/r04_d/x_v3.log 2>&1
python - <<'PY'
import json
for e in json.load(open("gpurun_out/r04_d/x_v3.json")):
    print(e["shape"])
    for k, v in e.items():
        if k != "shape": print(f"   {k:40s} fwd {v['fwd_us_median']:8.1f} us  {v['fwd_tflops']:7.1f} TF/s o_err {v.get('o_err', 0):.3e} lse_err {v.get('lse_err', 0):.3e}")
PY
