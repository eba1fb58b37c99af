#!/bin/bash
# Round 4, GPU visit G: -lse / -delta folded into the backward kernels' matrix products (pre-scaled Q, d_head 40).
mkdir -p gpurun_out/r04_g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tests/tools/attn_bench.py --variants 14,0p,1p --rounds 7 --bwd --spike --shapes "40,4096,4096,8;40,4096,77,8;80,1024,1024,8" --out gpurun_out/r04_g/attn.json > gpurun_out/r04_g/attn.log 2>&1
python - <<'PY'
import json
try:
    for e in json.load(open("gpurun_out/r04_g/attn.json")):
        print(e["shape"])
        for k, v in e.items():
            if k != "shape":
                print(f"   {k:26s} fwd {v['fwd_us_median']:7.1f} bwd {v.get('bwd_us_median', 0):7.1f} us {v.get('bwd_tflops', 0):6.1f} TF/s " +
                      " ".join(f"{n[:-4]} {v[n]:.2e}" for n in ("o_err", "lse_err", "dq_err", "dk_err", "dv_err") if n in v))
except Exception as ex:
    print("attn_bench failed:", ex); print(open("gpurun_out/r04_g/attn.log").read()[-3000:])
PY
timeout 500 python -m pytest tests/test_gpu_bench_shapes.py -q -x -k "attention" > gpurun_out/r04_g/pytest.log 2>&1; tail -4 gpurun_out/r04_g/pytest.log
