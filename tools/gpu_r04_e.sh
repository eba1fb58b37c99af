#!/bin/bash
# Round 4, GPU visit E: the product state after the attention work: attention / GroupNorm GPU tests, step A/B of the pre-scaled q.
mkdir -p gpurun_out/r04_e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity_r3.py -q -k "attention or groupnorm_one_launch or grouped" > gpurun_out/r04_e/pytest.log 2>&1; tail -4 gpurun_out/r04_e/pytest.log
for sw in 1 0; do
  CTRLORA_PRESCALE_Q=$sw timeout 300 python bench.py --steps 20 --warmup 5 --no-vae --no-cpu-baseline > gpurun_out/r04_e/bench_prescale$sw.log 2> gpurun_out/r04_e/bench_prescale$sw.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_e/bench_prescale$sw.log").read().strip().splitlines()[-1])
    print("prescale=$sw", d["value"], "img/s", d["ms_per_step"], "ms  loss", d["loss"], " attention family", d["roofline"].get("attention_family"), "ddim", d.get("ddim", {}).get("value"))
except Exception as ex:
    print("bench prescale=$sw failed", ex); print(open("gpurun_out/r04_e/bench_prescale$sw.err").read()[-2000:])
PY
done
