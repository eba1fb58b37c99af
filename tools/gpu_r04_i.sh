#!/bin/bash
# Round 4, GPU visit I: the full-N 128 x 320 tile (launch configuration 33) offered to the measured launch table, then the step with the
# old and the new table.
mkdir -p gpurun_out/r04_i
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp ctrlora_amd/gemm_tuned_gfx950.json gpurun_out/r04_i/table_before.json
timeout 700 python tools/gemm_autotune.py --merge ctrlora_amd/gemm_tuned_gfx950.json --retry-cfgs 33 --budget-s 420 \
   --out gpurun_out/r04_i/table_after.json --log gpurun_out/r04_i/autotune.log > gpurun_out/r04_i/autotune.out 2>&1
tail -25 gpurun_out/r04_i/autotune.out
for tb in before after; do
  CTRLORA_GEMM_TABLE=gpurun_out/r04_i/table_$tb.json timeout 300 python bench.py --steps 20 --warmup 5 --no-vae --no-cpu-baseline > gpurun_out/r04_i/bench_$tb.log 2> gpurun_out/r04_i/bench_$tb.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_i/bench_$tb.log").read().strip().splitlines()[-1])
    f = d["roofline"]["family"]
    print("table=$tb", d["value"], "img/s", d["ms_per_step"], "ms  gemm family ms", f["ms_per_step"], " ddim", d.get("ddim", {}).get("value"))
except Exception as ex:
    print("bench $tb failed", ex); print(open("gpurun_out/r04_i/bench_$tb.err").read()[-2000:])
PY
done
