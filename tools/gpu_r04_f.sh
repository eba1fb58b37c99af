#!/bin/bash
# Round 4, GPU visit F: the training-time LoRA fold: kernel + step parity + trajectory, then the step A/B.
mkdir -p gpurun_out/r04_f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity_r4.py -q -x > gpurun_out/r04_f/pytest.log 2>&1; tail -15 gpurun_out/r04_f/pytest.log
grep -h "train_fold\|lora_fold_kernel" gpurun_out/parity_measured.jsonl | tail -14
for sw in 1 0; do
  CTRLORA_TRAIN_FOLD=$sw timeout 300 python bench.py --steps 20 --warmup 5 --no-ddim --no-vae --no-cpu-baseline > gpurun_out/r04_f/bench_fold$sw.log 2> gpurun_out/r04_f/bench_fold$sw.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_f/bench_fold$sw.log").read().strip().splitlines()[-1])
    f = d["roofline"]["family"]
    print("fold=$sw", d["value"], "img/s", d["ms_per_step"], "ms  loss", d["loss"], " gemm family ms", f["ms_per_step"], "launches", f["launches_per_step"])
except Exception as ex:
    print("bench fold=$sw failed", ex); print(open("gpurun_out/r04_f/bench_fold$sw.err").read()[-2500:])
PY
done
