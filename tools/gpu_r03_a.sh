#!/bin/bash
# round 3, first visit: the new parity tests (configs[4] vs oracle, fold, conv-tap wgrad, SD1.5-width pre-training),
# the bench-shape tests under the tightened bf16 gates, and a bench line with the new fields
mkdir -p gpurun_out/r03_a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_a
rm -f gpurun_out/parity_measured.jsonl
timeout 1500 python -m pytest tests/test_gpu_parity_r3.py -q -x 2>&1 | grep -v Warning | tail -25 > $O/pytest_r3.log; tail -12 $O/pytest_r3.log
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -q -k "golden or graphed" 2>&1 | grep -v Warning | tail -12 > $O/pytest_shapes.log; tail -5 $O/pytest_shapes.log
cp gpurun_out/parity_measured.jsonl $O/ 2>/dev/null
timeout 900 python bench.py --no-cpu-baseline --steps 10 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-3000
