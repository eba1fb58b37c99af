#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final2; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_measured.jsonl
timeout 1400 python -m pytest tests/ -x -q -m gpu --durations=12 > $O/pytest_gpu.log 2>&1; tail -20 $O/pytest_gpu.log | grep "passed\|failed\|s call" | head -16
cp gpurun_out/parity_measured.jsonl $O/parity_measured.jsonl 2>/dev/null
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-200
timeout 400 python bench.py --pretrain-only --batch 4 > $O/bench_pretrain_b4.log 2>&1; tail -1 $O/bench_pretrain_b4.log | cut -c1-300
CTRLORA_GEMM_XS=0 timeout 400 python bench.py --pretrain-only --batch 4 > $O/bench_pretrain_b4_xs0.log 2>&1; tail -1 $O/bench_pretrain_b4_xs0.log | cut -c1-300
