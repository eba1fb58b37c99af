#!/bin/bash
# Short GPU-box visit: a pytest subset (-k "$1"), the training bench without DDIM / CPU baseline, kernel trace.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "$1" > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_quick.log
tail -3 gpurun_out/pytest_quick.log
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o train -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-ddim > gpurun_out/prof.log 2>&1
tail -1 gpurun_out/prof.log | cut -c1-200
python tools/prof_summary.py gpurun_out/prof/train_results.db > gpurun_out/prof_summary.txt 2>&1
head -24 gpurun_out/prof_summary.txt
grep -h "colsum\|ln_bwd" gpurun_out/prof_summary.txt | sort -u
