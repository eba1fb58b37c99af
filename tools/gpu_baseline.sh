#!/bin/bash
# One GPU-box visit: parity tests, headline bench, kernel-trace profile of the training step.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o train -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-ddim > gpurun_out/prof.log 2>&1
ls -R gpurun_out/prof | head -30
tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench.log
