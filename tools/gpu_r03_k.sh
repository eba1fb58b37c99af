#!/bin/bash
# full GPU suite as the driver runs it, with per-test durations
mkdir -p gpurun_out/r03_k
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu --durations=70 > gpurun_out/r03_k/pytest_gpu_durations.log 2>&1
tail -4 gpurun_out/r03_k/pytest_gpu_durations.log
