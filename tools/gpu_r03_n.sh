#!/bin/bash
mkdir -p gpurun_out/r03_n
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_scripts.py -x -q -s -m gpu -k finetune_script > gpurun_out/r03_n/f4_timing.log 2>&1; grep "\[f4" gpurun_out/r03_n/f4_timing.log; tail -2 gpurun_out/r03_n/f4_timing.log | cut -c1-200
