#!/bin/bash
# GN prologue change: kernel + model parity subset; a short bench; stage clock of the f4 script test
mkdir -p gpurun_out/r03_l
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l
timeout 120 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity.py -x -q -m gpu -k "groupnorm or (engine_forward_backward and tiny) or segmented" > $O/gn_tests.log 2>&1; tail -2 $O/gn_tests.log
timeout 150 python bench.py --no-ddim --no-cpu-baseline --no-vae --steps 20 --warmup 5 > $O/bench_short.log 2>&1; grep '^{' $O/bench_short.log | cut -c1-260
timeout 330 python -m pytest tests/test_gpu_scripts.py -x -q -s -m gpu -k finetune_script > $O/f4_timing.log 2>&1; grep "\[f4" $O/f4_timing.log; tail -2 $O/f4_timing.log | cut -c1-200
