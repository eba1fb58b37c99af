#!/bin/bash
# offer the 64 x 80 tiles (cfg 35 / 36) and re-offer cfg 34 to every signature; then the full GPU suite with durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_j; rm -rf $O; mkdir -p $O
timeout 900 python tools/gemm_autotune.py --merge ctrlora_amd/gemm_tuned_gfx950.json --retry-cfgs 35,36 --out $O/merged_6480.json --log $O/autotune_6480.log > $O/autotune_6480.out 2>&1
tail -30 $O/autotune_6480.out
timeout 1400 python -m pytest tests/ -q -m gpu --durations=25 > $O/pytest_gpu.log 2>&1; tail -45 $O/pytest_gpu.log
