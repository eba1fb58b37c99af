#!/bin/bash
# weight gradients on the side stream: A/B of the captured step, then the parity tests that exercise the backward
mkdir -p gpurun_out/r02_wgrad
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_wgrad
B="python bench.py --no-cpu-baseline --no-vae --no-ddim --steps 20"
CTRLORA_OVERLAP_WGRAD=0 timeout 600 $B > $O/bench_inline.log 2>&1; tail -1 $O/bench_inline.log | cut -c1-200
timeout 600 $B > $O/bench_side.log 2>&1; tail -1 $O/bench_side.log | cut -c1-200
CTRLORA_OVERLAP_WGRAD=0 timeout 600 $B > $O/bench_inline2.log 2>&1; tail -1 $O/bench_inline2.log | cut -c1-200
timeout 600 $B > $O/bench_side2.log 2>&1; tail -1 $O/bench_side2.log | cut -c1-200
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_pretrain.py -q -m gpu -x 2>&1 | grep -v Warning | tail -6 > $O/pytest.log; tail -3 $O/pytest.log
