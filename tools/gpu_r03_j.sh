#!/bin/bash
mkdir -p gpurun_out/r03_j
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_j
for v in DBG_NONE DBG_NO_OVERLAP; do
  env $v=1 REPS=4 timeout 600 python tests/tools/debug_segmented.py f32 segmented > $O/fixed_$v.log 2>&1; echo "== $v"; grep -E "^rep|failing|Error" $O/fixed_$v.log | tail -5 | cut -c1-200
done
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
