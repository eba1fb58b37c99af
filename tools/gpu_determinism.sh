#!/bin/bash
# Is the hot path a pure function of its inputs?  (a) attn_fwd40_kernel: eight launches on the same inputs, synthetic and model
# data, alone and next to a GEMM stream, bitwise compared; (b) the graphed training step at lr = 0 replayed six times under
# every engine switch: loss bits and gradient / weight checksums per replay.  Found the missing data dependence of the fwd40
# drains in round 4 (profiles/r04_determinism/).   usage (GPU box): bash tools/gpu_determinism.sh [all]
mkdir -p gpurun_out/determinism
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/determinism
timeout 600 python tests/tools/debug_fwd40_determinism.py > $O/fwd40.log 2>&1; grep "^\[" $O/fwd40.log
run() { tag=$1; shift; env "$@" timeout 280 python tests/tools/debug_determinism.py --tag $tag $EXTRA >> $O/step.log 2>&1; grep "^\[$tag\]" $O/step.log | tail -1; }
EXTRA="" run default A=1
if [ "$1" = "all" ]; then
  EXTRA="" run no_prescale CTRLORA_PRESCALE_Q=0
  EXTRA="" run no_hoist CTRLORA_HOIST_EMB_BWD=0
  EXTRA="--variant 14" run variant14 A=1
  EXTRA="--variant 1" run variant1 A=1
  EXTRA="--one-stream" run one_stream A=1
  EXTRA="" run no_wgrad_overlap CTRLORA_OVERLAP_WGRAD=0
fi
