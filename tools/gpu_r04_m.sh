#!/bin/bash
# Round 4, visit M: which switch makes the replayed step non-deterministic (test_graphed_two_stream...: loss differs between replays)?
mkdir -p gpurun_out/r04_m
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_m
run() { tag=$1; shift; env "$@" timeout 280 python tests/tools/debug_determinism.py --tag $tag $EXTRA >> $O/determinism.log 2>&1; tail -8 $O/determinism.log | grep "^\[" | tail -1; }
EXTRA="" run default A=1
EXTRA="" run no_prescale CTRLORA_PRESCALE_Q=0
EXTRA="" run no_hoist CTRLORA_HOIST_EMB_BWD=0
EXTRA="--variant 14" run variant14 A=1
EXTRA="--variant 1" run variant1 A=1
EXTRA="--one-stream" run one_stream A=1
EXTRA="" run no_wgrad_overlap CTRLORA_OVERLAP_WGRAD=0
