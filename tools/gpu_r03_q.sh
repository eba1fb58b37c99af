#!/bin/bash
# per-wave software-pipelined attention forward (variant 15) against the hybrid kernel: correctness + interleaved timing
mkdir -p gpurun_out/r03_q
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 60 python tests/tools/attn_bench.py --variants 14,15 --rounds 5 --shapes "40,4096,4096,8;40,1024,1024,8" --out gpurun_out/r03_q/fwd_wave_pipeline.json > gpurun_out/r03_q/attn.log 2>&1
tail -5 gpurun_out/r03_q/attn.log | cut -c1-700
