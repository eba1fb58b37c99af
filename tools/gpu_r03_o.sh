#!/bin/bash
# hybrid attention forward after the K-row permutation (LDS bank conflicts), interleaved against the round-2 form
mkdir -p gpurun_out/r03_o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 100 python tests/tools/attn_bench.py --variants 12,14,13 --rounds 7 --shapes "40,4096,4096,8;80,1024,1024,8" --out gpurun_out/r03_o/interleaved_fwd_after_krow_perm.json > gpurun_out/r03_o/attn.log 2>&1
tail -12 gpurun_out/r03_o/attn.log | cut -c1-220
