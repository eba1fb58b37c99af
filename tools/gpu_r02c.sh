#!/bin/bash
# attention forward A/B + ablations + PMC of the ping-pong kernel
mkdir -p gpurun_out/r02c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c
timeout 300 python tests/tools/attn_bench.py --variants 1,2,0,5,3,4 --shapes "40,4096,4096,8;80,1024,1024,8;80,1024,1024,32" --out $O/attn_fwd_ab.json > $O/attn_fwd_ab.log 2>&1
cat $O/attn_fwd_ab.log | cut -c1-1500
bash tools/pmc_kernel.sh attn_fwd_pp r02c/pmc_pp -- python tests/tools/attn_bench.py --no-check --variants 0 --shapes "40,4096,4096,8" > $O/pmc_attn_fwd_pp.txt 2>&1
cat $O/pmc_attn_fwd_pp.txt
