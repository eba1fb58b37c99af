#!/bin/bash
mkdir -p gpurun_out/r02e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e
timeout 300 python tests/tools/attn_bench.py --phase-profile > $O/phase_profile.log 2>&1; cat $O/phase_profile.log | cut -c1-800
timeout 900 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-3000
bash tools/pmc_kernel.sh attn_bwd_dkv_pp r02e/pmc_dkv -- python tests/tools/attn_bench.py --bwd --no-check --variants 0 --shapes "40,4096,4096,8" > $O/pmc_attn_dkv_pp.txt 2>&1
cat $O/pmc_attn_dkv_pp.txt
