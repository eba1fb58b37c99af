#!/bin/bash
# Round 4, visit J: refreshed contraction census of the training step, PMC passes on the two attention-backward kernels,
# steady-state kernel trace of the DDIM loop proper (no cold / image-hint legs).
mkdir -p gpurun_out/r04_j
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_j
timeout 400 python tools/gemm_census.py > $O/gemm_census_train.txt 2> $O/gemm_census_train.err
head -5 $O/gemm_census_train.txt
bash tools/pmc_kernel.sh attn_bwd_dkv r04_j/pmc_attn_bwd_dkv -- python tests/tools/attn_bench.py --bwd --variants 0p --rounds 1 --no-check --shapes "40,4096,4096,8" > $O/pmc_attn_bwd_dkv.txt 2>&1
bash tools/pmc_kernel.sh attn_bwd_dq r04_j/pmc_attn_bwd_dq -- python tests/tools/attn_bench.py --bwd --variants 0p --rounds 1 --no-check --shapes "40,4096,4096,8" > $O/pmc_attn_bwd_dq.txt 2>&1
tail -22 $O/pmc_attn_bwd_dkv.txt; tail -22 $O/pmc_attn_bwd_dq.txt
rm -rf $O/trace_ddim
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_ddim -o ddim -- python bench.py --ddim-only --ddim-core-only --ddim-loops 1 --ddim-warm 2 > $O/trace_ddim.log 2>&1
python tools/prof_summary.py $(find $O/trace_ddim -name "*results.db" | head -1) --steady ddim_step_dev_kernel 40 > $O/ddim_kernel_stats_steady.txt 2>&1
head -30 $O/ddim_kernel_stats_steady.txt | cut -c1-160
tail -2 $O/trace_ddim.log | cut -c1-600
find $O -name "*.db" -delete; rm -rf $O/pmc_attn_bwd_dkv/p*/*.db; du -sh $O
