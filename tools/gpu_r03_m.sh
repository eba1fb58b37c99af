#!/bin/bash
# closing profile: steady-state kernel stats of the training step with the fill-kernel clears and the batched GN prologue
mkdir -p gpurun_out/r03_m
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_m
timeout 240 rocprofv3 --kernel-trace -d $O/prof -o train -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim --no-vae > $O/prof_train.log 2>&1
DB=$(ls $O/prof/*/train_results.db 2>/dev/null | head -1); [ -z "$DB" ] && DB=$(find $O/prof -name "*.db" | head -1)
python tools/prof_summary.py $DB --steady adamw_dev_kernel 8 > $O/train_kernel_stats_steady.txt 2>&1; head -12 $O/train_kernel_stats_steady.txt | cut -c1-150
grep -E "gn_gapply|gn_bwd_gapply|zero_kernel|gn_gpartial" $O/train_kernel_stats_steady.txt | cut -c1-150
rm -rf $O/prof
