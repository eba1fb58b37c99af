#!/bin/bash
# round-2 closing visit (second): full GPU suite, smoke, default bench, segmented-graph bench, steady-state kernel profile
mkdir -p gpurun_out/r02_final2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_final2
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v Warning | tail -8 > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-3000
timeout 600 python bench.py --force-split-graphs --no-cpu-baseline --no-ddim --no-vae > $O/bench_segmented.log 2>&1; tail -1 $O/bench_segmented.log | cut -c1-900
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o train -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim --no-vae > $O/prof.log 2>&1
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/prof_summary.py $DB --steady adamw_dev_kernel 8 > $O/train_kernel_stats.txt 2>&1
head -50 $O/train_kernel_stats.txt
rm -rf $O/prof
