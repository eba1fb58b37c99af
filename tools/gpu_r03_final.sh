#!/bin/bash
# round 3 closing run: exactly what the driver runs (GPU suite with -x, smoke, default bench), then the steady-state kernel
# profile of the training step on this box
mkdir -p gpurun_out/r03_final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final
rm -f gpurun_out/parity_measured.jsonl
timeout 1800 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|^FAILED|^E " $O/pytest_gpu.log | tail -8
cp gpurun_out/parity_measured.jsonl $O/ 2>/dev/null
timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-4000
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o train -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim --no-vae > $O/prof_train.log 2>&1
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/prof_summary.py $DB --steady adamw_dev_kernel 8 > $O/train_kernel_stats_steady.txt 2>&1; head -40 $O/train_kernel_stats_steady.txt | cut -c1-150
rm -rf $O/prof
