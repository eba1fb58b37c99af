#!/bin/bash
# batched UNet emb / context-KV products: A/B; then the round's closing run: full GPU suite, smoke, default bench,
# dominant-shape probe under rocprofv3 --stats (bench's HIP-event figure vs the profiler's), steady-state kernel profile
mkdir -p gpurun_out/r02_final4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_final4
B="python bench.py --no-cpu-baseline --no-vae --no-ddim --steps 20"
CTRLORA_BATCH_EMB=0 timeout 600 $B > $O/bench_unbatched.log 2>&1; tail -1 $O/bench_unbatched.log | cut -c1-200
timeout 600 $B > $O/bench_batched.log 2>&1; tail -1 $O/bench_batched.log | cut -c1-200
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v Warning | tail -8 > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1800
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o probe --output-format csv -- python bench.py --probe-only > $O/probe_profiled.json 2> $O/probe_profiled.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/dominant_kernel_stats.csv 2>/dev/null
head -4 $O/dominant_kernel_stats.csv | cut -c1-250; tail -1 $O/probe_profiled.json | cut -c1-600
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o train -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim --no-vae > $O/prof.log 2>&1
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/prof_summary.py $DB --steady adamw_dev_kernel 8 > $O/train_kernel_stats.txt 2>&1
head -30 $O/train_kernel_stats.txt
rm -rf $O/prof
