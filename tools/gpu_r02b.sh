#!/bin/bash
# round-2 GPU visit: attention A/B, bench-shape parity tests, bench with the autograd-free graph, steady-state profile
mkdir -p gpurun_out/r02b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b
timeout 600 python tests/tools/attn_bench.py --bwd --out $O/attn_bench.json > $O/attn_bench.log 2>&1; tail -8 $O/attn_bench.log | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu -x 2>&1 | grep -v Warning | tail -25 > $O/pytest_bench_shapes.log; tail -4 $O/pytest_bench_shapes.log
timeout 600 python bench.py --no-cpu-baseline --no-ddim > $O/bench.log 2>&1; tail -2 $O/bench.log | cut -c1-400
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o train -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim > $O/prof.log 2>&1
python tools/prof_summary.py $O/prof/*/train_results.db --steady adamw_dev_kernel 8 > $O/train_kernel_stats.txt 2>&1 || python tools/prof_summary.py $O/prof/train_results.db --steady adamw_dev_kernel 8 > $O/train_kernel_stats.txt 2>&1
head -40 $O/train_kernel_stats.txt
rm -rf $O/prof
