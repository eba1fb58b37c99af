#!/bin/bash
# The full GPU suite as the driver runs it, smoke, the default bench line, steady-state kernel traces (training step,
# DDIM loop proper), the dominant-kernel probe under rocprofv3 --stats.  Outputs -> gpurun_out/final (copied to profiles/).
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
rm -f gpurun_out/parity_measured.jsonl
# $2 = "subset": only what changed since the last full run of the suite (the full run is 13 minutes)
if [ "$2" = "subset" ]; then
  timeout 1500 python -m pytest tests/test_gpu_scripts.py tests/test_gpu_parity_r4.py tests/test_gpu_bench_shapes.py -x -q -m gpu --durations=8 -k "scripts or r4 or pure_function or optimizer_state or finetune or pretraining or rank32 or reuse_graph or rccl" > $O/pytest_gpu_subset.log 2>&1; tail -4 $O/pytest_gpu_subset.log
else
  timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
fi
cp gpurun_out/parity_measured.jsonl $O/parity_measured.jsonl 2>/dev/null
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-300
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-400
if [ "$1" != "noprof" ]; then
rm -rf $O/trace_train
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_train -o train -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-ddim --no-vae > $O/trace_train.log 2>&1
python tools/prof_summary.py $(find $O/trace_train -name "*results.db" | head -1) --steady adamw_dev_kernel 4 > $O/train_kernel_stats_steady.txt 2>&1
head -12 $O/train_kernel_stats_steady.txt | cut -c1-170
rm -rf $O/trace_ddim
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_ddim -o ddim -- python bench.py --ddim-only --ddim-core-only --ddim-loops 1 --ddim-warm 2 > $O/trace_ddim.log 2>&1
python tools/prof_summary.py $(find $O/trace_ddim -name "*results.db" | head -1) --steady ddim_step_dev_kernel 40 > $O/ddim_kernel_stats_steady.txt 2>&1
head -8 $O/ddim_kernel_stats_steady.txt | cut -c1-170
rm -rf $O/prof_probe
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_probe -o probe --output-format csv -- python bench.py --probe-only > $O/probe_profiled.json 2> $O/probe_profiled.err
find $O/prof_probe -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/dominant_kernel_stats.csv
head -2 $O/dominant_kernel_stats.csv | cut -c1-200
timeout 400 python bench.py --pretrain-only > $O/bench_pretrain.log 2>&1; tail -1 $O/bench_pretrain.log | cut -c1-300
fi
find $O -name "*.db" -delete; rm -rf $O/trace_train $O/trace_ddim $O/prof_probe; du -sh $O
