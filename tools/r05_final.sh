#!/bin/bash
# Round-5 closing run: the full GPU suite as the driver runs it, smoke, the stock comparator, tagged steady-state traces of the
# training and the DDIM step (per-kernel + per-shape tables), the default bench line, pre-training at batch 8 and 4, the
# dominant-kernel probe under rocprofv3 --stats.  Outputs -> gpurun_out/r05_final (copied to profiles/r05_final).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; rm -rf $O; mkdir -p $O profiles/r05_final
rm -f gpurun_out/parity_measured.jsonl
timeout 1400 python -m pytest tests/ -x -q -m gpu --durations=20 > $O/pytest_gpu.log 2>&1; tail -30 $O/pytest_gpu.log | grep "passed\|failed\|s call" | head -24
cp gpurun_out/parity_measured.jsonl $O/parity_measured.jsonl 2>/dev/null
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -4 $O/smoke.log | cut -c1-200
timeout 600 python tests/tools/compare_stock.py --out $O/r05_compare_precision.json > $O/compare_stock.log 2>&1; tail -3 $O/compare_stock.log | cut -c1-300
cp $O/r05_compare_precision.json profiles/r05_compare_precision.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_train -o train -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-ddim --no-vae --tag-gemm $O/tags_train.json > $O/trace_train.log 2>&1
DB=$(find $O/trace_train -name "*results.db" | head -1)
python tools/prof_summary.py $DB --steady adamw_dev_kernel 4 > $O/train_kernel_stats_steady.txt 2>&1; head -14 $O/train_kernel_stats_steady.txt | cut -c1-170
python tools/prof_shapes.py $DB $O/tags_train.json --steady adamw_dev_kernel 4 --top 60 > $O/train_shapes_in_step.txt 2>&1; head -12 $O/train_shapes_in_step.txt | cut -c1-170
cp $O/train_shapes_in_step.txt profiles/r05_final/train_shapes_in_step.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_ddim -o ddim -- python bench.py --ddim-only --ddim-core-only --ddim-loops 1 --ddim-warm 2 --tag-gemm $O/tags_ddim.json > $O/trace_ddim.log 2>&1
DB=$(find $O/trace_ddim -name "*results.db" | head -1)
python tools/prof_summary.py $DB --steady ddim_step_dev_kernel 40 > $O/ddim_kernel_stats_steady.txt 2>&1; head -10 $O/ddim_kernel_stats_steady.txt | cut -c1-170
python tools/prof_shapes.py $DB $O/tags_ddim.json --steady ddim_step_dev_kernel 40 --top 50 > $O/ddim_shapes_in_step.txt 2>&1
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-600
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_probe -o probe --output-format csv -- python bench.py --probe-only > $O/probe_profiled.json 2> $O/probe_profiled.err
find $O/prof_probe -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/dominant_kernel_stats.csv
head -2 $O/dominant_kernel_stats.csv | cut -c1-200
timeout 400 python bench.py --pretrain-only > $O/bench_pretrain.log 2>&1; tail -1 $O/bench_pretrain.log | cut -c1-300
timeout 400 python bench.py --pretrain-only --batch 4 > $O/bench_pretrain_b4.log 2>&1; tail -1 $O/bench_pretrain_b4.log | cut -c1-300
find $O -name "*.db" -delete; rm -rf $O/trace_train $O/trace_ddim $O/prof_probe; du -sh $O
