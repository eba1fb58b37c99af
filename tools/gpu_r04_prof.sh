#!/bin/bash
# Round 4, profiling visit: steady-state kernel trace of the training step and of the DDIM loop, PMC passes on the new attention
# forward (attn_fwd40_kernel), TCC traffic of the dominant convolution kernel, the dominant-kernel probe under rocprofv3 --stats.
mkdir -p gpurun_out/r04_prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_prof
rm -rf $O/trace_train
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_train -o train -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-ddim --no-vae > $O/trace_train.log 2>&1
python tools/prof_summary.py $(find $O/trace_train -name "*results.db" | head -1) --steady adamw_dev_kernel 4 > $O/train_kernel_stats_steady.txt 2>&1
head -45 $O/train_kernel_stats_steady.txt
rm -rf $O/trace_ddim
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_ddim -o ddim -- python bench.py --ddim-only --ddim-loops 1 --ddim-warm 2 > $O/trace_ddim.log 2>&1
python tools/prof_summary.py $(find $O/trace_ddim -name "*results.db" | head -1) --steady ddim_step_dev_kernel 40 > $O/ddim_kernel_stats_steady.txt 2>&1
head -14 $O/ddim_kernel_stats_steady.txt
# PMC on the attention forward (isolated launches through attn_bench, pre-scaled Q)
bash tools/pmc_kernel.sh attn_fwd40 r04_prof/pmc_attn_fwd40 -- python tests/tools/attn_bench.py --variants 0p --rounds 1 --no-check --shapes "40,4096,4096,8" > $O/pmc_attn_fwd40.txt 2>&1
cat $O/pmc_attn_fwd40.txt | tail -24
# TCC traffic of the dominant kernel
bash tools/build_probes.sh > $O/build_probes.log 2>&1
bash tools/pmc_traffic.sh > $O/pmc_traffic.txt 2>&1; tail -3 $O/pmc_traffic.txt
mkdir -p $O/pmc_traffic_csv; cp gpurun_out/pmc_traffic/*/*counter_collection.csv $O/pmc_traffic_csv/ 2>/dev/null
# dominant kernel under --stats vs bench's HIP events
rm -rf $O/prof_probe
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_probe -o probe --output-format csv -- python bench.py --probe-only > $O/probe_profiled.json 2> $O/probe_profiled.err
tail -1 $O/probe_profiled.json | cut -c1-400
find $O/prof_probe -name "*kernel_stats.csv" | head -1 | xargs -r head -4
rm -rf $O/trace_train/*/*.db $O/trace_ddim/*/*.db 2>/dev/null; find $O -name "*.db" -delete; du -sh $O
