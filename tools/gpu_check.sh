#!/bin/bash
# One GPU-box visit: kernel probes, parity tests, headline bench (no CPU baseline), kernel-trace profile.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -2 gpurun_out/bench.log
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o train -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-ddim > gpurun_out/prof.log 2>&1
python tools/prof_summary.py gpurun_out/prof/train_results.db > gpurun_out/prof_summary.txt 2>&1
head -30 gpurun_out/prof_summary.txt
head -30 gpurun_out/prof_summary.txt
# dominant kernel alone: the rocprofv3 --stats average that bench.py's roofline.ms_per_launch must agree with
rm -rf gpurun_out/prof_probe
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_probe -o probe --output-format csv -- python bench.py --probe-only > gpurun_out/prof_probe.log 2>&1
tail -1 gpurun_out/prof_probe.log | cut -c1-300
find gpurun_out/prof_probe -name "*kernel_stats.csv" | head -1 | xargs -r head -5
