#!/bin/bash
# A/B: phase-decomposed Upsample conv / Downsample data gradient (CTRLORA_CONV_PHASE=1, default) against the nine-tap modes (=0):
# parity suites first, then alternating training-step and DDIM benches on the same box.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_phase; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_conv_phase.py tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_gpu_parity_r3.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python tools/time_conv_phase.py 2>&1 | grep -v Warning > $O/time_conv_phase.log
for i in 1 2 3; do for w in 0 1; do
  CTRLORA_CONV_PHASE=$w timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --no-ddim > $O/bench_train_phase${w}_$i.log 2>> $O/err.log
done; done
for i in 1 2; do for w in 0 1; do CTRLORA_CONV_PHASE=$w timeout 400 python bench.py --ddim-only --ddim-core-only > $O/bench_ddim_phase${w}_$i.log 2>> $O/err.log; done; done
for f in $O/bench_train_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
for f in $O/bench_ddim_*.log; do echo $f $(grep -o '"value": [0-9.]*' $f | head -1); done
