#!/bin/bash
# second A/B of the phase forms (now including the source-grid Upsample data gradient), training step only, + timing table
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_phase; mkdir -p $O
python tools/time_conv_phase.py 2>&1 | grep -v Warning > $O/time_conv_phase.log
for i in 1 2 3; do for w in 0 1; do
  CTRLORA_CONV_PHASE=$w timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --no-ddim > $O/bench2_train_phase${w}_$i.log 2>> $O/err.log
done; done
for f in $O/bench2_train_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r4.py -x -q > $O/pytest2.log 2>&1; tail -2 $O/pytest2.log
