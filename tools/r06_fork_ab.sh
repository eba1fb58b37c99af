#!/bin/bash
# A/B: the (t, c)-only products of both networks before (default) / after (CTRLORA_FORK_EARLY=1) the forward's stream fork.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_fork; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -x -q -k "not attention and not lora_fused and not groupnorm" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2 3; do for w in 1 0; do
  CTRLORA_FORK_EARLY=$w timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --no-ddim > $O/bench_train_early${w}_$i.log 2>> $O/err.log
done; done
for w in 1 0; do CTRLORA_FORK_EARLY=$w timeout 400 python bench.py --ddim-only --ddim-core-only > $O/bench_ddim_early${w}.log 2>> $O/err.log; done
for f in $O/bench_train_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
for f in $O/bench_ddim_*.log; do echo $f $(grep -o '"value": [0-9.]*' $f | head -1); done
bash tools/r06_gaps.sh CTRLORA_FORK_EARLY=0 fork_late 2>&1 | tail -3
