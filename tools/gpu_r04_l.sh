#!/bin/bash
# Round 4, visit L: hoisted emb_layers backward -- whole-model gradient parity (tiny + SD1.5 width, rank 128 and rank 32,
# segmented / data-parallel forms), then a same-box A/B of the step with the switch off / on.
mkdir -p gpurun_out/r04_l
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_l
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r4.py -q -x -m gpu > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -q -x -m gpu -k "sd15_latent64 or graphed_two_stream" > $O/pytest_b.log 2>&1; tail -3 $O/pytest_b.log
for v in 0 1 0 1; do
  CTRLORA_HOIST_EMB_BWD=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ddim --no-vae > $O/bench_hoist${v}_$RANDOM.log 2>&1
  echo "hoist=$v $(cat $O/bench_hoist${v}_*.log | grep -o '"ms_per_step": [0-9.]*' | head -20 | tr '\n' ' ')"
done
