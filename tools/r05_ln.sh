#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_f; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm_xs.py tests/test_gpu_parity_r3.py::test_inference_executor_sd15_latent64_eps_vs_oracle -x -q -m gpu > $O/pytest_ln.log 2>&1; tail -15 $O/pytest_ln.log
for i in 1 2; do for ln in 0 1; do
  CTRLORA_LN_PROLOGUE=$ln timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-ddim > $O/bench_train_ln${ln}_$i.log 2>> $O/bench_err.log
  CTRLORA_LN_PROLOGUE=$ln timeout 400 python bench.py --ddim-only --ddim-core-only > $O/bench_ddim_ln${ln}_$i.log 2>> $O/bench_err.log
done; done
for f in $O/bench_train_ln*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
for f in $O/bench_ddim_ln*.log; do echo $f $(grep -o '"value": [0-9.]*' $f | head -1); done
tail -5 $O/bench_err.log
