#!/bin/bash
# measured GEMM launch table: search, then A/B the whole step and the DDIM loop with and without it
mkdir -p gpurun_out/r02_tune
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_tune
rm -f ctrlora_amd/gemm_tuned_gfx950.json
CTRLORA_GEMM_TUNED=0 timeout 900 python tools/gemm_autotune.py --out $O/gemm_tuned_gfx950.json --log $O/autotune.log > $O/autotune.out 2>&1
tail -30 $O/autotune.out; head -40 $O/gemm_census_train_ingraph.txt
[ -f $O/gemm_tuned_gfx950.json ] || exit 1
cp $O/gemm_tuned_gfx950.json ctrlora_amd/gemm_tuned_gfx950.json
CTRLORA_GEMM_TUNED=0 timeout 600 python bench.py --no-cpu-baseline --no-vae > $O/bench_builtin.log 2>&1; tail -1 $O/bench_builtin.log | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline --no-vae > $O/bench_tuned.log 2>&1; tail -1 $O/bench_tuned.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu -x 2>&1 | grep -v Warning | tail -5 > $O/pytest_tuned.log; tail -3 $O/pytest_tuned.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "segmented or virtual_ranks or graph" 2>&1 | grep -v Warning | tail -4 > $O/pytest_segmented.log; tail -2 $O/pytest_segmented.log
