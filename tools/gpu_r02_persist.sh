#!/bin/bash
# persistent small-K GEMM probe (bit-exactness vs the one-tile form, timing, s_memtime phase breakdown) + DDIM with merged LoRA
mkdir -p gpurun_out/r02_persist
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_persist
timeout 400 build/probe_gemm_t --persist > $O/probe_persist.log 2>&1; tail -60 $O/probe_persist.log
CTRLORA_MERGE_LORA=0 timeout 600 python bench.py --ddim-only > $O/ddim_unmerged.log 2>&1; tail -1 $O/ddim_unmerged.log | cut -c1-300
timeout 600 python bench.py --ddim-only > $O/ddim_merged.log 2>&1; tail -1 $O/ddim_merged.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -q -m gpu -x -k "ddim or inference or lora or api or sample or bank" 2>&1 | grep -v Warning | tail -5 > $O/pytest_infer.log; tail -3 $O/pytest_infer.log
