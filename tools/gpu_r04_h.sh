#!/bin/bash
# Round 4, GPU visit H: the new parity tests (configs[0] at its own shape, sampler graph cache, RCCL world-size 1 next to
# segment graphs) and the re-gated pre-training tests.
mkdir -p gpurun_out/r04_h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity_r4.py tests/test_pretrain.py -q -s > gpurun_out/r04_h/pytest.log 2>&1; grep -v "amdgpu.ids" gpurun_out/r04_h/pytest.log | tail -30
grep -h "rank32\|rccl" gpurun_out/parity_measured.jsonl | tail -4
