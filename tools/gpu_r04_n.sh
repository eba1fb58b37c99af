#!/bin/bash
# Round 4, visit N: attn_fwd40_kernel repeatability (synthetic, sliced, second pass, next to a GEMM stream, model data).
mkdir -p gpurun_out/r04_n
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tests/tools/debug_fwd40_determinism.py > gpurun_out/r04_n/fwd40_determinism.log 2>&1
grep "^\[\|captured\|call" gpurun_out/r04_n/fwd40_determinism.log | head -60
