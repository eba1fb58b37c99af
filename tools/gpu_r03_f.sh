#!/bin/bash
# round 3, visit f: permlane16_swap semantics, hybrid 32x32 Q.K^T forward vs the default (correctness vs fp64 + timing),
# test_gpu_parity.py alone in a fresh process (order-dependent failure seen once)
mkdir -p gpurun_out/r03_f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_f
timeout 60 build/probe_swap > $O/probe_swap.log 2>&1; cat $O/probe_swap.log | cut -c1-400
timeout 600 python tests/tools/attn_bench.py --variants 0,13,14 --shapes "40,4096,4096,8;80,1024,1024,8;40,1024,1024,2" --out $O/attn_hybrid.json 2>&1 | tail -3 | cut -c1-1200
timeout 300 python tests/tools/attn_bench.py --variants 0,13,14 --no-check --shapes "40,4096,4096,32;80,1024,1024,32" --out $O/attn_hybrid_b32.json 2>&1 | tail -2 | cut -c1-800
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider > $O/pytest_parity_alone.log 2>&1; grep -v "Warning\|warnings.warn" $O/pytest_parity_alone.log | grep -E "passed|failed|^E |Error|^FAILED" | tail -30
