#!/bin/bash
mkdir -p gpurun_out/r02i
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i
timeout 600 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py -q -m gpu -x -k "attention or whole or engine_forward" 2>&1 | grep -v Warning | tail -6 > $O/pytest_attention.log; tail -3 $O/pytest_attention.log
timeout 600 python tests/tools/attn_bench.py --bwd --variants 1,0,3 --shapes "40,4096,4096,8;80,1024,1024,8;40,4096,4096,32" --out $O/attn_ab.json > $O/attn_ab.log 2>&1
cat $O/attn_ab.log | cut -c1-1600
