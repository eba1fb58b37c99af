#!/bin/bash
# attention backward ping-pong A/B + correctness, packed math in the old kernels, attention pytest
mkdir -p gpurun_out/r02d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d
timeout 600 python tests/tools/attn_bench.py --bwd --variants 1,0 --shapes "40,4096,4096,8;80,1024,1024,8;40,4096,4096,1;80,1024,1024,32" --out $O/attn_bwd_ab.json > $O/attn_bwd_ab.log 2>&1
cat $O/attn_bwd_ab.log | cut -c1-1200
timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu -x -k "attention" 2>&1 | grep -v Warning | tail -8 > $O/pytest_attention.log; tail -4 $O/pytest_attention.log
