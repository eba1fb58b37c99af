#!/bin/bash
mkdir -p gpurun_out/r02f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f
timeout 300 python tests/tools/attn_bench.py --phase-profile > $O/phase_profile.log 2>&1; cat $O/phase_profile.log | cut -c1-800
timeout 600 python tests/tools/attn_bench.py --bwd --variants 1,0,2 --shapes "40,4096,4096,8;80,1024,1024,8;80,1024,1024,32" --out $O/attn_ab.json > $O/attn_ab.log 2>&1
cat $O/attn_ab.log | cut -c1-1400
timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu -x -k "attention" 2>&1 | grep -v Warning | tail -8 > $O/pytest_attention.log; tail -4 $O/pytest_attention.log
