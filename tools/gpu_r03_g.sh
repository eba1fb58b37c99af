#!/bin/bash
# round 3, visit g: INTERLEAVED A/B of the forward attention variants (round-robin, 7 rounds, median / min)
mkdir -p gpurun_out/r03_g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g
timeout 900 python tests/tools/attn_bench.py --rounds 7 --variants 12,6,7,10,8,13,14 --shapes "40,4096,4096,8;80,1024,1024,8" --out $O/attn_fwd_interleaved.json 2>&1 | tail -2 | cut -c1-2600
timeout 600 python tests/tools/attn_bench.py --rounds 7 --no-check --variants 12,10,13,14 --shapes "40,4096,4096,32;80,1024,1024,32" --out $O/attn_fwd_interleaved_b32.json 2>&1 | tail -2 | cut -c1-1600
timeout 600 python tests/tools/attn_bench.py --rounds 5 --bwd --no-check --variants 12,11 --shapes "40,4096,4096,8;80,1024,1024,8" --out $O/attn_bwd_interleaved.json 2>&1 | tail -2 | cut -c1-1200
