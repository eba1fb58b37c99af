#!/bin/bash
# round 3, visit d: the whole GPU suite (incl. the f4 script test and the graphed pre-training test), attention variants
# with the backward, default bench + pre-training leg
mkdir -p gpurun_out/r03_d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_d
rm -f gpurun_out/parity_measured.jsonl
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 > $O/pytest_gpu_full.log; grep -v "Warning\|warnings.warn\|^  " $O/pytest_gpu_full.log | grep -E "passed|failed|^FAILED|^ERROR|Error|assert " | tail -40
cp gpurun_out/parity_measured.jsonl $O/ 2>/dev/null
timeout 600 python tests/tools/attn_bench.py --bwd --variants 12,0,10,11 --shapes "40,4096,4096,8;80,1024,1024,8" --out $O/attn_variants_bwd.json 2>&1 | tail -3 | cut -c1-1600
timeout 300 python tests/tools/attn_bench.py --variants 12,0,10 --no-check --shapes "40,4096,4096,32;80,1024,1024,32" --out $O/attn_variants_b32.json 2>&1 | tail -3 | cut -c1-800
timeout 900 python bench.py --no-cpu-baseline --no-vae --steps 20 > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['attention_family'], d['ddim']['value'])"
timeout 900 python bench.py --pretrain-only > $O/bench_pretrain.log 2>&1; tail -1 $O/bench_pretrain.log | cut -c1-900
