#!/bin/bash
# round 3, visit c: grouped LoRA products + one-launch GroupNorm (tests, then A/B in the bench), forward-attention variants
mkdir -p gpurun_out/r03_c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_c
rm -f gpurun_out/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_parity_r3.py -q -k "grouped or one_launch" 2>&1 | grep -v Warning | tail -30 > $O/pytest_new.log; grep -E "passed|failed|Error|assert|FAILED" $O/pytest_new.log | tail -15
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_pretrain.py -q -x 2>&1 | grep -v Warning | tail -30 > $O/pytest_model.log; grep -E "passed|failed|Error|assert|FAILED" $O/pytest_model.log | tail -15
cp gpurun_out/parity_measured.jsonl $O/ 2>/dev/null
B="python bench.py --no-cpu-baseline --no-vae --no-ddim --steps 20"
CTRLORA_GROUP_LORA=0 CTRLORA_GN_ONE_PASS=0 timeout 600 $B > $O/bench_off.log 2>&1; tail -1 $O/bench_off.log | cut -c1-160
CTRLORA_GROUP_LORA=1 CTRLORA_GN_ONE_PASS=0 timeout 600 $B > $O/bench_group.log 2>&1; tail -1 $O/bench_group.log | cut -c1-160
CTRLORA_GROUP_LORA=0 CTRLORA_GN_ONE_PASS=1 timeout 600 $B > $O/bench_gn1.log 2>&1; tail -1 $O/bench_gn1.log | cut -c1-160
timeout 600 $B > $O/bench_both.log 2>&1; tail -1 $O/bench_both.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'].get('norm_elementwise_family',{}).get('per_kernel')))"
timeout 600 python tests/tools/attn_bench.py --variants 0,6,7,8,9 --shapes "40,4096,4096,8;80,1024,1024,8;40,4096,4096,32" --out $O/attn_variants.json 2>&1 | tail -4 | cut -c1-1200
