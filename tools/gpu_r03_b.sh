#!/bin/bash
# round 3, visit b: MFMA issue-rate probe (K = 16 legacy form), fp32 atomics probe, the DDIM-vs-oracle test + remaining r3 tests,
# bench with the per-kernel HBM breakdown
mkdir -p gpurun_out/r03_b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_b
timeout 120 build/probe_mfma > $O/probe_mfma.log 2>&1; cat $O/probe_mfma.log
timeout 120 build/probe_atomic > $O/probe_atomic.log 2>&1; cat $O/probe_atomic.log
rm -f gpurun_out/parity_measured.jsonl
timeout 1500 python -m pytest tests/test_gpu_parity_r3.py -q -k "not inference_executor and not lora_fold" 2>&1 | grep -v Warning | tail -40 > $O/pytest_r3.log; grep -E "passed|failed|Error|assert" $O/pytest_r3.log | tail -12
cp gpurun_out/parity_measured.jsonl $O/ 2>/dev/null
timeout 900 python bench.py --no-cpu-baseline --no-vae --steps 10 > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], json.dumps(d['roofline'].get('norm_elementwise_family'))[:1500]); print(json.dumps(d.get('ddim'))[:600])"
