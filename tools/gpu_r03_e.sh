#!/bin/bash
# round 3, visit e: the order-dependent failure of the segmented-graph test (alone, fresh process, full traceback), the
# fixed script / pre-training tests, bench with the HBM family's top signatures
mkdir -p gpurun_out/r03_e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_e
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "segmented" > $O/pytest_segmented.log 2>&1; grep -v "Warning\|warnings.warn" $O/pytest_segmented.log | grep -E "passed|failed|^E |Error" | tail -30
timeout 900 python -m pytest tests/test_gpu_scripts.py tests/test_pretrain.py -q -p no:cacheprovider > $O/pytest_scripts.log 2>&1; grep -E "passed|failed|^E |^FAILED" $O/pytest_scripts.log | tail -12
timeout 900 python bench.py --no-cpu-baseline --no-vae --no-ddim --steps 20 > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline']['norm_elementwise_family']; print(d['value'], d['ms_per_step'], f['ms_per_step'], f['frac']); print(json.dumps(f['per_kernel'])); print(json.dumps(f['top'], indent=0))"
