#!/bin/bash
# round 3, visit h: launch table for the NEW product signatures (grouped LoRA products, folded grouped q|k|v of DDIM) and the
# 128 x 80 full-line tiles offered to the signatures that already have an entry; A/B of the step with the old and new tables
mkdir -p gpurun_out/r03_h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_h
cp ctrlora_amd/gemm_tuned_gfx950.json $O/table_before.json
CTRLORA_GEMM_TUNED=0 timeout 900 python tools/gemm_autotune.py --merge ctrlora_amd/gemm_tuned_gfx950.json --only-new --retry-cfgs 31,32 --budget-s 480 --out $O/table_after.json --log $O/autotune.log > $O/autotune.out 2>&1; tail -45 $O/autotune.out | cut -c1-200
B="python bench.py --no-cpu-baseline --no-vae --steps 20"
timeout 600 $B > $O/bench_table_before.log 2>&1; tail -1 $O/bench_table_before.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('before', d['value'], d['ms_per_step'], d['ddim']['value'])"
CTRLORA_GEMM_TABLE=$O/table_after.json timeout 600 $B > $O/bench_table_after.log 2>&1; tail -1 $O/bench_table_after.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('after', d['value'], d['ms_per_step'], d['ddim']['value'], d['config']['gemm_launch_table_entries'])"
