#!/bin/bash
# delta fused into the dQ kernel: attention parity at production shapes, A/B of the step; then the launch table
# re-measured with the persistent forms on offer and the merged-LoRA DDIM shapes
mkdir -p gpurun_out/r02_delta
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_delta
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py -q -m gpu -x -k "attention or attn or train or step or grad or sd15 or finetune" 2>&1 | grep -v Warning | tail -5 > $O/pytest_attn.log; tail -3 $O/pytest_attn.log
B="python bench.py --no-cpu-baseline --no-vae --no-ddim --steps 20"
CTRLORA_ATTN_FUSE_DELTA=0 timeout 600 $B > $O/bench_sep_delta.log 2>&1; tail -1 $O/bench_sep_delta.log | cut -c1-200
timeout 600 $B > $O/bench_fused_delta.log 2>&1; tail -1 $O/bench_fused_delta.log | cut -c1-200
CTRLORA_GEMM_TUNED=0 timeout 900 python tools/gemm_autotune.py --out $O/gemm_tuned_gfx950.json --log $O/autotune.log > $O/autotune.out 2>&1
tail -22 $O/autotune.out | cut -c1-200
[ -f $O/gemm_tuned_gfx950.json ] || exit 1
cp $O/gemm_tuned_gfx950.json ctrlora_amd/gemm_tuned_gfx950.json
cp gpurun_out/gemm_census_train_ingraph.txt $O/ 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --no-vae > $O/bench_tuned_v3.log 2>&1; tail -1 $O/bench_tuned_v3.log | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_delta/bench_tuned_v3.log').read().strip().splitlines()[-1])
print('train', d['value'], d['ms_per_step'], 'ddim', d['ddim']['value'], d['ddim']['ms_per_step'])
PY
