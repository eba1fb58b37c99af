#!/bin/bash
# two-launch GroupNorm: kernel + whole-model parity, A/B of the step; HBM traffic of the production conv template (PMC)
mkdir -p gpurun_out/r02_gn
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_gn
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_vae.py -q -m gpu -x 2>&1 | grep -v Warning | tail -5 > $O/pytest.log; tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-ddim --steps 20"
CTRLORA_GN_THREE_PASS=1 timeout 600 $B > $O/bench_gn3.log 2>&1; tail -1 $O/bench_gn3.log | cut -c1-200
timeout 600 $B > $O/bench_gn2.log 2>&1; tail -1 $O/bench_gn2.log | cut -c1-200
python - <<'PY'
import json
for f in ('gn3','gn2'):
    d=json.loads(open(f'gpurun_out/r02_gn/bench_{f}.log').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['end_to_end']['vae_encode'])
PY
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log
