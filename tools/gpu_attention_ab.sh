#!/bin/bash
# Attention kernels A/B on one box: the attention parity tests, interleaved timing of forward + backward of the default path
# (attn_fwd40 + fold backward, pre-scaled Q) against the hybrid / tile-synchronous kernels at the production shapes and the
# spike case, then the training step.  Outputs -> gpurun_out/attention_ab/.
mkdir -p gpurun_out/attention_ab
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/attention_ab
timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -q -x -k "attention" > $O/pytest_attention.log 2>&1; tail -3 $O/pytest_attention.log
timeout 300 python tests/tools/attn_bench.py --bwd --variants 14,0p,1p --rounds 7 --shapes "40,4096,4096,8;40,4096,77,8;80,1024,1024,8;160,256,256,8;40,1024,1024,2" --spike --out $O/attn_bwd.json > $O/attn_bwd.log 2>&1
python - <<'PY'
import json
for c in json.load(open("gpurun_out/attention_ab/attn_bwd.json")):
    print(c["shape"])
    for k,v in c.items():
        if k!="shape": print("   ",k,{a:(round(b,5) if isinstance(b,float) else b) for a,b in v.items() if a in("dq_err","dk_err","dv_err","fwd_us_median","bwd_us_median","bwd_us_min","bwd_tflops")})
PY
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ddim --no-vae > $O/bench_train.log 2>&1; tail -1 $O/bench_train.log | cut -c1-300
