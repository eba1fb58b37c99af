#!/bin/bash
# Next round, first GPU minute: the per-wave pipelined attention forwards (cl_attention_force_variant 15-18) against the
# hybrid kernel (14): correctness vs the fp64 reference (o_err / lse_err must equal variant 14's) + interleaved timing.
# 15 / 17 were measured correct at the end of round 3 (265 / 271 us vs 244-247); 16, 18, 19, 20 have never run on a GPU.
# 19 / 20 take a pre-scaled Q (attn_bench does that) and are checked against the reference evaluated on that Q.
mkdir -p gpurun_out/r04_attn
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 120 python tests/tools/attn_bench.py --variants 14,15,16,18,19,20 --rounds 7 --shapes "40,4096,4096,8;40,1024,1024,8;40,4096,4096,32" \
  --out gpurun_out/r04_attn/fwd_candidates.json > gpurun_out/r04_attn/attn.log 2>&1
python - <<'PY'
import json
for e in json.load(open("gpurun_out/r04_attn/fwd_candidates.json")):
    print(e["shape"])
    for k, v in e.items():
        if k != "shape":
            print(f"   {k:34s} {v['fwd_us_median']:8.1f} us  {v['fwd_tflops']:7.1f} TF/s  o_err {v['o_err']:.6e}  lse_err {v['lse_err']:.3e}")
PY
