#!/bin/bash
# Round 4, GPU visit A: (1) the attention forward candidates that never ran (16 / 18 / 19 / 20) against the hybrid kernel,
# interleaved A/B; (2) the GroupNorm small-pixel-count fix; (3) this round's "before" bench line, incl. the new
# DDIM legs (cold call, image hint hoisted / reference-faithful).
mkdir -p gpurun_out/r04_a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 150 python tests/tools/attn_bench.py --variants 14,15,16,18,19,20 --rounds 7 --shapes "40,4096,4096,8;40,1024,1024,8" \
  --out gpurun_out/r04_a/fwd_candidates.json > gpurun_out/r04_a/attn.log 2>&1
python - <<'PY'
import json
try:
    for e in json.load(open("gpurun_out/r04_a/fwd_candidates.json")):
        print(e["shape"])
        for k, v in e.items():
            if k != "shape":
                print(f"   {k:34s} {v['fwd_us_median']:8.1f} us  {v['fwd_tflops']:7.1f} TF/s  o_err {v['o_err']:.6e}  lse_err {v['lse_err']:.3e}")
except Exception as ex:
    print("attn_bench failed:", ex); print(open("gpurun_out/r04_a/attn.log").read()[-3000:])
PY
timeout 200 python -m pytest tests/test_gpu_parity_r3.py -q -x -k "groupnorm_one_launch" > gpurun_out/r04_a/pytest_gn.log 2>&1; tail -3 gpurun_out/r04_a/pytest_gn.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_a/bench.log 2> gpurun_out/r04_a/bench.err; tail -c 6000 gpurun_out/r04_a/bench.log; tail -5 gpurun_out/r04_a/bench.err
