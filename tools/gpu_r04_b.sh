#!/bin/bash
# Round 4, GPU visit B: the pre-scaled-Q attention forward (csrc/attention_fwd40.hip) against the hybrid kernel -- correctness
# (incl. the forced second pass) + interleaved timing --, the attention / GroupNorm GPU tests, and the step with the engine
# writing a pre-scaled q (A/B: CTRLORA_PRESCALE_Q=0).
mkdir -p gpurun_out/r04_b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python tests/tools/attn_bench.py --variants 14,0p,21p,1p --rounds 7 --spike --bwd --shapes "40,4096,4096,8;40,4096,4096,32" \
  --out gpurun_out/r04_b/attn.json > gpurun_out/r04_b/attn.log 2>&1
python - <<'PY'
import json
try:
    for e in json.load(open("gpurun_out/r04_b/attn.json")):
        print(e["shape"])
        for k, v in e.items():
            if k != "shape":
                print(f"   {k:30s} fwd {v['fwd_us_median']:8.1f} us {v['fwd_tflops']:7.1f} TF/s  bwd {v.get('bwd_us_median', 0):8.1f} us  " +
                      " ".join(f"{n} {v[n]:.3e}" for n in ("o_err", "lse_err", "dq_err", "dk_err", "dv_err") if n in v))
except Exception as ex:
    print("attn_bench failed:", ex); print(open("gpurun_out/r04_b/attn.log").read()[-3000:])
PY
timeout 400 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity_r3.py -q -x -k "attention or groupnorm_one_launch or grouped" > gpurun_out/r04_b/pytest.log 2>&1; tail -5 gpurun_out/r04_b/pytest.log
for sw in 1 0; do
  CTRLORA_PRESCALE_Q=$sw timeout 300 python bench.py --steps 20 --warmup 5 --no-ddim --no-vae --no-cpu-baseline > gpurun_out/r04_b/bench_prescale$sw.log 2> gpurun_out/r04_b/bench_prescale$sw.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_b/bench_prescale$sw.log").read().strip().splitlines()[-1])
    print("prescale=$sw", d["value"], "img/s", d["ms_per_step"], "ms  attention family", d["roofline"].get("attention_family"))
except Exception as ex:
    print("bench prescale=$sw failed", ex); print(open("gpurun_out/r04_b/bench_prescale$sw.err").read()[-2000:])
PY
done
