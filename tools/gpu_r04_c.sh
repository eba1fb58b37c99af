#!/bin/bash
# Round 4, GPU visit C: the one-wave-per-SIMD / two-query-block forward (variant 22, attention_fwd40x.hip) against the hybrid
# kernel (14) and the 8-wave pre-scaled-Q forward (0p): correctness incl. the forced second pass + interleaved timing; the
# attention / GroupNorm GPU tests; the step with / without the pre-scaled q.
mkdir -p gpurun_out/r04_c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python tests/tools/attn_bench.py --variants 14,0p,22p --rounds 7 --spike --shapes "40,4096,4096,8;40,4096,4096,32;40,1024,1024,8" \
  --out gpurun_out/r04_c/attn.json > gpurun_out/r04_c/attn.log 2>&1
python - <<'PY'
import json
try:
    for e in json.load(open("gpurun_out/r04_c/attn.json")):
        print(e["shape"])
        for k, v in e.items():
            if k != "shape":
                print(f"   {k:30s} fwd {v['fwd_us_median']:8.1f} us {v['fwd_tflops']:7.1f} TF/s  " +
                      " ".join(f"{n} {v[n]:.3e}" for n in ("o_err", "lse_err") if n in v))
except Exception as ex:
    print("attn_bench failed:", ex); print(open("gpurun_out/r04_c/attn.log").read()[-3000:])
PY
timeout 400 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity_r3.py -q -k "attention or groupnorm_one_launch or grouped" > gpurun_out/r04_c/pytest.log 2>&1; tail -8 gpurun_out/r04_c/pytest.log
for sw in 1 0; do
  CTRLORA_PRESCALE_Q=$sw timeout 300 python bench.py --steps 20 --warmup 5 --no-ddim --no-vae --no-cpu-baseline > gpurun_out/r04_c/bench_prescale$sw.log 2> gpurun_out/r04_c/bench_prescale$sw.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_c/bench_prescale$sw.log").read().strip().splitlines()[-1])
    print("prescale=$sw", d["value"], "img/s", d["ms_per_step"], "ms  loss", d["loss"], " attention family", d["roofline"].get("attention_family"))
except Exception as ex:
    print("bench prescale=$sw failed", ex); print(open("gpurun_out/r04_c/bench_prescale$sw.err").read()[-2000:])
PY
done
