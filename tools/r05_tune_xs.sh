#!/bin/bash
# Offer configuration 34 (x-stationary kernel) to every signature of the training step, the DDIM step and the VAE; the rows it wins
# become ctrlora_amd/gemm_tuned_gfx950_xs.json (an overlay of the base table).  Then the bench with and without the overlay.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_d; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm_xs.py -x -q -m gpu > $O/pytest_xs.log 2>&1; tail -3 $O/pytest_xs.log
timeout 900 python tools/gemm_autotune.py --merge ctrlora_amd/gemm_tuned_gfx950.json --retry-cfgs 34 --out $O/merged.json --log $O/autotune_xs.log > $O/autotune_xs.out 2>&1
tail -45 $O/autotune_xs.out
python - <<PY
import json
t = json.load(open("$O/merged.json"))
rows = [r for r in t["entries"] if r[7] == 34]
head = {"device": t.get("device"), "columns": t["columns"], "note": "signatures won by the x-stationary kernel (cfg 34; splitk = column runs per group), layered over gemm_tuned_gfx950.json; CTRLORA_GEMM_XS=0 skips it", "predicted_saving_ms": t.get("predicted_saving_ms")}
with open("$O/gemm_tuned_gfx950_xs.json", "w") as f:
    f.write("{" + ", ".join(f"{json.dumps(k)}: {json.dumps(v)}" for k, v in head.items()) + ',\n"entries": [\n')
    f.write(",\n".join(json.dumps(r) for r in sorted(rows)))
    f.write("\n]}\n")
print("xs rows:", len(rows))
PY
cp $O/gemm_tuned_gfx950_xs.json ctrlora_amd/gemm_tuned_gfx950_xs.json
for i in 1 2; do for xs in 0 1; do
  CTRLORA_GEMM_XS=$xs timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-ddim > $O/bench_train_xs${xs}_$i.log 2>> $O/bench_err.log
  CTRLORA_GEMM_XS=$xs timeout 400 python bench.py --ddim-only --ddim-core-only > $O/bench_ddim_xs${xs}_$i.log 2>> $O/bench_err.log
done; done
for f in $O/bench_train_xs*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
for f in $O/bench_ddim_xs*.log; do echo $f $(grep -o '"value": [0-9.]*' $f | head -1); done
ls $O
