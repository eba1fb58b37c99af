#!/bin/bash
# Offer the round-6 configurations -- 40 / 41 / 47 / 48 (loader / consumer kernel, gemm_w4.hip; 47 / 48 persistent) and 42 .. 46
# (generic kernel, 8-slot ring) --
# to every signature of the training step, the DDIM step and the VAE, against the table as it stands (base + x-stationary overlay);
# the rows they win become ctrlora_amd/gemm_tuned_gfx950_r06.json (a second overlay).  Then the bench with and without it.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_tune; rm -rf $O; mkdir -p $O
python - <<PY
import json
b = json.load(open("ctrlora_amd/gemm_tuned_gfx950.json")); x = json.load(open("ctrlora_amd/gemm_tuned_gfx950_xs.json"))
rows = {tuple(r[:7]): r for r in b["entries"]}
rows.update({tuple(r[:7]): r for r in x["entries"]})
b["entries"] = sorted(rows.values())
json.dump(b, open("$O/current.json", "w"))
print("current table:", len(b["entries"]), "rows")
PY
CTRLORA_GEMM_R06=0 timeout 2400 python tools/gemm_autotune.py --merge $O/current.json --retry-cfgs 40,41,42,43,44,45,46,47,48 --out $O/merged.json --log $O/autotune_r06.log > $O/autotune_r06.out 2>&1
tail -12 $O/autotune_r06.out
python - <<PY
import json
t = json.load(open("$O/merged.json"))
rows = [r for r in t["entries"] if r[7] in (40, 41, 42, 43, 44, 45, 46, 47, 48)]
head = {"device": t.get("device"), "columns": t["columns"], "note": "signatures won by the round-6 configurations (40 / 41 / 47 / 48: loader / consumer tile kernel, csrc/gemm_w4.hip, 47 / 48 its persistent form; 42 .. 46: generic kernel with an 8-slot ring), layered over gemm_tuned_gfx950.json and the x-stationary overlay; CTRLORA_GEMM_R06=0 skips it", "predicted_saving_ms": t.get("predicted_saving_ms")}
with open("$O/gemm_tuned_gfx950_r06.json", "w") as f:
    f.write("{" + ", ".join(f"{json.dumps(k)}: {json.dumps(v)}" for k, v in head.items()) + ',\n"entries": [\n')
    f.write(",\n".join(json.dumps(r) for r in sorted(rows)))
    f.write("\n]}\n")
print("r06 rows:", len(rows))
for r in sorted(rows): print(r)
PY
cp $O/gemm_tuned_gfx950_r06.json ctrlora_amd/gemm_tuned_gfx950_r06.json
for i in 1 2; do for w in 0 1; do
  CTRLORA_GEMM_R06=$w timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-ddim > $O/bench_train_r06_${w}_$i.log 2>> $O/bench_err.log
  CTRLORA_GEMM_R06=$w timeout 400 python bench.py --ddim-only --ddim-core-only > $O/bench_ddim_r06_${w}_$i.log 2>> $O/bench_err.log
done; done
for f in $O/bench_train_r06_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
for f in $O/bench_ddim_r06_*.log; do echo $f $(grep -o '"value": [0-9.]*' $f | head -1); done
ls $O
