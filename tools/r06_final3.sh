#!/bin/bash
# Round-6 closing run ($1 = the commit the tree was at): the full GPU suite as the driver runs it, smoke, the stock comparator,
# tagged steady-state traces of the training and the DDIM step (per-kernel + per-shape tables), the default bench line, the
# dominant-kernel probe under rocprofv3 --stats, its HBM traffic from PMC passes, pre-training at batch 8 and 4.
# Outputs -> gpurun_out/r06_final3 (copied to profiles/r06_final3 by hand afterwards); every static artefact the bench quotes gets
# the commit and the sha256 of csrc/gemm.hip it was measured on (bench.py marks it `stale` when the source changes).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
COMMIT=${1:-unknown}
O=gpurun_out/r06_final3; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_measured.jsonl
SHA=$(python -c "import hashlib;print(hashlib.sha256(open('ctrlora_amd/csrc/gemm.hip','rb').read()).hexdigest()[:12])")
echo "{\"measured_at_commit\": \"$COMMIT\", \"source_sha256\": \"$SHA\", \"source\": \"ctrlora_amd/csrc/gemm.hip\"}" > $O/provenance.json
timeout 1800 python -m pytest tests/ -x -q -m gpu --durations=20 > $O/pytest_gpu.log 2>&1; tail -30 $O/pytest_gpu.log | grep "passed\|failed\|s call" | head -24
cp gpurun_out/parity_measured.jsonl $O/parity_measured.jsonl 2>/dev/null
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -4 $O/smoke.log | cut -c1-200
timeout 600 python tests/tools/compare_stock.py --out $O/r06_compare_precision.json > $O/compare_stock.log 2>&1; tail -3 $O/compare_stock.log | cut -c1-300
python - <<PY
import json
try:
    d = json.load(open("$O/r06_compare_precision.json")); d["measured_at_commit"] = "$COMMIT"; d["source_sha256"] = "$SHA"
    json.dump(d, open("$O/r06_compare_precision.json", "w"), indent=1)
except Exception as e:
    print("compare_stock json:", e)
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_train -o train -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-ddim --no-vae --tag-gemm $O/tags_train.json > $O/trace_train.log 2>&1
DB=$(find $O/trace_train -name "*results.db" | head -1)
python tools/prof_summary.py $DB --steady adamw_dev_kernel 4 > $O/train_kernel_stats_steady.txt 2>&1; head -14 $O/train_kernel_stats_steady.txt | cut -c1-170
python tools/prof_shapes.py $DB $O/tags_train.json --steady adamw_dev_kernel 4 --top 60 > $O/train_shapes_in_step.txt 2>&1; head -12 $O/train_shapes_in_step.txt | cut -c1-170
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_ddim -o ddim -- python bench.py --ddim-only --ddim-core-only --ddim-loops 1 --ddim-warm 2 --tag-gemm $O/tags_ddim.json > $O/trace_ddim.log 2>&1
DB=$(find $O/trace_ddim -name "*results.db" | head -1)
python tools/prof_summary.py $DB --steady ddim_step_dev_kernel 40 > $O/ddim_kernel_stats_steady.txt 2>&1; head -10 $O/ddim_kernel_stats_steady.txt | cut -c1-170
python tools/prof_shapes.py $DB $O/tags_ddim.json --steady ddim_step_dev_kernel 40 --top 50 > $O/ddim_shapes_in_step.txt 2>&1
# HBM traffic of the dominant kernel (separate PMC passes), then the artefacts the bench reads -- BEFORE the bench line
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o r --output-format csv -- ./build/probe_gemm --one 16 0 > $O/pmc_$c.log 2>&1
done
python - <<PY
import csv, glob, json
v = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [r for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True) for r in csv.DictReader(open(f))
            if "gemm_fl" in r["Kernel_Name"] and r["Counter_Name"] == c]
    x = [float(r["Counter_Value"]) for r in rows]
    v[c] = (sum(x) / max(1, len(x)), len(x))
    print(c, "dispatches", len(x), "mean KiB", v[c][0])
if v["FETCH_SIZE"][1] and v["WRITE_SIZE"][1]:
    d = {"kernel": "gemm_fl_kernel<bf16,256,160,4x2 waves,conv-s1,3-slot ring,ping-pong schedule (the production template <...,1,3,3>, cfg 16)> conv3x3 320->320 @64x64 B8",
         "source": "tools/r06_final3.sh on MI355X, round 6 (rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, %d / %d dispatches; ./build/probe_gemm --one 16 0)" % (v["FETCH_SIZE"][1], v["WRITE_SIZE"][1]),
         "FETCH_SIZE_KiB_raw": round(v["FETCH_SIZE"][0], 2), "WRITE_SIZE_KiB_raw": round(v["WRITE_SIZE"][0], 2),
         "correction": "FETCH_SIZE doubled (gfx950 counts 128-byte requests at 64 bytes for wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as counted",
         "hbm_bytes_per_launch": int(round((2 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024)), "algorithmic_bytes_per_launch": 43800000,
         "measured_at_commit": "$COMMIT", "source_sha256": "$SHA"}
    json.dump(d, open("$O/dominant_kernel_traffic.json", "w"), indent=1)
    json.dump(d, open("profiles/dominant_kernel_traffic.json", "w"), indent=1)
PY
mkdir -p profiles/r06_final3
cp $O/train_shapes_in_step.txt $O/provenance.json profiles/r06_final3/
cp $O/r06_compare_precision.json profiles/r06_compare_precision.json 2>/dev/null
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-700
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_probe -o probe --output-format csv -- python bench.py --probe-only > $O/probe_profiled.json 2> $O/probe_profiled.err
find $O/prof_probe -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/dominant_kernel_stats.csv
head -2 $O/dominant_kernel_stats.csv | cut -c1-200
timeout 400 python bench.py --pretrain-only > $O/bench_pretrain.log 2>&1; tail -1 $O/bench_pretrain.log | cut -c1-300
timeout 400 python bench.py --pretrain-only --batch 4 > $O/bench_pretrain_b4.log 2>&1; tail -1 $O/bench_pretrain_b4.log | cut -c1-300
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ddim > $O/bench_with_vae.log 2>&1; tail -1 $O/bench_with_vae.log | cut -c1-300
find $O -name "*.db" -delete; rm -rf $O/trace_train $O/trace_ddim $O/prof_probe $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE; du -sh $O
