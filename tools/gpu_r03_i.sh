#!/bin/bash
# round 3, visit i: LayerNorm multi-row test + A/B, then the round's profiles: steady-state kernel stats of the training step
# and of the DDIM loop, the dominant-shape probe under rocprofv3 --stats, PMC of the hybrid attention forward, TCC traffic of
# the two heaviest norm signatures
mkdir -p gpurun_out/r03_i
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_i
timeout 600 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity.py -q -p no:cacheprovider -k "layernorm" 2>&1 | grep -E "passed|failed|^E " | tail -5
B="python bench.py --no-cpu-baseline --no-vae --no-ddim --steps 20"
timeout 600 $B > $O/bench_ln_rows.log 2>&1; tail -1 $O/bench_ln_rows.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline']['norm_elementwise_family']; print('ln rows', d['value'], d['ms_per_step'], json.dumps(f['per_kernel']['layernorm_fwd']))"
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o train -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim --no-vae > $O/prof_train.log 2>&1
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/prof_summary.py $DB --steady adamw_dev_kernel 8 > $O/train_kernel_stats_steady.txt 2>&1; head -24 $O/train_kernel_stats_steady.txt | cut -c1-150
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o ddim -- python bench.py --ddim-only --ddim-warm 1 --ddim-loops 1 > $O/prof_ddim.log 2>&1
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/prof_summary.py $DB --steady ddim_step_dev_kernel 40 > $O/ddim_kernel_stats_steady.txt 2>&1; head -12 $O/ddim_kernel_stats_steady.txt | cut -c1-150
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o probe --output-format csv -- python bench.py --probe-only > $O/probe_profiled.json 2> $O/probe_profiled.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/dominant_kernel_stats.csv 2>/dev/null
head -3 $O/dominant_kernel_stats.csv | cut -c1-220; tail -1 $O/probe_profiled.json | cut -c1-400
rm -rf $O/prof
bash tools/pmc_kernel.sh attn_fwd_hyb r03_i/pmc_attn -- python tests/tools/attn_bench.py --variants 0 --no-check --shapes "40,4096,4096,8" > $O/pmc_attn_fwd_hyb.txt 2>&1; cat $O/pmc_attn_fwd_hyb.txt | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_norm_$c -o r --output-format csv -- python tools/probe_norm.py > $O/pmc_norm_$c.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$O/pmc_norm_%s/**/*counter_collection.csv"%c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==c and ("gn_" in r["Kernel_Name"] or "ln_fwd" in r["Kernel_Name"]):
                acc[r["Kernel_Name"].split("<")[0].split("(")[0][-40:]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(c, k, "dispatches", len(v), "mean", round(sum(v)/len(v),1), "(KiB; FETCH_SIZE under-reports wide coalesced reads 2x on gfx950)")
PY
rm -rf $O/prof $O/pmc_attn/p*/  $O/pmc_norm_*/ 2>/dev/null; true
