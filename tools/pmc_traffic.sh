#!/bin/bash
# HBM traffic of the dominant kernel (conv 320->320 @64x64, B=8, full-line GEMM) from TCC counters, separate passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o r --output-format csv -- ./build/probe_gemm --one 16 0 > $OUT/$c.log 2>&1
done
python3 - <<PY
import csv,glob
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows=[r for f in glob.glob("$OUT/%s/*counter_collection.csv"%c) for r in csv.DictReader(open(f)) if "gemm_fl" in r["Kernel_Name"] and r["Counter_Name"]==c]
    v=[float(r["Counter_Value"]) for r in rows]
    print(c, "dispatches", len(v), "mean", sum(v)/max(1,len(v)), "(rocprofv3 units: KiB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950)")
PY
