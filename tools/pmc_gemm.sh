#!/bin/bash
# PMC passes over one GEMM shape (rocprofv3 --pmc in its own runs, kernel-trace only).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG=${1:-8}; CASE=${2:-0}
OUT=gpurun_out/pmc_gemm_c${CFG}_k${CASE}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > gpurun_out/rocprof_counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o r --output-format csv -- ./build/probe_gemm --one $CFG $CASE > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        if "gemm" not in r["Kernel_Name"]: continue
        a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,(n,v) in acc.items(): print(f"{f.split('/')[-2]:4s} {k:32s} per-dispatch {v/n:16.1f}  (n={n})")
PY
