#!/bin/bash
# PMC passes (each counter set its own rocprofv3 run, kernel-trace only) on the x-stationary kernel at (32768, 2560, 320)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_o; rm -rf $O; mkdir -p $O
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "WRITE_SIZE" "FETCH_SIZE" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -o r --output-format csv -- ./build/probe_gemm --one 34 3 > $O/pmc$i.log 2>&1
done
python3 - <<PY > $O/pmc_xs_32768x2560x320.txt
import csv,glob,collections
for f in sorted(glob.glob("$O/pmc*/*counter_collection.csv")):
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        if "gemm_xs" not in r["Kernel_Name"]: continue
        a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,(n,v) in acc.items(): print(f"{f.split('/')[-2]:5s} {k:32s} per-dispatch {v/n:16.1f}  (n={n})")
PY
cat $O/pmc_xs_32768x2560x320.txt; tail -2 $O/pmc1.log
