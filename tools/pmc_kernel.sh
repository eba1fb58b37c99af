#!/bin/bash
# PMC passes over any command, reporting counters for kernels whose name contains $1 and whose grid is the
# largest seen (the timed shape).  usage: tools/pmc_kernel.sh <kernel-substring> <outdir> -- <command...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KSUB=$1; OUT=gpurun_out/$2; shift 3
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o r --output-format csv -- "$@" > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    rows=[r for r in csv.DictReader(open(f)) if "$KSUB" in r["Kernel_Name"]]
    if not rows: continue
    gmax=max(int(r["Grid_Size"]) for r in rows)
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in rows:
        if int(r["Grid_Size"])!=gmax: continue
        a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,(n,v) in acc.items(): print(f"{f.split('/')[-2]:4s} {k:32s} per-dispatch {v/n:16.1f}  (n={n}, grid {gmax}, vgpr {rows[0].get('VGPR_Count','?')}, lds {rows[0].get('LDS_Block_Size','?')})")
PY
