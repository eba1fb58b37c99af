#!/bin/bash
# Round 5, first GPU visit: (1) x-stationary streaming probe on the wide-N / short-K products, (2) PMC passes on the product's
# GEGLU projection (32768, 2560, 320) -- the store path --, (3) rocprofv3 kernel trace of the VAE encode.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_a; rm -rf $O; mkdir -p $O
timeout 300 ./build/probe_stream_gemm > $O/probe_stream_gemm.log 2>&1; tail -22 $O/probe_stream_gemm.log
timeout 100 ./build/probe_gemm --one 16 3 > $O/probe_gemm_c16_geglu.log 2>&1; tail -3 $O/probe_gemm_c16_geglu.log
for c in 16 20; do timeout 100 ./build/probe_gemm --one $c 5 > $O/probe_gemm_c${c}_wide_nostores.log 2>&1; tail -10 $O/probe_gemm_c${c}_wide_nostores.log; done
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
           "WRITE_SIZE" "FETCH_SIZE" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -o r --output-format csv -- ./build/probe_gemm --one 16 3 > $O/pmc$i.log 2>&1
done
python3 - <<PY > $O/pmc_geglu_proj_c16.txt
import csv,glob,collections
for f in sorted(glob.glob("$O/pmc*/*counter_collection.csv")):
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        if "gemm" not in r["Kernel_Name"]: continue
        a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,(n,v) in acc.items(): print(f"{f.split('/')[-2]:5s} {k:32s} per-dispatch {v/n:16.1f}  (n={n})")
PY
cat $O/pmc_geglu_proj_c16.txt
timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace_vae -o vae -- python tools/vae_bench.py --iters 3 > $O/vae_bench.log 2> $O/vae_bench.err
tail -2 $O/vae_bench.log
DB=$(ls $O/trace_vae/*/*.db $O/trace_vae/*.db 2>/dev/null | head -1)
python tools/prof_summary.py $DB > $O/vae_kernel_stats.txt 2>&1; head -30 $O/vae_kernel_stats.txt
find $O -name "*.db" -delete
