#!/bin/bash
# What bounds the d_head-40 fold backward kernels?  The product library against five probe builds with one
# ingredient removed each (tools/build_probes.sh attn_bwd_abl), same shape, separate processes (absolute times), plus per-kernel
# durations from rocprofv3 for the product and the no-MFMA / no-LDS builds.
mkdir -p gpurun_out/attention_bwd_ablations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/attention_bwd_ablations
for a in 0 1 2 3 4 5 0; do
  if [ $a = 0 ]; then L=""; else L="--lib build/abl/libctrlora_hip_abl$a.so"; fi
  timeout 120 python tests/tools/attn_bench.py --bwd --variants 0p --rounds 5 --no-check --shapes "40,4096,4096,8" $L > $O/abl$a.log 2>&1
  echo "abl $a: $(grep -o '"bwd_us_median": [0-9.]*' $O/abl$a.log | head -1) $(grep -o '"fwd_us_median": [0-9.]*' $O/abl$a.log | head -1)"
done
for a in 0 2 3; do
  if [ $a = 0 ]; then L=""; else L="--lib build/abl/libctrlora_hip_abl$a.so"; fi
  rm -rf $O/tr$a
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr$a -o t --output-format csv -- python tests/tools/attn_bench.py --bwd --variants 0p --rounds 3 --no-check --shapes "40,4096,4096,8" $L > /dev/null 2>&1
  echo "abl $a kernels:"; f=$(find $O/tr$a -name "*kernel_stats.csv" | head -1); grep "attn_bwd" $f | awk -F, '{print "   ", $1, $2, $4}' | cut -c1-160
  rm -rf $O/tr$a
done
