#!/bin/bash
# PMC passes of one visit (each counter set in its own rocprofv3 run, --kernel-trace only: tools/pmc_kernel.sh): the attention
# forward and the two backward kernels at B x H = 64, N = 4096, d_head 40 (matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024
# SIMDs against GRBM_GUI_ACTIVE / 8 XCDs), the TCC traffic of the dominant convolution kernel (tools/pmc_traffic.sh ->
# profiles/dominant_kernel_traffic.json), and the contraction census of the training step.
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc
bash tools/pmc_kernel.sh attn_fwd40 pmc/attn_fwd40 -- python tests/tools/attn_bench.py --variants 0p --rounds 1 --no-check --shapes "40,4096,4096,8" > $O/pmc_attn_fwd40.txt 2>&1
for k in attn_bwd_dkv attn_bwd_dq; do
  bash tools/pmc_kernel.sh $k pmc/$k -- python tests/tools/attn_bench.py --bwd --variants 0p --rounds 1 --no-check --shapes "40,4096,4096,8" > $O/pmc_$k.txt 2>&1
done
tail -22 $O/pmc_attn_fwd40.txt
bash tools/build_probes.sh gemm > $O/build_probes.log 2>&1
bash tools/pmc_traffic.sh > $O/pmc_traffic.txt 2>&1; tail -3 $O/pmc_traffic.txt
timeout 400 python tools/gemm_census.py > $O/gemm_census_train.txt 2> $O/gemm_census_train.err; head -5 $O/gemm_census_train.txt
find $O -name "*.db" -delete
