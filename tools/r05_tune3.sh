#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_l; rm -rf $O; mkdir -p $O
timeout 900 python tools/gemm_autotune.py --quick --merge ctrlora_amd/gemm_tuned_gfx950.json --retry-cfgs 35,36 --out $O/merged_conv6480.json --log $O/autotune_conv6480.log > $O/autotune_conv6480.out 2>&1
head -24 $O/autotune_conv6480.out | cut -c1-160
