#!/bin/bash
# VALU || MFMA overlap ceiling for the attention forward's instruction mix (tools/probe_interleave.hip)
mkdir -p gpurun_out/r03_p
cd $GRAFT_REPO_ROOT
[ -x tmp/probe_interleave ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o tmp/probe_interleave tools/probe_interleave.hip
timeout 60 ./tmp/probe_interleave > gpurun_out/r03_p/probe_interleave.log 2>&1; cat gpurun_out/r03_p/probe_interleave.log
