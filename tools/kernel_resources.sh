#!/bin/bash
# Per-kernel VGPR / scratch / occupancy summary of one HIP source (compile-only, no GPU needed).
# usage: tools/kernel_resources.sh ctrlora_amd/csrc/gemm.hip
src=$(realpath "$1")
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -c "$src" -o /tmp/_kr.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m:
        cur={"name":subprocess.run(["c++filt",m.group(1)],capture_output=True,text=True).stdout.strip()[:90]}; rows.append(cur); continue
    m=re.search(r"remark:\s+([A-Za-z][\w /\[\]]*?): (\d+)",line)
    if m and cur is not None: cur[m.group(1).strip()]=m.group(2)
for r in rows:
    print("%-92s vgpr %-4s agpr %-4s scratch %-4s spill %-3s occ %s" % (r["name"], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("VGPRs Spill"), r.get("Occupancy [waves/SIMD]")))
'
