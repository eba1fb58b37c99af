#!/bin/bash
# Build the standalone GPU probes (not part of the product) into build/.
set -e
cd "$(dirname "$0")/.."
mkdir -p build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value"
C=ctrlora_amd/csrc
for p in "$@"; do
  case $p in
    gemm) hipcc $FLAGS tools/probe_gemm.hip $C/gemm.hip $C/gemm_xs.hip $C/gemm_w4.hip $C/wgrad.hip -o build/probe_gemm ;;
    gemm_w4) hipcc $FLAGS -DW4_PROBE -DFL_TIMING tools/probe_gemm.hip $C/gemm.hip $C/gemm_xs.hip $C/gemm_w4.hip $C/wgrad.hip -o build/probe_gemm_w4 ;;
    gemm_t) hipcc $FLAGS -DFL_TIMING tools/probe_gemm.hip $C/gemm.hip $C/gemm_xs.hip $C/gemm_w4.hip $C/wgrad.hip -o build/probe_gemm_t ;;
    attn) hipcc $FLAGS tools/probe_attn.hip $C/gemm.hip $C/gemm_xs.hip $C/gemm_w4.hip $C/attention_fwd.hip $C/attention_bwd.hip $C/attention_tr.hip $C/elementwise.hip -o build/probe_attn ;;
    attn_bwd) hipcc $FLAGS tools/probe_attn_bwd.hip $C/gemm.hip $C/gemm_xs.hip $C/gemm_w4.hip $C/attention_fwd.hip $C/attention_bwd.hip $C/attention_tr.hip $C/elementwise.hip -o build/probe_attn_bwd ;;
    stream_gemm) hipcc $FLAGS tools/probe_stream_gemm.hip -o build/probe_stream_gemm ;;
    attn_bwd_abl)   # the product library with ONE ingredient of the d_head-40 fold backward kernels removed (attention_tr.hip: ATTN_BWD_ABL)
      python -m ctrlora_amd.build > /dev/null
      mkdir -p build/abl
      for a in 1 2 3 4 5; do
        hipcc $FLAGS -fPIC -DATTN_BWD_ABL=$a -c $C/attention_tr.hip -o build/abl/attention_tr_abl$a.o
        hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/obj/*.o | grep -v attention_tr.o) build/abl/attention_tr_abl$a.o -o build/abl/libctrlora_hip_abl$a.so
      done ;;
    preload)        # the product library with the command processor preloading kernel arguments into SGPRs (A/B: tools/r06_preload_ab.sh)
      mkdir -p build/preload
      for f in gemm gemm_xs gemm_w4 wgrad norm norm_coop elementwise attention_fwd attention_bwd attention_tr attention_fwd40 capi; do
        hipcc $FLAGS -fPIC -mllvm -amdgpu-kernarg-preload-count=16 -c $C/$f.hip -o build/preload/$f.o &
      done; wait
      hipcc --offload-arch=gfx950 -shared -fPIC build/preload/*.o -o build/preload/libctrlora_hip_preload.so ;;
    *) echo "unknown probe $p"; exit 1 ;;
  esac
done
