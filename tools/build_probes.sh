#!/bin/bash
# Build the standalone GPU probes (not part of the product) into build/.
set -e
cd "$(dirname "$0")/.."
mkdir -p build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17"
for p in "$@"; do
  case $p in
    gemm) hipcc $FLAGS tools/probe_gemm.hip ctrlora_amd/csrc/gemm.hip -o build/probe_gemm ;;
    *) echo "unknown probe $p"; exit 1 ;;
  esac
done
