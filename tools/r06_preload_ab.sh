#!/bin/bash
# A/B of one compile flag over the whole library: -mllvm -amdgpu-kernarg-preload-count=16 (the command processor puts the first
# 14-16 kernel-argument dwords into SGPRs at dispatch instead of the kernel's first instructions waiting on s_load of the kernarg
# segment).  build/preload/libctrlora_hip_preload.so vs the in-tree library, same box, interleaved.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_preload; mkdir -p $O
timeout 600 env CTRLORA_LIB=$PWD/build/preload/libctrlora_hip_preload.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_gemm_w4.py -x -q > $O/pytest_preload.log 2>&1; tail -2 $O/pytest_preload.log
for i in 1 2; do for w in base preload; do
  L=""; [ $w = preload ] && L=$PWD/build/preload/libctrlora_hip_preload.so
  CTRLORA_LIB=$L timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-ddim > $O/bench_train_${w}_$i.log 2>> $O/err.log
  CTRLORA_LIB=$L timeout 400 python bench.py --ddim-only --ddim-core-only > $O/bench_ddim_${w}_$i.log 2>> $O/err.log
done; done
for f in $O/bench_train_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
for f in $O/bench_ddim_*.log; do echo $f $(grep -o '"value": [0-9.]*' $f | head -1); done
