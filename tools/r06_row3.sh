#!/bin/bash
# three-taps-per-problem conv weight gradient (pre-training): kernel parity, the pre-training suites, same-box A/B of the pre-training bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_row3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_r3.py -x -q -k "three_taps or conv_tap" > $O/pytest_kernel.log 2>&1; grep "passed\|failed" $O/pytest_kernel.log | tail -1
grep -n "^E  \|Error" $O/pytest_kernel.log | head -10
timeout 1500 python -m pytest tests/test_gpu_parity_r3.py tests/test_pretrain.py tests/test_gpu_scripts.py -x -q -m gpu -k "pretrain" > $O/pytest_pretrain.log 2>&1; grep "passed\|failed" $O/pytest_pretrain.log | tail -1
for i in 1 2; do for w in 0 1; do
  CTRLORA_WGRAD_ROW3=$w timeout 400 python bench.py --pretrain-only > $O/bench_pretrain_row3_${w}_$i.log 2>> $O/err.log
done; done
for f in $O/bench_pretrain_row3_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"eager_ms_per_step": [0-9.]*' $f | head -1); done
