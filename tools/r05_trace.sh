#!/bin/bash
# steady-state kernel traces of the training step and the DDIM step with launch tags -> per-kernel and per-SHAPE tables
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05_g}; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_gemm_xs.py "tests/test_gpu_parity_r3.py::test_inference_executor_sd15_latent64_eps_vs_oracle" -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_train -o train -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-ddim --no-vae --tag-gemm $O/tags_train.json > $O/trace_train.log 2>&1
DB=$(ls $O/trace_train/*/*.db $O/trace_train/*.db 2>/dev/null | head -1)
python tools/prof_summary.py $DB --steady adamw_dev_kernel 4 > $O/train_kernel_stats_steady.txt 2>&1; head -42 $O/train_kernel_stats_steady.txt
python tools/prof_shapes.py $DB $O/tags_train.json --steady adamw_dev_kernel 4 --top 45 > $O/train_shapes_in_step.txt 2>&1; head -50 $O/train_shapes_in_step.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_ddim -o ddim -- python bench.py --ddim-only --ddim-core-only --ddim-loops 1 --ddim-warm 2 --tag-gemm $O/tags_ddim.json > $O/trace_ddim.log 2>&1
DB=$(ls $O/trace_ddim/*/*.db $O/trace_ddim/*.db 2>/dev/null | head -1)
python tools/prof_summary.py $DB --steady ddim_step_dev_kernel 40 > $O/ddim_kernel_stats_steady.txt 2>&1; head -40 $O/ddim_kernel_stats_steady.txt
python tools/prof_shapes.py $DB $O/tags_ddim.json --steady ddim_step_dev_kernel 40 --top 40 > $O/ddim_shapes_in_step.txt 2>&1; head -45 $O/ddim_shapes_in_step.txt
find $O -name "*.db" -delete
