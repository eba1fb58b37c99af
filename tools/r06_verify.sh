#!/bin/bash
# final-tree verification: the GPU suite and smoke as the driver runs them, then the bench line the driver records
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_verify; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-400
