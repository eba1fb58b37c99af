#!/bin/bash
# The data-parallel code path on ONE GPU: segment graphs with bucketed all-reduces in between (--force-split-graphs), plain and
# under torch.distributed.run with one rank (RCCL initialised, the all-reduces go through it).
mkdir -p gpurun_out/dp_check
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/dp_check
timeout 280 python bench.py --gpus 1 --force-split-graphs --steps 8 --warmup 3 --no-ddim --no-vae --no-cpu-baseline > $O/split.log 2> $O/split.err; tail -1 $O/split.log | cut -c1-700; tail -3 $O/split.err | cut -c1-300
timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --force-split-graphs --steps 8 --warmup 3 --no-ddim --no-vae --no-cpu-baseline > $O/torchrun.log 2> $O/torchrun.err; tail -1 $O/torchrun.log | cut -c1-400; tail -3 $O/torchrun.err | cut -c1-300
