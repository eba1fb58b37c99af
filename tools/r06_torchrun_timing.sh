#!/bin/bash
# where does a one-rank torchrun launch of the bench spend its wall time on a fresh box? (the driver launches N > 1 this way)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_verify; mkdir -p $O
t0=$(date +%s)
python -c "import torch; print('import torch', torch.__version__)" ; t1=$(date +%s); echo "import torch: $((t1-t0)) s"
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ddim --no-vae > $O/bench_plain.log 2>&1; t2=$(date +%s); echo "plain bench: $((t2-t1)) s"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ddim --no-vae > $O/bench_torchrun2.log 2>&1; t3=$(date +%s); echo "torchrun bench: $((t3-t2)) s"
CTRLORA_BENCH_TRACE_TIMES=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ddim --no-vae --dist-dry-run > $O/bench_torchrun_dry.log 2>&1; t4=$(date +%s); echo "torchrun dry run: $((t4-t3)) s"
grep -o '"ms_per_step": [0-9.]*' $O/bench_plain.log $O/bench_torchrun2.log | head
