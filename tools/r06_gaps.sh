#!/bin/bash
# Idle time between kernels in the replayed training step: rocprofv3 kernel trace -> tools/prof_summary.py --gaps (union of the
# kernel intervals over all queues; the largest gaps with the kernels on either side).  $1 = extra environment, e.g.
# CTRLORA_OVERLAP_STREAMS=0; $2 = output tag.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${2:-default}
O=gpurun_out/r06_gaps; mkdir -p $O; rm -rf $O/trace_$TAG
env $1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_$TAG -o train -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-ddim --no-vae > $O/trace_$TAG.log 2>&1
DB=$(find $O/trace_$TAG -name "*results.db" | head -1)
python tools/prof_summary.py $DB --steady adamw_dev_kernel 4 --gaps > $O/train_timeline_$TAG.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/trace_$TAG.log | head -1
head -3 $O/train_timeline_$TAG.txt | cut -c1-200; grep "^timeline" $O/train_timeline_$TAG.txt | cut -c1-300
mv $DB $O/train_$TAG.db; rm -rf $O/trace_$TAG
