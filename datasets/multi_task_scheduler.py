"""Single-task-per-batch index scheduler for multi-task pre-training.

Drop-in for the reference's `datasets.multi_task_scheduler.BatchSchedulerSampler`
(datasets/multi_task_scheduler.py:18-80): same constructor, same `len()`, and -- given the same torch / numpy
RNG state -- the same index stream (pinned by tests/golden/next_rows.pt, which was produced by the reference
class).  An "epoch" is ceil(largest / (batch * tasks)) rounds; every round visits the tasks in a fresh
`np.random.permutation` (identity when not shuffling) and takes `batch_size` consecutive indices from that
task's own sampler, restarting an exhausted task sampler -- so small tasks are re-sampled until the largest
has been covered, and every batch a DataLoader cuts from the stream is single-task.

Under DDP the reference builds one `DistributedSampler` per task (rank/world from torch.distributed) and leaves
numpy's RNG unseeded per rank: ranks may train DIFFERENT tasks in the same step.  That behaviour is kept; the
gradient exchange for it is `ctrlora_amd.parallel.BankedGradAllReduce`.
"""
import math

import numpy as np
from torch.utils.data import DistributedSampler, RandomSampler, Sampler, SequentialSampler


class BatchSchedulerSampler(Sampler):
    def __init__(self, dataset, batch_size, distributed: bool = True, shuffle: bool = True):
        # `dataset` is a torch ConcatDataset of the per-task datasets
        self.dataset = dataset
        self.batch_size = batch_size
        self.distributed = distributed
        self.shuffle = shuffle
        self.number_of_datasets = len(dataset.datasets)
        self.largest_dataset_size = max(len(d) for d in dataset.datasets)

    def __len__(self):
        rounds = math.ceil(self.largest_dataset_size / self.batch_size)
        return self.batch_size * rounds * self.number_of_datasets

    def _task_sampler(self, task_dataset):
        if self.distributed:
            return DistributedSampler(task_dataset, shuffle=self.shuffle)
        return RandomSampler(task_dataset) if self.shuffle else SequentialSampler(task_dataset)

    def __iter__(self):
        n = self.number_of_datasets
        samplers = [self._task_sampler(d) for d in self.dataset.datasets]
        streams = [iter(s) for s in samplers]
        offsets = [0] + list(self.dataset.cumulative_sizes[:-1])   # position of each task inside the concat

        def take(task):
            try:
                return next(streams[task])
            except StopIteration:       # task exhausted before the largest one: start it over
                streams[task] = iter(samplers[task])
                return next(streams[task])

        out = []
        total = self.largest_dataset_size * n
        for _ in range(0, total, self.batch_size * n):
            order = np.random.permutation(n) if self.shuffle else np.arange(n)
            for task in order:
                out.extend(take(task) + offsets[task] for _ in range(self.batch_size))
        return iter(out)
