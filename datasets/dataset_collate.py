"""`collate_fn` of the multi-task pre-training DataLoader (datasets/dataset_collate.py:44-97 in the reference): drop
samples whose images failed to load (jpg / hint None), return None for an empty batch, otherwise collate field by field
-- numpy arrays and tensors are stacked, numbers become tensors, strings stay lists, mappings / sequences recurse."""
import collections.abc as abc

import numpy as np
import torch


def collate_fn(batch):
    if isinstance(batch, list) and batch and isinstance(batch[0], dict):
        batch = [d for d in batch if d["jpg"] is not None and d["hint"] is not None]
    if batch == []:
        return None
    e = batch[0]
    if isinstance(e, torch.Tensor):
        return torch.stack(batch, 0)
    if isinstance(e, np.ndarray):
        if e.dtype.kind in "SaUO":
            raise TypeError(f"batch must contain tensors, numbers, dicts or lists; found {e.dtype}")
        return collate_fn([torch.from_numpy(b) for b in batch])
    if isinstance(e, np.generic):
        return torch.tensor(np.asarray(batch))
    if isinstance(e, float):
        return torch.tensor(batch, dtype=torch.float64)
    if isinstance(e, int):
        return torch.tensor(batch)
    if isinstance(e, (str, bytes)):
        return batch
    if isinstance(e, abc.Mapping):
        return {k: collate_fn([d[k] for d in batch]) for k in e}
    if isinstance(e, tuple) and hasattr(e, "_fields"):
        return type(e)(*(collate_fn(s) for s in zip(*batch)))
    if isinstance(e, abc.Sequence):
        return [collate_fn(s) for s in zip(*batch)]
    raise TypeError(f"batch must contain tensors, numbers, dicts or lists; found {type(e)}")
