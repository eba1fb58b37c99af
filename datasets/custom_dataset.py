"""Paired-image fine-tuning dataset with the layout and item format of the reference's
`datasets.custom_dataset.CustomDataset` (datasets/custom_dataset.py:9-87):

    root/prompt.json      one JSON object per line: {"source": "source/0000.jpg", "target": "target/0000.jpg", "prompt": "..."}
    root/source/*.jpg     condition images  -> item["hint"]  float32 H x W x 3 in [0, 1]
    root/target/*.jpg     training images   -> item["jpg"]   float32 H x W x 3 in [-1, 1]
                                               item["txt"]   the prompt, or '' with probability drop_rate
                                                             (classifier-free-guidance dropout, numpy RNG)

Lines whose files are missing are skipped.  Images are decoded with Pillow as RGB (the reference decodes with
OpenCV and converts BGR -> RGB: same pixels).
"""
import json
import os

import numpy as np
from PIL import Image
from torch.utils.data import Dataset


class CustomDataset(Dataset):
    def __init__(self, root: str, drop_rate: float = 0.0):
        self.root = root
        self.drop_rate = drop_rate
        base = os.path.expanduser(root)
        for need, kind in (("prompt.json", os.path.isfile), ("source", os.path.isdir), ("target", os.path.isdir)):
            if not kind(os.path.join(base, need)):
                raise FileNotFoundError(f"{os.path.join(base, need)} not found.")
        have = {d: set(os.listdir(os.path.join(base, d))) for d in ("source", "target")}
        self.data = []
        with open(os.path.join(base, "prompt.json"), "rt") as f:
            for line in f:
                if not line.strip():
                    continue
                rec = json.loads(line)
                if all(rec[d].removeprefix(d + "/") in have[d] for d in ("source", "target")):
                    self.data.append(rec)

    def __len__(self):
        return len(self.data)

    def _rgb(self, rel):
        with Image.open(os.path.join(self.root, rel)) as im:
            return np.asarray(im.convert("RGB"), dtype=np.float32)

    def __getitem__(self, idx):
        rec = self.data[idx]
        prompt = "" if np.random.rand() < self.drop_rate else rec["prompt"]
        return dict(jpg=self._rgb(rec["target"]) / 127.5 - 1.0, txt=prompt, hint=self._rgb(rec["source"]) / 255.0)
