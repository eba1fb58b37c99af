"""MultiGen-20M reader with the item format of the reference's `datasets.multigen20m.MultiGen20M`
(datasets/multigen20m.py:20-142): one JSON object per line of `path_json`, with keys
`control_<task>` (condition image under <path_meta>/conditions/), `source` (training image under
<path_meta>/images/, a leading './' is stripped) and `prompt`.

Item: {jpg: float32 512x512x3 in [-1,1], txt: prompt ('' with probability drop_rate, python `random`),
hint: float32 512x512x3 in [0,1], task: 'control_<task>'}.  Both images get the SAME square crop (random along the
long side, or centred), expressed as fractions of the condition image's size and re-applied to the training image,
then are resized to 512 (Lanczos when enlarging, area averaging when shrinking).  Unreadable samples are skipped by
walking forward through the list.  Decoding / resampling use Pillow (OpenCV is not in this image): LANCZOS for
INTER_LANCZOS4, BOX for INTER_AREA -- same geometry, resampling kernels differ in the last bits.
"""
import json
import os
import random

import numpy as np
from PIL import Image
from torch.utils.data import Dataset

_TASK_KEYS = {t: "control_" + t for t in ("hed", "canny", "seg", "depth", "normal", "openpose", "hedsketch", "bbox",
                                           "outpainting", "inpainting", "blur", "grayscale")}
_TASK_KEYS["segbase"] = "control_seg"


class MultiGen20M(Dataset):
    def __init__(self, path_json, path_meta, task, drop_rate=0.3, random_cropping=True):
        with open(path_json, "rt") as f:
            self.data = [json.loads(line) for line in f if line.strip()]
        self.path_meta = path_meta
        if task not in _TASK_KEYS:
            raise ValueError(f"unknown MultiGen-20M task '{task}'")
        self.key_prompt = _TASK_KEYS[task]
        self.resolution = 512
        self.none_loop = 0
        self.drop_rate = drop_rate
        self.random_cropping = random_cropping

    def __len__(self):
        return len(self.data)

    @staticmethod
    def _read(path):
        try:
            with Image.open(path) as im:
                return np.asarray(im.convert("RGB"))
        except (OSError, ValueError):
            return None

    @staticmethod
    def _resize(img, resolution, k):
        res = Image.LANCZOS if k > 1 else Image.BOX
        return np.asarray(Image.fromarray(img).resize((resolution, resolution), res))

    def resize_image_control(self, control_image, resolution):
        H, W, _ = control_image.shape
        if W >= H:
            crop = H
            crop_l = random.randint(0, W - crop) if self.random_cropping else (W - crop) // 2
            crop_t, crop_b, crop_r = 0, H, crop_l + crop
        else:
            crop = W
            crop_t = random.randint(0, H - crop) if self.random_cropping else (H - crop) // 2
            crop_l, crop_r, crop_b = 0, W, crop_t + crop
        img = self._resize(control_image[crop_t:crop_b, crop_l:crop_r], resolution, float(resolution) / min(H, W))
        return img, [crop_t / float(H), crop_b / float(H), crop_l / float(W), crop_r / float(W)]

    def resize_image_target(self, target_image, resolution, sizes):
        H, W, _ = target_image.shape
        t, b, l, r = int(sizes[0] * H), int(sizes[1] * H), int(sizes[2] * W), int(sizes[3] * W)
        return self._resize(target_image[t:b, l:r], resolution, float(resolution) / min(H, W))

    def _load(self, idx):
        item = self.data[idx]
        src = self._read(os.path.join(self.path_meta, "conditions", item[self.key_prompt]))
        tf = item["source"]
        if tf[0:2] == "./":
            tf = tf[2:]
        return src, self._read(os.path.join(self.path_meta, "images", tf)), item.get("prompt")

    def __getitem__(self, idx):
        source_img, target_img, prompt = self._load(idx)
        while source_img is None or target_img is None or prompt is None:      # corner cases: walk forward
            idx = idx + 1 if 0 <= idx < len(self.data) - 1 else 0
            source_img, target_img, prompt = self._load(idx)
            self.none_loop += 1
            if self.none_loop > 10000:
                break
        source_img, sizes = self.resize_image_control(source_img, self.resolution)
        target_img = self.resize_image_target(target_img, self.resolution, sizes)
        source_img = source_img.astype(np.float32) / 255.0
        target_img = target_img.astype(np.float32) / 127.5 - 1.0
        prompt = prompt if random.uniform(0, 1) > self.drop_rate else ""
        return dict(jpg=target_img, txt=prompt, hint=source_img, task=self.key_prompt)
