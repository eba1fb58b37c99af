"""First-stage (AutoencoderKL.encode on the engine) alone: the leg of bench.py's `end_to_end`, for rocprofv3 runs.
usage: python tools/vae_bench.py [--iters 3] [--batch 8]   (encodes 2 * batch 512x512 images per call, as one training step does)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print(json.dumps(bench.vae_bench(dev, torch.bfloat16, a.batch, a.iters)))
