"""Is the graphed training step a pure function of its inputs?  Builds BASELINE configs[1] (rank-128 YAML, B = 8, latent 64x64,
bf16, lr = 0 so the weights stay put), replays the step N times on the same inputs and prints, per replay, the loss bits and
checksums of eps-side and gradient-side state.  `--variant V` selects a cl_debug_attention_variant code first; environment
switches (CTRLORA_PRESCALE_Q, CTRLORA_HOIST_EMB_BWD, ...) are read by the engine as usual.  Debug aid, GPU only.
    python tests/tools/debug_determinism.py [--replays 6] [--variant 14] [--one-stream]
"""
import argparse
import os
import struct
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replays", type=int, default=6)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--one-stream", action="store_true")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    import bench
    from ctrlora_amd import hip
    from ctrlora_amd.train import GraphedTrainStep
    assert hip.lib().cl_debug_attention_variant(a.variant) == 0
    model = bench.build_model("ctrlora_finetune_sd15_rank128.yaml", 0).cuda().train()
    model.set_engine_dtype(torch.bfloat16)
    model.learning_rate = 0.0
    if a.one_stream:
        model.engine().overlap_streams = False
    opt = model.configure_optimizers()
    d = bench.synth(8, 64, model.control_model.context_dim, "cuda", 99, 1)
    args = (d["z"][0], d["ctx"][0], d["hint"][0], d["t"][0], d["noise"][0])
    g = GraphedTrainStep(model, opt, *args, warmup=1)
    ex = model.control_model.executor()
    rows = []
    for i in range(a.replays):
        loss = float(g(*args))
        torch.cuda.synchronize()
        fg = ex.tr.flat_grad
        rows.append((struct.pack("f", loss).hex(), float(fg.double().sum()), float(fg.double().abs().sum()),
                     float(ex.tr.flat.double().abs().sum())))
    same_loss = len({r[0] for r in rows}) == 1
    same_grad = len({(r[1], r[2]) for r in rows}) == 1
    same_w = len({r[3] for r in rows}) == 1
    print(f"[{a.tag}] variant={a.variant} one_stream={a.one_stream} PRESCALE_Q={os.environ.get('CTRLORA_PRESCALE_Q', '1')} "
          f"HOIST={os.environ.get('CTRLORA_HOIST_EMB_BWD', '1')}: loss_same={same_loss} grad_same={same_grad} weights_same={same_w}")
    for r in rows:
        print("    ", r)


if __name__ == "__main__":
    main()
