"""Debug aid for an intermittent NaN in the segmented-graph training step (fp32, tiny model): repeats the scenario of
tests/test_gpu_parity.py::test_segmented_graph_step... and, when the graph's gradients differ from the eager ones, prints which
trainable tensors differ / are non-finite.  Not part of the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ctrlora_amd.train import GraphedTrainStep  # noqa: E402
from oracle import arch  # noqa: E402
from tests.golden.make_golden import inputs_for  # noqa: E402

cfg = arch.TINY
inp = inputs_for(cfg, 2, 16, 8)
cu = lambda v: v.cuda()
dtype = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == "f32") else torch.bfloat16
SPLIT = sys.argv[2] if len(sys.argv) > 2 else "segmented"
SPLIT = False if SPLIT == "one" else SPLIT
junk = []
held = []
from ctrlora_amd.engine import blocks as _blocks  # noqa: E402
if os.environ.get("DBG_NO_TCACHE"):
    def _fresh(self, x):
        M, Cc = x.shape
        Mp = _blocks.rup(M, 32)
        t = torch.empty((Cc, Mp), dtype=x.dtype, device=x.device)
        _blocks.hip.transpose(x, t, 1, M, Cc, Mp)
        return t
    _blocks.Ctx.transposed = _fresh
if os.environ.get("DBG_KZERO"):          # clears as ATen fill kernels instead of memset nodes
    def _kzero(t):
        return t.zero_()
    _blocks.hip.zero_ = _kzero
if os.environ.get("DBG_HOLD_ALL"):       # transposes as well
    _tr = _blocks.Ctx.transposed
    def _htr(self, x):
        t = _tr(self, x); held.append(t); held.append(x); return t
    _blocks.Ctx.transposed = _htr
if os.environ.get("DBG_SYNC_REPLAY"):
    _rp = torch.cuda.CUDAGraph.replay
    def _srp(self):
        _rp(self); torch.cuda.synchronize()
    torch.cuda.CUDAGraph.replay = _srp
BUCKET = (1 << 40) if os.environ.get("DBG_BIG_BUCKET") else (256 << 10)
if os.environ.get("DBG_HOLD") or os.environ.get("DBG_HOLD_ALL"):          # nothing allocated through Ctx is ever freed: no block reuse inside the captures
    _new, _zeros = _blocks.Ctx.new, _blocks.Ctx.zeros
    def _hnew(self, *a, **k):
        t = _new(self, *a, **k); held.append(t); return t
    def _hzeros(self, *a, **k):
        t = _zeros(self, *a, **k); held.append(t); return t
    _blocks.Ctx.new, _blocks.Ctx.zeros = _hnew, _hzeros


def make():
    m = bench.build_model("ctrlora_finetune_sd15_rank128.yaml", 0, tiny=True).cuda().train()
    m.set_engine_dtype(dtype)
    m.learning_rate = 1e-3
    if os.environ.get("DBG_NO_OVERLAP"):
        m.engine().overlap_streams = False
    return m, m.configure_optimizers()


def scenario(rep):
    ma, oa = make()
    cond = {"c_crossattn": [cu(inp["ctx"])], "c_concat": [cu(inp["hint_z"])]}
    losses, grads = [], []
    for _ in range(3):
        oa.zero_grad()
        loss, _ = ma.p_losses(cu(inp["z"]), cond, cu(inp["t"]), noise=cu(inp["noise"]))
        loss.backward()
        torch.cuda.synchronize()
        grads.append(ma.control_model.executor().tr.flat_grad.clone())
        oa.step()
        losses.append(float(loss))
    mb, ob = make()
    ex = mb.control_model.executor()
    g = GraphedTrainStep(mb, ob, cu(inp["z"]), cu(inp["ctx"]), cu(inp["hint_z"]), cu(inp["t"]), cu(inp["noise"]), warmup=1,
                         split_graphs=SPLIT, bucket_bytes=BUCKET, reduce_fn=lambda buf: None)
    out = []
    for k in (1, 2):
        l = float(g(cu(inp["z"]), cu(inp["ctx"]), cu(inp["hint_z"]), cu(inp["t"]), cu(inp["noise"])))
        torch.cuda.synchronize()
        gg = ex.tr.flat_grad.clone()
        bad = []
        for t in ex.tr.items:
            a = gg[t.offset:t.offset + t.master.numel()]
            b = grads[k][t.offset:t.offset + t.master.numel()]
            if not torch.isfinite(a).all() or float((a - b).abs().max()) > 1e-3 * (float(b.abs().max()) + 1e-12):
                bad.append((t.name, int((~torch.isfinite(a)).sum()), float((a - b).abs().max()), float(b.abs().max())))
        out.append((l, losses[k], bad))
    print(f"rep {rep}: segments {len(g.segments)}, " + "; ".join(f"replay {i + 1}: loss {l:.6f} (eager {le:.6f}), {len(bad)} bad tensors"
                                                             for i, (l, le, bad) in enumerate(out)), flush=True)
    for i, (l, le, bad) in enumerate(out):
        for b in bad[:12]:
            print(f"    replay {i + 1}: {b}", flush=True)
    # perturb the allocator state between repetitions: different amounts of garbage-filled memory returned to the cache
    junk.append(torch.full((int(1e6) * (rep + 1),), float("nan"), device="cuda"))
    if rep % 2:
        junk.clear()
    return any(bad for _, _, bad in out)


nbad = 0
for rep in range(int(os.environ.get("REPS", "8"))):
    nbad += int(scenario(rep))
print("failing repetitions:", nbad)
