"""Host-side scheduling logic that needs no GPU: the deferred stage reports of the side-stream weight gradients
(ctrlora_amd/engine/blocks.py:Ctx.flush_wgrad / retire_wgrad) and the candidate rules of the launch-table search
(tools/gemm_autotune.py).  Streams / events / the grouped launch are replaced by recorders."""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _FakeEvent:
    def __init__(self, log):
        self.log = log

    def record(self, stream=None):
        self.log.append(("record", id(stream)))


class _FakeStream:
    def __init__(self, log, name):
        self.log, self.name = log, name

    def wait_stream(self, other):
        self.log.append((self.name + ".wait_stream", other.name))

    def wait_event(self, ev):
        self.log.append((self.name + ".wait_event",))

    def __enter__(self):
        self.log.append(("enter", self.name)); return self

    def __exit__(self, *a):
        self.log.append(("exit", self.name))


def _ctx_with_fakes(monkeypatch, log, side=True):
    from ctrlora_amd import hip
    from ctrlora_amd.engine import blocks
    main = _FakeStream(log, "main")
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: main)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: s)
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: _FakeEvent(log))
    monkeypatch.setattr(hip, "weight_grad_tn_group", lambda probs: log.append(("group", len(probs))))
    monkeypatch.setattr(hip, "colsum", lambda dy, db, B, rows, scale=1.0: log.append(("colsum", rows)))
    ctx = blocks.Ctx(torch.bfloat16, "cpu", True)
    if side:
        ctx.wstream = _FakeStream(log, "side")
    return ctx


def test_inline_weight_gradients_report_their_stage_at_once(monkeypatch):
    log = []
    ctx = _ctx_with_fakes(monkeypatch, log, side=False)
    ctx.queue_wgrad("dy", "x", "dW")
    ctx.queue_bias_grad("dy", "db", 7)
    ctx.flush_wgrad(after=lambda: log.append(("report", 0)))
    assert log == [("colsum", 7), ("group", 1), ("report", 0)]


def test_side_stream_weight_gradients_report_one_stage_late_and_in_order(monkeypatch):
    """Stage i's group is launched on the side stream after it has waited for the main stream; its slice is reported
    only once the main stream has joined it -- at the next flush (before that flush's own launch, so that a segment
    graph can be cut there) or at retire_wgrad()."""
    log = []
    ctx = _ctx_with_fakes(monkeypatch, log)
    for stage in range(3):
        ctx.queue_wgrad("dy", "x", "dW")
        if stage == 1:
            ctx.queue_bias_grad("dy", "db", 5)
        ctx.flush_wgrad(after=lambda s=stage: log.append(("report", s)))
    ctx.retire_wgrad()
    ops = [e for e in log if e[0] in ("group", "colsum", "report", "main.wait_event", "side.wait_stream")]
    assert ops == [
        ("side.wait_stream", "main"), ("group", 1),                                   # stage 0 goes out, nothing reported
        ("main.wait_event",), ("report", 0), ("side.wait_stream", "main"), ("group", 1), ("colsum", 5),
        ("main.wait_event",), ("report", 1), ("side.wait_stream", "main"), ("group", 1),
        ("main.wait_event",), ("report", 2),                                          # retire: the last one
    ]
    assert ctx._wpending is None and not ctx._wq and not ctx._bq
    # a stage without weight gradients still reports, after the group in flight has been joined
    log.clear()
    ctx.queue_wgrad("dy", "x", "dW")
    ctx.flush_wgrad(after=lambda: log.append(("report", "a")))
    ctx.flush_wgrad(after=lambda: log.append(("report", "b")))
    assert [e for e in log if e[0] == "report"] == [("report", "a"), ("report", "b")]
    # the operands of a group in flight stay referenced until the join
    ctx.queue_wgrad("dy2", "x2", "dW2")
    ctx.flush_wgrad()
    assert ctx._wpending is not None and ctx._wpending[1][0][0][0] == "dy2"
    ctx.retire_wgrad()
    assert ctx._wpending is None


def _load_autotune():
    spec = importlib.util.spec_from_file_location("gemm_autotune", os.path.join(ROOT, "tools", "gemm_autotune.py"))
    mod = importlib.util.module_from_spec(spec)
    env = os.environ.get("CTRLORA_GEMM_TUNED")
    try:
        spec.loader.exec_module(mod)
    finally:                                  # the tool switches the table off for itself; not for this test process
        if env is None:
            os.environ.pop("CTRLORA_GEMM_TUNED", None)
        else:
            os.environ["CTRLORA_GEMM_TUNED"] = env
    return mod


def test_launch_table_search_offers_only_configurations_the_launcher_accepts():
    at = _load_autotune()
    from ctrlora_amd import hip
    # GEGLU epilogue: only configurations with a value / gate wave pair per 160-column tile
    cfgs, sks = at.candidates((hip.BF16, hip.LINEAR, 32768, 2560, 320, 0, 1))
    assert cfgs and set(cfgs) <= set(at.GEGLU_OK) and 0 in sks and 1 in sks and max(sks) * 2 <= 5
    # conv: no persistent forms, no second K segment involved; deep K offers split factors up to 16
    cfgs, sks = at.candidates((hip.BF16, hip.CONV_S1, 512, 1280, 1280, 0, 0))
    assert not set(cfgs) & set(at.PERSIST) and 16 in sks and 16 in cfgs
    # K not a multiple of 64: no full-line configuration
    cfgs, _ = at.candidates((hip.BF16, hip.LINEAR, 4096, 320, 96, 0, 0))
    assert not set(cfgs) & (set(at.FL) | set(at.PERSIST))
    # N = 128: 128-column tiles only; tiny M: the generic kernel only
    cfgs, _ = at.candidates((hip.BF16, hip.LINEAR, 32768, 128, 320, 0, 0))
    assert not set(cfgs) & set(at.W160)
    cfgs, _ = at.candidates((hip.BF16, hip.LINEAR, 8, 1280, 1280, 0, 0))
    assert not set(cfgs) & (set(at.FL) | set(at.PERSIST))
    # every configuration named anywhere is one the library accepts in a table row
    L = hip.lib()
    for c in set(at.FL) | set(at.PERSIST) | set(at.W160) | set(at.W128) | {0, 24}:
        assert L.cl_gemm_tune_set(0, 0, 1, 8, 32, 0, 0, c, 0) == 0
    hip.load_gemm_table(hip.GEMM_TABLE_PATH)
