"""Base-ControlNet multi-task PRE-TRAINING (BASELINE.json configs[3]) on the HIP engine, against the
UNMODIFIED reference (tests/golden/pretrain.pt from tests/golden/make_golden_pretrain.py): three optimizer steps with
the task sequence hed, canny, hed through ControlPretrainLDM.p_losses -> backward -> configure_optimizers().step().
Checked per step: the loss and the gradient of EVERY control_model parameter (conv weights -- incl. the input conv
and the stride-2 Downsample -- linears, biases, all norms, zero convs, the task's LoRA factors; the other bank: no
gradient before its first use, a zero gradient afterwards), then selected parameters after the three AdamW steps
(the bank of the idle task keeps moving once it has been used: torch 1.13 zero_grad semantics)."""
import os

import pytest
import torch

from tests.util import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu


def _model(dtype):
    import bench
    from oracle import arch
    from tests.golden.make_golden_pretrain import SEED, TASKS, bank_state
    cfg = arch.TINY

    def mutate(p):
        p["control_stage_config"]["params"]["tasks"] = list(TASKS)

    m = bench.build_model("ctrlora_pretrain_sd15_9tasks_rank128.yaml", 0, tiny=True, mutate=mutate)
    m.model.diffusion_model.load_state_dict(arch.make_state(arch.unet_shapes(cfg), SEED), strict=True)
    cm = m.control_model
    cm.switch_lora("hed")
    cm.load_state_dict(arch.make_state(arch.controlnet_shapes(cfg), SEED), strict=False)
    cm.switch_lora("canny")
    cm.load_state_dict(bank_state(cfg, SEED + 1), strict=False)
    m = m.cuda().train()
    m.set_engine_dtype(dtype)
    return m, cfg


@pytest.mark.parametrize("dtype,tol_g,tol_p", [(torch.float32, 1e-3, 5e-3), (torch.bfloat16, 2.6e-1, 3e-1)])
def test_pretraining_three_steps_two_tasks_vs_reference(dtype, tol_g, tol_p):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.golden.make_golden_pretrain import LR, SEQ, TASKS, digest, step_inputs, unpack
    gold = torch.load(os.path.join(GOLDEN, "pretrain.pt"), weights_only=False)
    m, cfg = _model(dtype)
    m.learning_rate = LR
    opt = m.configure_optimizers()
    cm = m.control_model
    cu = lambda v: v.cuda()

    def current(kind):
        out = {}
        for t in TASKS:
            for j, lora in enumerate(cm.loras_dict[t]):
                for part in ("down", "up"):
                    w = getattr(lora, part).weight
                    out[f"loras_dict.{t}.{j}.{part}.weight"] = w.grad if kind == "grad" else w
        for n, p in cm.named_parameters():
            if "lora_layer" in n or n.startswith("loras_dict."):
                continue
            out[n] = p.grad if kind == "grad" else p
        return out

    for i, task in enumerate(SEQ):
        inp = step_inputs(cfg, i)
        cond = dict(c_crossattn=[cu(inp["ctx"])], c_concat=[cu(inp["hint_z"])], task=task)
        opt.zero_grad()
        loss, _ = m.p_losses(cu(inp["z"]), cond, cu(inp["t"]), noise=cu(inp["noise"]))
        loss.backward()
        torch.cuda.synchronize()
        g = gold["steps"][i]
        assert g["task"] == task
        assert abs(float(loss) - g["loss"]) < (1e-4 if dtype == torch.float32 else 3e-2) * g["loss"], (i, float(loss), g["loss"])
        ref = unpack(g["grads"])
        ours = current("grad")
        assert set(ref) == set(ours)
        worst = []
        for n, r in ref.items():
            bank = n.split(".")[1] if n.startswith("loras_dict.") else None
            if r is None:        # the reference has not touched this bank yet: no gradient, and the optimizer skips it
                assert bank is not None and bank not in opt.active
                assert float(ours[n].abs().max()) == 0.0
                continue
            l2, v = digest(ours[n])
            if r[0] < 1e-12:     # idle but already active bank: zero gradient on both sides
                assert l2 < 1e-12, n
                continue
            e = max(rel_l2(v, r[1]), abs(l2 - r[0]) / r[0])
            worst.append((e, n))
        worst.sort(reverse=True)
        print(f"[pretrain {dtype} step {i} {task}] loss {float(loss):.6f} (ref {g['loss']:.6f}); worst grad errors {worst[:3]}")
        # bf16: the first step measures the kernels (6e-2 at this d_head-8 width); from the second step on the two parameter
        # trajectories have drifted apart (AdamW's first updates are lr * sign(g), lr = 1e-3 on 0.02-scale weights), which
        # test_pretraining_bf16_error_is_flat_on_the_fp32_parameter_trajectory separates from kernel error: on the fp32
        # trajectory the error stays at 5e-2.  Free-running it was measured at 1.3e-1 / 1.8e-1 .. 2.0e-1 (chaotic in the
        # last digit of every bf16 rounding), hence the looser gate for steps 1, 2.
        gate = tol_g if (dtype == torch.float32 or i > 0) else 8e-2
        assert worst[0][0] < gate, worst[:5]
        opt.step()
    assert opt.active == ["hed", "canny"]
    ref = unpack(gold["after"])
    ours = current("param")
    errs = sorted(((max(rel_l2(digest(ours[n])[1], r[1]), abs(digest(ours[n])[0] - r[0]) / r[0]), n) for n, r in ref.items()), reverse=True)
    print(f"[pretrain {dtype}] worst parameter errors after 3 steps {errs[:3]}")
    assert errs[0][0] < tol_p, errs[:5]
