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


# fp32 follows the reference at both learning rates.  bf16 runs FREE only at lr 1e-4: at the reference recipe's 1e-3 AdamW's first
# updates (lr * sign(g) on 0.02-scale weights) separate a bf16 trajectory from the fp32 one within a step -- what is measured
# from step 1 on is then parameter divergence (1.3e-1 .. 2.0e-1, chaotic), not kernel error; that case is gated where it can be:
# test_gpu_parity_r3.py::test_pretraining_bf16_error_is_flat_on_the_fp32_parameter_trajectory pins the parameters to the fp32
# trajectory at lr 1e-3 and requires a flat error.  At 1e-4 the trajectories stay together and every step is a kernel check.
# The bf16 run at the REFERENCE learning rate is back as a fourth case with two gates that say what they measure: step 0 --
# identical parameters on both sides -- at the kernel tolerance (8e-2); steps 1-2 and the final parameters at a DRIFT bound
# (2.6e-1 / 5e-2: round 3 measured 1.3e-1 .. 2.0e-1 and 2.8e-2 with correct kernels), which a regression that only shows through
# real AdamW updates at 1e-3 -- a wrong moment, a bank updated twice -- would exceed by an order of magnitude.
@pytest.mark.parametrize("dtype,fixture,tol_g,tol_p", [(torch.float32, "pretrain.pt", 1e-3, 5e-3),
                                                         (torch.float32, "pretrain_lr1e-4.pt", 1e-3, 5e-3),
                                                         (torch.bfloat16, "pretrain_lr1e-4.pt", 9e-2, 5e-3),
                                                         (torch.bfloat16, "pretrain.pt", (8e-2, 2.6e-1), 5e-2)])
def test_pretraining_three_steps_two_tasks_vs_reference(dtype, fixture, tol_g, tol_p):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.golden.make_golden_pretrain import SEQ, TASKS, digest, step_inputs, unpack
    gold = torch.load(os.path.join(GOLDEN, fixture), weights_only=False)
    LR = gold["meta"]["lr"]
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
        ltol = 1e-4 if dtype == torch.float32 else (3e-2 if not isinstance(tol_g, tuple) or i == 0 else 1e-1)
        assert abs(float(loss) - g["loss"]) < ltol * g["loss"], (i, float(loss), g["loss"])
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
        # (bf16 at this d_head-8 width, measured: 6.5e-2, 6.8e-2, 6.5e-2 -- flat; parameters after 3 steps 3.1e-3; the gate
        # holds for EVERY step)
        tol_i = tol_g if not isinstance(tol_g, tuple) else (tol_g[0] if i == 0 else tol_g[1])
        assert worst[0][0] < tol_i, worst[:5]
        opt.step()
    assert opt.active == ["hed", "canny"]
    ref = unpack(gold["after"])
    ours = current("param")
    errs = sorted(((max(rel_l2(digest(ours[n])[1], r[1]), abs(digest(ours[n])[0] - r[0]) / r[0]), n) for n, r in ref.items()), reverse=True)
    print(f"[pretrain {dtype}] worst parameter errors after 3 steps {errs[:3]}")
    assert errs[0][0] < tol_p, errs[:5]
