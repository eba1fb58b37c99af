"""Emit configs/*.yaml: the `target:` / `params:` trees the cldm.* classes are instantiated from.

The values are the SD1.5 / CtrLoRA hyper-parameters the reference's configs/*.yaml carry (that is the
drop-in contract: same keys, same targets); the files are generated so that the tree is defined in
one place.  Usage: python tools/gen_configs.py
"""
import copy
import os

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NET = dict(image_size=32, in_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
           channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768,
           use_checkpoint=True, legacy=False)

VAE = dict(target="ldm.models.autoencoder.AutoencoderKL", params=dict(
    embed_dim=4, monitor="val/rec_loss",
    ddconfig=dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                  num_res_blocks=2, attn_resolutions=[], dropout=0.0),
    lossconfig=dict(target="torch.nn.Identity")))


def ldm(kind, control_params):
    mod = f"cldm.cldm_ctrlora_{kind}"
    cls = {"finetune": ("ControlFinetuneLDM", "ControlNetFinetune"), "pretrain": ("ControlPretrainLDM", "ControlNetPretrain"),
           "inference": ("ControlInferenceLDM", "ControlNetInference")}[kind]
    control = dict(copy.deepcopy(NET), hint_channels=3, **control_params)
    unet = dict(copy.deepcopy(NET), out_channels=4)
    return dict(model=dict(target=f"{mod}.{cls[0]}", params=dict(
        linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000,
        first_stage_key="jpg", cond_stage_key="txt", control_key="hint", image_size=64, channels=4,
        cond_stage_trainable=False, conditioning_key="crossattn", monitor="val/loss_simple_ema", scale_factor=0.18215,
        use_ema=False, only_mid_control=False,
        control_stage_config=dict(target=f"{mod}.{cls[1]}", params=control),
        unet_config=dict(target="cldm.cldm.ControlledUnetModel", params=unet),
        first_stage_config=copy.deepcopy(VAE),
        cond_stage_config=dict(target="ldm.modules.encoders.modules.FrozenCLIPEmbedder"))))


def write(rel, tree):
    path = os.path.join(ROOT, "configs", rel)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        yaml.safe_dump(tree, f, sort_keys=False, default_flow_style=None, width=100)
    print("wrote", rel)


if __name__ == "__main__":
    for r in (32, 64, 128, 256, 512):
        write(f"ctrlora_finetune_sd15_rank{r}.yaml", ldm("finetune", dict(ft_with_lora=True, lora_rank=r, norm_trainable=True)))
        write(f"inference/ctrlora_sd15_rank{r}_1lora.yaml", ldm("inference", dict(lora_rank=r, lora_num=1)))
    write("ctrlora_finetune_sd15_full.yaml", ldm("finetune", dict(ft_with_lora=False)))
    write("inference/ctrlora_sd15_rank128_2loras.yaml", ldm("inference", dict(lora_rank=128, lora_num=2)))
    write("ctrlora_pretrain_sd15_9tasks_rank128.yaml", ldm("pretrain", dict(
        lora_rank=128, tasks=["hed", "canny", "seg", "depth", "normal", "openpose", "hedsketch", "bbox", "outpainting"])))
