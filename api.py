"""High-level inference API (same surface as the reference's api.py:15-163).

    from api import CtrLoRA
    ctrlora = CtrLoRA(num_loras=1)
    ctrlora.create_model(sd_file=..., basecn_file=..., lora_files=(...,))
    images = ctrlora.sample(cond_image_paths=..., prompt=..., n_prompt=..., num_samples=..., ddim_steps=..., scale=...)

`create_model` assembles the inference model from three kinds of checkpoint: SD1.5 (UNet, VAE, CLIP), the Base
ControlNet (everything under `control_model.` that is NOT LoRA-file material) and one LoRA file per slot
(`check_key` material: LoRA layers, zero convs, norm layers), each loaded into its switchable bank with the
reference's sequence switch_lora(i) -> load_state_dict(strict=False) -> copy_weights_to_switchable().
The denoising loop runs on the MI355X engine (cldm.ddim_hacked.DDIMSampler -> ControlInferenceLDM.apply_model).
"""
import os

import numpy as np
import torch

from cldm.ddim_hacked import DDIMSampler
from cldm.model import create_model, load_state_dict

_HERE = os.path.dirname(os.path.abspath(__file__))


def hwc3(x: np.ndarray) -> np.ndarray:
    """uint8 image -> H x W x 3 (grey replicated; RGBA composited on white), as annotator/util.py:11-27."""
    assert x.dtype == np.uint8
    if x.ndim == 2:
        x = x[:, :, None]
    assert x.ndim == 3 and x.shape[2] in (1, 3, 4)
    if x.shape[2] == 3:
        return x
    if x.shape[2] == 1:
        return np.repeat(x, 3, axis=2)
    rgb, a = x[:, :, :3].astype(np.float32), x[:, :, 3:4].astype(np.float32) / 255.0
    return (rgb * a + 255.0 * (1.0 - a)).clip(0, 255).astype(np.uint8)


def center_crop_to_common(a: np.ndarray, b: np.ndarray):
    """Crop two H x W x C images around their centres to the smaller height and the smaller width (api.py:113-126)."""
    def crop(img, H, W):
        h, w = img.shape[:2]
        if h > H:
            img = img[(h - H) // 2:(h + H) // 2]
        if w > W:
            img = img[:, (w - W) // 2:(w + W) // 2]
        return img
    H, W = min(a.shape[0], b.shape[0]), min(a.shape[1], b.shape[1])
    a, b = crop(a, H, W), crop(b, H, W)
    assert a.shape[:2] == b.shape[:2]
    return a, b


class CtrLoRA:
    def __init__(self, num_loras=1):
        self.model = None
        self.num_loras = num_loras
        if num_loras not in (1, 2):
            raise ValueError("Invalid number of LoRAs. Only 1 or 2 are supported.")
        name = "ctrlora_sd15_rank128_1lora.yaml" if num_loras == 1 else "ctrlora_sd15_rank128_2loras.yaml"
        self.config_file = os.path.join("configs", "inference", name)

    @staticmethod
    def check_key(k):
        return "lora_layer" in k or "zero_convs" in k or "middle_block_out" in k or "norm" in k

    # ------------------------------------------------------------------ checkpoint assembly
    def load_weights(self, model, sd_state_dict=None, cn_state_dict=None, lora_state_dicts=()):
        """The load sequence of create_model on already-read state dicts (api.py:46-62)."""
        if sd_state_dict is not None:
            model.load_state_dict(sd_state_dict, strict=False)
        if cn_state_dict is not None:
            base = {k: v for k, v in cn_state_dict.items() if k.startswith("control_model") and not self.check_key(k)}
            model.load_state_dict(base, strict=False)
        for i, lora_sd in enumerate(lora_state_dicts):
            lora = {k: v for k, v in lora_sd.items() if self.check_key(k)}
            model.control_model.switch_lora(i)
            model.load_state_dict(lora, strict=False)
            model.control_model.copy_weights_to_switchable()
        return model

    def create_model(self, sd_file="ckpts/sd15/v1-5-pruned.ckpt",
                     basecn_file="ckpts/ctrlora-basecn/ctrlora_sd15_basecn700k.ckpt",
                     lora_files=("ckpts/ctrlora-loras/novel-conditions/"
                                 "ctrlora_sd15_basecn700k_lineart_rank128_1kimgs_1ksteps.ckpt",)):
        if not isinstance(lora_files, (tuple, list)):
            lora_files = (lora_files,)
        for f in (sd_file, basecn_file, *lora_files):
            assert os.path.exists(f), f"File not found: {f}"
        cfg = self.config_file if os.path.exists(self.config_file) else os.path.join(_HERE, self.config_file)
        self.model = create_model(cfg).cuda()
        self.load_weights(self.model, sd_state_dict=load_state_dict(sd_file, location="cpu"))
        self.load_weights(self.model, cn_state_dict=load_state_dict(basecn_file, location="cpu"))
        self.load_weights(self.model, lora_state_dicts=[load_state_dict(f, location="cpu") for f in lora_files])

    # ------------------------------------------------------------------ sampling
    def sample(self, cond_image_paths, prompt, n_prompt="", num_samples=1, ddim_steps=20, scale=7.5,
               lora_weights=(1.0, 1.0)):
        from PIL import Image
        assert self.model is not None, "Model is not loaded. Please call create_model() first."
        if not isinstance(cond_image_paths, (tuple, list)):
            cond_image_paths = (cond_image_paths,)
        assert len(cond_image_paths) == self.num_loras, f"Expected {self.num_loras} images, got {len(cond_image_paths)}"
        images = [hwc3(np.array(Image.open(p))) for p in cond_image_paths]
        if self.num_loras == 1:
            return self.sample_1lora(images[0], prompt, n_prompt, num_samples, ddim_steps, scale)
        return self.sample_2loras(images, prompt, n_prompt, num_samples, ddim_steps, scale, lora_weights)

    def _control(self, image: np.ndarray, num_samples: int) -> torch.Tensor:
        c = torch.from_numpy(image.copy()).float().cuda() / 255.0           # H x W x 3 in [0, 1]
        return c.permute(2, 0, 1).unsqueeze(0).repeat(num_samples, 1, 1, 1).contiguous()

    @torch.no_grad()
    def _run(self, images, prompt, n_prompt, num_samples, ddim_steps, scale, lora_weights=None):
        from PIL import Image
        H, W, _ = images[0].shape
        m = self.model
        txt = m.get_learned_conditioning([prompt] * num_samples)
        ntxt = m.get_learned_conditioning([n_prompt] * num_samples)
        conds = [{"c_concat": [self._control(im, num_samples)], "c_crossattn": [txt]} for im in images]
        unconds = [{"c_concat": c["c_concat"], "c_crossattn": [ntxt]} for c in conds]
        m.control_scales = [1] * 13
        if lora_weights is not None:
            m.lora_weights = [lora_weights[0], lora_weights[1]]
        single = len(images) == 1
        samples, _ = DDIMSampler(m).sample(
            ddim_steps, num_samples, (4, H // 8, W // 8), conds[0] if single else conds, verbose=False, eta=0,
            unconditional_guidance_scale=scale, unconditional_conditioning=unconds[0] if single else unconds)
        x = m.decode_first_stage(samples)
        x = (x.permute(0, 2, 3, 1) * 127.5 + 127.5).cpu().numpy().clip(0, 255).astype(np.uint8)
        return [Image.fromarray(x[i]) for i in range(num_samples)]

    def sample_1lora(self, detected_image, prompt, n_prompt="", num_samples=1, ddim_steps=20, scale=7.5):
        return self._run([detected_image], prompt, n_prompt, num_samples, ddim_steps, scale)

    def sample_2loras(self, detected_images, prompt, n_prompt="", num_samples=1, ddim_steps=20, scale=7.5,
                      lora_weights=(1.0, 1.0)):
        a, b = center_crop_to_common(*detected_images)
        return self._run([a, b], prompt, n_prompt, num_samples, ddim_steps, scale, lora_weights)
