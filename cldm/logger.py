"""Training callbacks with the constructor arguments and file naming of the reference's `cldm.logger`
(cldm/logger.py:12-120), for `ctrlora_amd.trainer.Trainer` (same hook names as Lightning 1.5).

`CheckpointEveryNSteps` writes `<prefix>_epoch=<e>_global_step=<s>.ckpt` into the trainer's checkpoint directory
whenever `global_step == 0` or `(global_step + 1) % save_step_frequency == 0` -- the reference's rule, evaluated
after every micro-batch (so with gradient accumulation the same step can be written more than once, as there).
`ImageLogger` calls `module.log_images(batch, split=...)` under no_grad on the same schedule and stores 4-per-row
PNG grids under `<log_dir>/image_log/<split>/<key>/gs-XXXXXX_e-XXXXXX_b-XXXXXX.png` (rank 0 only).
"""
import os

import numpy as np
import torch


class CheckpointEveryNSteps:
    def __init__(self, save_step_frequency, prefix="N-Step-Checkpoint", use_modelcheckpoint_filename=False):
        self.save_step_frequency = save_step_frequency
        self.prefix = prefix
        self.use_modelcheckpoint_filename = use_modelcheckpoint_filename

    def check_frequency(self, check_idx):
        return check_idx == 0 or (check_idx + 1) % self.save_step_frequency == 0

    def on_batch_end(self, trainer, _module=None):
        step, epoch = trainer.global_step, trainer.current_epoch
        if not self.check_frequency(step):
            return
        name = (trainer.checkpoint_callback.filename if self.use_modelcheckpoint_filename
                else f"{self.prefix}_epoch={epoch}_global_step={step}.ckpt")
        trainer.save_checkpoint(os.path.join(trainer.checkpoint_callback.dirpath, name))


def _grid(images: torch.Tensor, nrow: int = 4, pad: int = 2) -> torch.Tensor:
    """(N, C, H, W) -> (C, rows*(H+pad)+pad, cols*(W+pad)+pad), zero padding, row-major (torchvision.utils.make_grid)."""
    n, c, h, w = images.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    out = images.new_zeros((c, rows * (h + pad) + pad, cols * (w + pad) + pad))
    for i in range(n):
        r, q = divmod(i, cols)
        out[:, pad + r * (h + pad): pad + r * (h + pad) + h, pad + q * (w + pad): pad + q * (w + pad) + w] = images[i]
    return out


class ImageLogger:
    def __init__(self, batch_frequency=2000, max_images=4, clamp=True, increase_log_steps=True, rescale=True,
                 disabled=False, log_on_batch_idx=False, log_first_step=False, log_images_kwargs=None):
        self.rescale, self.batch_freq, self.max_images, self.clamp = rescale, batch_frequency, max_images, clamp
        self.disabled, self.log_on_batch_idx, self.log_first_step = disabled, log_on_batch_idx, log_first_step
        self.log_images_kwargs = log_images_kwargs or {}

    def check_frequency(self, check_idx):
        return check_idx == 0 or (check_idx + 1) % self.batch_freq == 0

    def log_local(self, save_dir, split, images, global_step, current_epoch, batch_idx):
        from PIL import Image
        for key, batch in images.items():
            grid = _grid(batch, nrow=4)
            if self.rescale:
                grid = (grid + 1.0) / 2.0
            arr = (grid.permute(1, 2, 0).squeeze(-1).numpy() * 255).astype(np.uint8)
            path = os.path.join(save_dir, "image_log", split, key,
                                "gs-{:06}_e-{:06}_b-{:06}.png".format(global_step, current_epoch, batch_idx))
            os.makedirs(os.path.dirname(path), exist_ok=True)
            Image.fromarray(arr).save(path)

    def on_train_batch_end(self, trainer, module, outputs, batch, batch_idx, *args):
        if self.disabled or self.max_images <= 0 or not callable(getattr(module, "log_images", None)):
            return
        if not self.check_frequency(batch_idx if self.log_on_batch_idx else trainer.global_step):
            return
        was_training = module.training
        module.eval()
        with torch.no_grad():
            images = module.log_images(batch, split="train", **self.log_images_kwargs)
        out = {}
        for k, v in images.items():
            v = v[:min(v.shape[0], self.max_images)]
            if torch.is_tensor(v):
                v = v.detach().float().cpu()
                if self.clamp:
                    v = v.clamp(-1., 1.)
            out[k] = v
        if trainer.is_global_zero:
            self.log_local(trainer.log_dir, "train", out, trainer.global_step, trainer.current_epoch, batch_idx)
        if was_training:
            module.train()
