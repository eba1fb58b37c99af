"""FrozenCLIPEmbedder (API of ldm/modules/encoders/modules.py:88-131).

The frozen CLIP ViT-L/14 text encoder is outside the hand-written path (about 1 % of the FLOPs,
SURVEY.md 2.1 row 14); it runs through HF transformers when the weights are available locally.
There is no network in the build / bench environment, so construction never downloads: the
tokenizer / weights are loaded lazily on first `encode`, and benchmarks feed a synthetic (B,77,768) context.
"""
import torch
import torch.nn as nn


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class FrozenCLIPEmbedder(AbstractEncoder):
    LAYERS = ["last", "pooled", "hidden"]

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, freeze=True, layer="last",
                 layer_idx=None):
        super().__init__()
        assert layer in self.LAYERS
        self.version, self.device, self.max_length, self.layer, self.layer_idx = version, device, max_length, layer, layer_idx
        self.tokenizer = None
        self.transformer = None

    def _load(self):
        if self.transformer is None:
            from transformers import CLIPTextModel, CLIPTokenizer
            self.tokenizer = CLIPTokenizer.from_pretrained(self.version, local_files_only=True)
            self.transformer = CLIPTextModel.from_pretrained(self.version, local_files_only=True).eval()
            for p in self.transformer.parameters():
                p.requires_grad = False

    def freeze(self):
        pass

    @torch.no_grad()
    def forward(self, text):
        self._load()
        dev = next(self.transformer.parameters()).device
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        out = self.transformer(input_ids=enc["input_ids"].to(dev), output_hidden_states=self.layer == "hidden")
        if self.layer == "last":
            return out.last_hidden_state
        if self.layer == "pooled":
            return out.pooler_output[:, None, :]
        return out.hidden_states[self.layer_idx]

    def encode(self, text):
        return self(text)
