"""FrozenCLIPEmbedder (API of ldm/modules/encoders/modules.py:88-131).

The frozen CLIP ViT-L/14 text encoder is outside the hand-written path (about 1 % of the FLOPs,
SURVEY.md 2.1 row 14); it runs through HF transformers.  There is no network in the build / bench environment, so
construction never downloads; benchmarks feed a synthetic (B,77,768) context.
"""
import torch
import torch.nn as nn


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


def _hash_tokens(texts, max_length, vocab=49408, bos=49406, eos=49407):
    """Offline stand-in for CLIPTokenizer (its vocab / merges files are not in this image and there is no network):
    lower-cased whitespace words hashed into the word-piece range, BOS ... EOS, EOS padding -- same tensor layout, NOT the
    real BPE.  Only used when CTRLORA_SYNTHETIC_TOKENIZER=1 (synthetic-data smoke runs of the training / sampling scripts)."""
    import zlib
    ids = torch.full((len(texts), max_length), eos, dtype=torch.long)
    for i, t in enumerate(texts):
        toks = [bos] + [zlib.crc32(w.encode()) % (bos - 1) + 1 for w in t.lower().split()][:max_length - 2] + [eos]
        ids[i, :len(toks)] = torch.tensor(toks)
    return ids


class FrozenCLIPEmbedder(AbstractEncoder):
    """CLIP ViT-L/14 text encoder through HF transformers (ldm/modules/encoders/modules.py:88-131).  The module tree
    (`transformer.text_model.*`, the keys an SD checkpoint carries under `cond_stage_model.`) is built from the model's
    CONFIG -- no download; pretrained weights are picked up from a local HF cache when one exists, otherwise they come
    from the SD checkpoint the scripts load.  The tokenizer needs its vocabulary files locally."""
    LAYERS = ["last", "pooled", "hidden"]

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, freeze=True, layer="last",
                 layer_idx=None):
        super().__init__()
        assert layer in self.LAYERS
        from transformers import CLIPTextConfig, CLIPTextModel
        self.version, self.device, self.max_length, self.layer, self.layer_idx = version, device, max_length, layer, layer_idx
        self.weights_from_checkpoint = False
        try:
            self.transformer = CLIPTextModel.from_pretrained(version, local_files_only=True)
        except (OSError, ValueError) as e:      # no local HF files (offline): anything else is a real error and propagates
            import warnings
            warnings.warn(f"CLIP text encoder '{version}': no local pretrained files ({type(e).__name__}); the module is built "
                          "from its config with RANDOM weights -- they must come from the SD checkpoint's cond_stage_model.* "
                          "keys (load_state_dict), otherwise the text conditioning is garbage")
            self.weights_from_checkpoint = True
            cfg = CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                 num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu",
                                 projection_dim=768)
            self.transformer = CLIPTextModel(cfg)
        self.tokenizer = None
        if layer == "hidden":
            assert layer_idx is not None and 0 <= abs(layer_idx) <= 12
        if freeze:
            self.freeze()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for p in self.parameters():
            p.requires_grad = False

    def _tokens(self, text):
        import os
        if self.tokenizer is None:
            try:
                from transformers import CLIPTokenizer
                self.tokenizer = CLIPTokenizer.from_pretrained(self.version, local_files_only=True)
            except Exception:
                if os.environ.get("CTRLORA_SYNTHETIC_TOKENIZER") != "1":
                    raise RuntimeError(f"CLIP tokenizer files for '{self.version}' are not available locally (no network); "
                                       "set CTRLORA_SYNTHETIC_TOKENIZER=1 for synthetic-data smoke runs")
                self.tokenizer = "synthetic"
        if self.tokenizer == "synthetic":
            return _hash_tokens(list(text), self.max_length)
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return enc["input_ids"]

    @torch.no_grad()
    def forward(self, text):
        if isinstance(text, str):
            text = [text]
        dev = next(self.transformer.parameters()).device
        out = self.transformer(input_ids=self._tokens(text).to(dev), output_hidden_states=self.layer == "hidden")
        if self.layer == "last":
            return out.last_hidden_state
        if self.layer == "pooled":
            return out.pooler_output[:, None, :]
        return out.hidden_states[self.layer_idx]

    def encode(self, text):
        return self(text)
