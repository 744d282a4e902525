"""Offline plugin: random-init LLaVA-architecture models with a toy tokenizer.

There are no weights, tokenizers or preprocessor configs in the build or benchmark
environment (no network), so end-to-end runs use a seeded random model of the right
ARCHITECTURE and synthetic token ids - "build vs pure-torch reference path on the same random
model" is what loss parity means there (hard parts).

    synthetic/tiny-llava      CLIP-like 56x56/14 tower, 2-layer 64-wide Llama: CPU-runnable
    synthetic/llava-1.5-7b    LlavaConfig() defaults = CLIP-L/14-336 + Llama-7B, random init

The processor object duck-types what the trainers touch: `.image_processor`
(image_mean/std, crop_size, do_convert_rgb), `.tokenizer` (encode, __call__, pad ids,
padding_side, decode), `.apply_chat_template`, and `__call__(text=, images=)` which expands
the image placeholder to one token per vision patch like the HF LLaVA processor does.
"""
import zlib
from types import SimpleNamespace

import torch

from ..plan import CLIP_MEAN, CLIP_STD
from ..processors.llavaprocessor import AdvLlavaInputs, DifferentiableLlavaImageProcessor  # noqa: F401  (registry looks them up here)

IMAGE_TOKEN = "<image>"


class ToyTokenizer:
    """Whitespace tokenizer with a stable hash vocabulary; id 0 = pad, 1 = BOS, 2 = EOS."""
    padding_side = "left"
    pad_token_id = 0
    bos_token_id = 1
    eos_token_id = 2

    def __init__(self, vocab_size, image_token_id):
        self.vocab_size = vocab_size
        self.image_token_id = image_token_id

    def _word(self, w):
        if w == IMAGE_TOKEN:
            return self.image_token_id
        if w == "</s>":
            return self.eos_token_id
        i = 3 + zlib.crc32(w.encode()) % (self.vocab_size - 4)
        return i + 1 if i == self.image_token_id else i     # a word never aliases the placeholder

    def _ids(self, text, add_special_tokens=True):
        text = text.replace(IMAGE_TOKEN, f" {IMAGE_TOKEN} ").replace("</s>", " </s> ")
        ids = [self._word(w) for w in text.split()]
        return ([self.bos_token_id] + ids) if add_special_tokens else ids

    def encode(self, text, add_special_tokens=True):
        return self._ids(text, add_special_tokens)

    def __call__(self, text, return_tensors=None, add_special_tokens=True, **kw):
        ids = self._ids(text, add_special_tokens)
        return SimpleNamespace(input_ids=torch.tensor([ids], dtype=torch.long))

    def decode(self, ids, skip_special_tokens=False):
        return " ".join(f"<{int(i)}>" for i in ids)


class ToyLlavaProcessor:
    def __init__(self, vocab_size, image_token_id, image_size, patch_size):
        self.tokenizer = ToyTokenizer(vocab_size, image_token_id)
        self.image_processor = SimpleNamespace(image_mean=list(CLIP_MEAN), image_std=list(CLIP_STD), do_convert_rgb=True,
                                               crop_size={"height": image_size, "width": image_size})
        self.num_image_tokens = (image_size // patch_size) ** 2
        self.image_token_id = image_token_id

    def apply_chat_template(self, messages, add_generation_prompt=False):
        parts = []
        for m in messages:
            role = "USER:" if m["role"] == "user" else "ASSISTANT:"
            body = " ".join(IMAGE_TOKEN if c["type"] == "image" else c["text"] for c in m["content"])
            # a finished assistant turn ends with the EOS token, so a training row ends with
            # target + extra token - the alignment the suffix loss relies on (Q7)
            parts.append(f"{role} {body}" + (" </s>" if m["role"] == "assistant" else ""))
        if add_generation_prompt:
            parts.append("ASSISTANT:")
        return " ".join(parts)

    def __call__(self, text=None, images=None, padding=True, return_tensors="pt"):
        from transformers.feature_extraction_utils import BatchFeature
        rows = []
        for t in text:
            ids = []
            for i in self.tokenizer.encode(t):
                ids.extend([self.image_token_id] * self.num_image_tokens if i == self.image_token_id else [i])
            rows.append(torch.tensor(ids, dtype=torch.long))
        L = max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.tokenizer.pad_token_id, dtype=torch.long)
        att = torch.zeros((len(rows), L), dtype=torch.long)
        for k, r in enumerate(rows):
            ids[k, L - len(r):] = r
            att[k, L - len(r):] = 1
        return BatchFeature({"input_ids": ids, "attention_mask": att})


def _config(model_name):
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig
    if model_name == "synthetic/tiny-llava":
        vc = CLIPVisionConfig(image_size=56, patch_size=14, hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                              num_attention_heads=2, projection_dim=32)
        tc = LlamaConfig(vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                         num_key_value_heads=4, max_position_embeddings=512)
        return LlavaConfig(vision_config=vc, text_config=tc, image_token_index=511, vision_feature_layer=-1,
                           vision_feature_select_strategy="default")
    if model_name == "synthetic/llava-1.5-7b":
        # defaults ARE CLIP-L/14-336 + Llama-7B (transformers configuration_llava.py); the released
        # checkpoint extends the vocabulary to 32064 so that image_token_index 32000 is a valid row
        cfg = LlavaConfig()
        cfg.text_config.vocab_size = 32064
        return cfg
    raise ValueError(model_name)


def load_model_and_processor(model_name: str, device, seed: int = 0, dtype=None):
    from transformers import LlavaForConditionalGeneration
    cfg = _config(model_name)
    big = model_name.endswith("7b")
    dtype = dtype or (torch.float16 if big else torch.float32)
    torch.manual_seed(seed)
    if big:
        with torch.device(device):
            model = LlavaForConditionalGeneration(cfg).to(dtype)
    else:
        model = LlavaForConditionalGeneration(cfg).to(dtype).to(device)
    model.eval().requires_grad_(False)
    image_token = getattr(cfg, "image_token_index", None) or getattr(cfg, "image_token_id")
    rows = model.get_input_embeddings().weight.shape[0]
    if not (0 <= image_token < rows) or cfg.text_config.vocab_size > rows:
        raise ValueError(f"image token id {image_token} / vocab {cfg.text_config.vocab_size} outside the {rows}-row embedding")
    proc = ToyLlavaProcessor(cfg.text_config.vocab_size, image_token, cfg.vision_config.image_size, cfg.vision_config.patch_size)
    return model, proc
