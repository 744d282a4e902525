"""Shared machinery of the four processor plugins.

The reference defines its plugin contract by example (`src/processors/abstract_processor.py:13-208`):
an `Adv*Inputs` class (prompt batch + target-suffix cross-entropy) and a
`Differentiable*ImageProcessor` class (`process(image) -> {"pixel_values", ...}`,
`pil_to_tensor`, `tensor2pil`).  The classes in this package keep those names, constructor
signatures, attributes and return dictionaries; what changed is underneath:

  * `process()` runs advx_emit / advx_collect (HIP) through `ops.ProcessFunction` instead of
    F.interpolate / pad / reshape-permute chains, and exposes `plan_for(H, W)` so that the
    trainer can drive the fused PGD kernels with the same geometry;
  * `get_inputs_train()` tokenises every (question, target) pair ONCE and assembles batches
    from the cache, instead of re-tokenising and re-preprocessing the original image B times
    per step (`llavaprocessor.py:80-108`); the model-specific index tensors the HF processor
    returns (aspect_ratio_ids / image_sizes / image_grid_thw ...) are captured with the cache.
"""
import random
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from .. import ops
from ..plan import Plan


class DifferentiableProcessorBase:
    """Common part of the four Differentiable*ImageProcessor classes."""

    def __init__(self, orig_processor, device):
        self.orig_processor = orig_processor
        self.device = device
        self.mean = tuple(float(v) for v in orig_processor.image_mean)
        self.std = tuple(float(v) for v in orig_processor.image_std)
        self.image_mean = torch.tensor(self.mean).view(-1, 1, 1).to(device)
        self.image_std = torch.tensor(self.std).view(-1, 1, 1).to(device)
        self.do_convert_rgb = getattr(orig_processor, "do_convert_rgb", True)
        self._plans: Dict[Tuple[int, int], Plan] = {}

    # -- geometry: one plan per native image size
    def _make_plan(self, H: int, W: int) -> Plan:
        raise NotImplementedError

    def plan_for(self, H: int, W: int) -> Plan:
        key = (int(H), int(W))
        if key not in self._plans:
            self._plans[key] = self._make_plan(*key)
        return self._plans[key]

    def _extras(self, plan: Plan) -> dict:
        return {}

    def index_tensors(self, H: int, W: int, batch: int) -> dict:
        """Integer side tensors of `batch` copies of an H x W image as the HF image processor of this model family
        returns them (none for LLaVA)."""
        return {}

    def process(self, image: torch.Tensor) -> dict:
        """image: float tensor [3,H,W] in [0,1] on a ROCm device (may require grad)."""
        plan = self.plan_for(image.shape[1], image.shape[2])
        data = {"pixel_values": ops.ProcessFunction.apply(image, plan)}
        data.update(self._extras(plan))
        return data

    # -- PIL bridges (llavaprocessor.py:151-161 and the identical copies in the other plugins)
    def tensor2pil(self, x: torch.Tensor) -> Image.Image:
        x = x.clamp(0, 1)
        arr = (x * 255).cpu().detach().permute(1, 2, 0).numpy().astype(np.uint8)   # truncation (Q1)
        return Image.fromarray(arr)

    def pil_to_tensor(self, image: Image.Image, resize: bool = False) -> torch.Tensor:
        if self.do_convert_rgb:
            image = image.convert("RGB")
        t = torch.tensor(np.array(image).astype(np.float32) / 255).permute(2, 0, 1)
        if resize:
            raise NotImplementedError("resize=True is not used by the trainers (attack_model.py:170)")
        return t


class AdvInputsBase:
    """Prompt batches and the target-suffix loss, shared by the four Adv*Inputs classes.

    Subclasses set: `refuses`, `extra_token`, `_shift(tokenizer)` and `_conversation(q, a)`
    (how one training prompt is rendered).  Constructor signature follows
    `llavaprocessor.py:35-44`.
    """
    refuses: List[str] = []
    extra_token: str = ""
    PER_TOKEN_KEYS = ("mm_token_type_ids", "token_type_ids")     # [1, S] tensors of a sample that follow input_ids' padding

    def __init__(self, questions: list, test_questions: list, batch_size: int, original_image, processor,
                 device: str = "cuda:0", target_text: Union[str, list] = "sure, here it is!", rng=None):
        self.questions = questions
        self.test_questions = test_questions
        self.batch_size = batch_size
        self.processor = processor
        self.original_image = original_image
        self.device = device
        self.rng = rng if rng is not None else random       # reference uses the global `random`
        self.shift = self._shift(processor.tokenizer)
        if isinstance(target_text, list):
            self.target_texts = target_text
            self.target_text = target_text[0]
        else:
            self.target_texts = [target_text]
            self.target_text = target_text
        self._cache: Dict[Tuple[str, str], dict] = {}
        self._geometry = None       # (differentiable processor, H, W): source of the index tensors, see bind_geometry
        self.update_target_tokens()

    # ---- per-plugin hooks
    def _shift(self, tokenizer) -> int:
        return len(tokenizer.encode(self.extra_token))

    def _render_train(self, question: str, answer: str) -> str:
        raise NotImplementedError

    def _render_inference(self, question: str) -> str:
        raise NotImplementedError

    def _encode(self, prompts: List[str], images: list):
        """One HF processor call (text + images) -> BatchFeature on CPU."""
        return self.processor(text=prompts, images=images, padding=True, return_tensors="pt")

    # ---- target handling (llavaprocessor.py:64-71)
    def update_target_tokens(self):
        tok = self.processor.tokenizer(self.target_text + self.extra_token, return_tensors="pt",
                                       add_special_tokens=False).input_ids.to(self.device)
        self.target_tokens = tok
        self.suffix_length = tok.shape[1]
        self.target = tok[:, :-self.shift].repeat(self.batch_size, 1).to(self.device)

    def set_target_text(self, target_text: str):
        self.target_text = target_text
        self.update_target_tokens()

    # ---- loss (llavaprocessor.py:73-78)
    def get_loss(self, logits: torch.Tensor) -> torch.Tensor:
        suffix = logits[:, -self.suffix_length:-self.shift, :].permute(0, 2, 1)
        return F.cross_entropy(suffix, self.target)

    def get_loss_suffix_only(self, model, inputs) -> torch.Tensor:
        """The same loss without the [B, S, V] logits tensor: the model computes
        its last suffix_length + 1 positions only (left-padded batches end with the target), and
        the supervised ones - positions [-suffix_length-1, -shift-1) of the full sequence, i.e.
        the first suffix_length - shift kept ones - go through the HIP log-softmax + NLL."""
        from ..ce import suffix_cross_entropy
        out = model(**inputs, logits_to_keep=self.suffix_length + 1)
        return suffix_cross_entropy(out.logits, self.target)

    # ---- index tensors from the plan geometry
    def bind_geometry(self, adv_processor, H: int, W: int):
        """The integer side tensors of a batch (Mllama aspect_ratio_ids / aspect_ratio_mask and the tile axis of
        cross_attention_mask, Qwen2-VL image_grid_thw, Phi-3.5 image_sizes) then come from the SAME geometry the
        pixel_values are made with (`adv_processor.index_tensors`, pinned bit for bit against the HF image
        processors by tests/test_index_tensors.py) instead of from an HF pass over the original image - the
        reference keeps whatever its per-step processor(...) call returned (llama32processor.py:119-147,
        qwen2VLprocessor.py:68-96, phi3processor.py:88-95)."""
        self._geometry = (adv_processor, int(H), int(W))

    # ---- cached batch assembly (replaces llavaprocessor.py:80-108)
    def _sample(self, question: str) -> dict:
        key = (question, self.target_text)
        hit = self._cache.get(key)
        if hit is None:
            enc = self._encode([self._render_train(question, self.target_text)], [self.original_image])
            hit = {k: v for k, v in enc.items() if k != "pixel_values"}
            self._cache[key] = hit
        return hit

    def get_inputs_train(self):
        from transformers.feature_extraction_utils import BatchFeature
        batch_questions = self.rng.choices(self.questions, k=self.batch_size)
        samples = [self._sample(q) for q in batch_questions]
        tok = self.processor.tokenizer
        pad_id = tok.pad_token_id if tok.pad_token_id is not None else 0
        left = getattr(tok, "padding_side", "right") == "left"
        ids = [s["input_ids"][0] for s in samples]
        L = max(int(t.shape[0]) for t in ids)
        input_ids = torch.full((len(ids), L), pad_id, dtype=ids[0].dtype)
        attention = torch.zeros((len(ids), L), dtype=torch.long)
        for r, t in enumerate(ids):
            n = int(t.shape[0])
            if left:
                input_ids[r, L - n:] = t
                attention[r, L - n:] = 1
            else:
                input_ids[r, :n] = t
                attention[r, :n] = 1
        data = {"input_ids": input_ids, "attention_mask": attention}
        # model-specific index tensors: identical for every sample (same image), batch them
        for k, v in samples[0].items():
            if k in ("input_ids", "attention_mask"):
                continue
            if torch.is_tensor(v):
                if k in self.PER_TOKEN_KEYS:
                    # one id per token (Qwen2-VL's modality ids of transformers 5.x): padded like input_ids, with zeros
                    full = torch.zeros((len(ids), L), dtype=v.dtype)
                    for r, s in enumerate(samples):
                        t = s[k][0]
                        n = int(t.shape[0])
                        if left:
                            full[r, L - n:] = t
                        else:
                            full[r, :n] = t
                    data[k] = full
                elif k == "cross_attention_mask":
                    # [1, S, images, tiles]: per-token; pad along S like the ids.  Left padding: pad tokens precede
                    # the image, rows of zeros.  Right padding: HF lets the last image's span run to the padded
                    # length (convert_sparse_cross_attention_mask_to_dense), i.e. the last row repeats.
                    rows = []
                    for s in samples:
                        c = s[k][0]
                        padn = L - int(c.shape[0])
                        if left:
                            z = torch.zeros((padn,) + tuple(c.shape[1:]), dtype=c.dtype)
                            rows.append(torch.cat([z, c], 0))
                        else:
                            rows.append(torch.cat([c, c[-1:].expand((padn,) + tuple(c.shape[1:]))], 0))
                    data[k] = torch.stack(rows)
                else:
                    data[k] = torch.cat([s[k] for s in samples], dim=0)
        if self._geometry is not None:
            adv, H, W = self._geometry
            own = adv.index_tensors(H, W, len(samples))
            for k, v in own.items():
                if k in data and torch.is_tensor(v):
                    data[k] = v.to(data[k].dtype)
            if "cross_attention_mask" in data and "aspect_ratio_mask" in own:
                # which tokens see the image comes from the tokenizer pass; WHICH TILES exist from the plan
                seen = (data["cross_attention_mask"].amax(dim=-1, keepdim=True) > 0).to(data["cross_attention_mask"].dtype)
                data["cross_attention_mask"] = seen * own["aspect_ratio_mask"][:, None, :, :].to(seen.dtype)
        return BatchFeature(data).to(torch.device(self.device))

    def get_inputs_inference(self, img, question: Optional[str] = None):
        if question is None:
            question = self.test_questions[0]
        enc = self._encode([self._render_inference(question)], [img])
        return enc.to(self.device)


def chat_template_render(processor, question: str, answer: Optional[str], image_first: bool) -> str:
    """`processor.apply_chat_template` with the message layout the reference uses
    (llavaprocessor.py:83-99 text-then-image; llama32processor.py:122-138 image-then-text)."""
    img, txt = {"type": "image"}, {"type": "text", "text": question}
    content = [img, txt] if image_first else [txt, img]
    msgs = [{"role": "user", "content": content}]
    if answer is not None:
        msgs.append({"role": "assistant", "content": [{"type": "text", "text": answer}]})
        return processor.apply_chat_template(msgs)
    return processor.apply_chat_template(msgs, add_generation_prompt=True)
