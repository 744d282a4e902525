"""Offline twin of the Phi-3.5-Vision interface: `synthetic/tiny-phi3v`.

Phi-3.5-Vision's processor and model are REMOTE code (`trust_remote_code=True`, phi3processor.py:25-36): nothing of
them ships with transformers, so no real class can stand in the loop here.  What the reference's trainer and plugin
rely on is an INTERFACE, and this module restates that interface from the model's published description, small
enough for a CPU:

  processor(prompt, [image], return_tensors="pt")  ->  input_ids [1, S] in which `<|image_1|>` became
        num_img_tokens copies of the NEGATIVE id -1, pixel_values [1, num_crops + 1, 3, 336, 336] (global view
        first, then the HD canvas cut into 336 x 336 crops row-major, then zero crops), image_sizes [1, 2] = the HD
        canvas (phi3processor.py:275-302 consumes exactly these keys);  .tokenizer (BOS on encode, so that the
        plugin's `shift` is 1 as with the real tokenizer, phi3processor.py:61), .image_processor with
        image_mean / image_std / num_crops / num_img_tokens (phi3processor.py:134-139)
  model(input_ids, attention_mask, pixel_values, image_sizes) -> .logits: every crop goes through a patch-14 vision
        stem (24 x 24 features), 2 x 2 neighbours are merged, the crops named by image_sizes are laid out as one
        feature image with a learned separator after each feature row, the global view follows behind a learned
        separator ("sub_glb" order), and the projected features replace the embeddings at the negative ids:
        (h*w + 1) * 144 + 1 + (h + 1) * 12 positions, the count the processor inserted.  Zero crops are never read.
        `generate` is a plain greedy loop (the probe's `out[out != -1]`, train_test.py:58).

It is a twin of the WIRING (negative ids, 5-D pixel_values, image_sizes, crop order, token count), not of the
weights or the exact tower; parity is taken between the HIP engine and the oracle around the SAME twin
(tests/test_gpu_e2e_families.py).  The language model is transformers' own LlamaForCausalLM.
"""
import re
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from PIL import Image

from ..plan import CLIP_MEAN, CLIP_STD
from ..processors.phi3processor import AdvPhiInputs, DifferentiablePhi3VImageProcessor  # noqa: F401  (registry looks them up here)
from .synthetic_vlms import _vocabulary

CROP = 336
PHI_SPECIAL = ["<|endoftext|>", "<s>", "<|user|>", "<|assistant|>", "<|end|>", "<unk>"]
PHI_NUM_CROPS = 4


def _tokenizer(words):
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", words.index("<s>"))])
    return PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", bos_token="<s>", pad_token="<|endoftext|>",
                                   eos_token="<|endoftext|>", padding_side="left",
                                   additional_special_tokens=["<|user|>", "<|assistant|>", "<|end|>"])


def hd_size(width, height, hd_num):
    """Size of the HD canvas of a width x height image: the longer side becomes scale * 336 with the largest scale whose
    crop count scale * ceil(scale / ratio) fits hd_num, the shorter side follows the ratio and is padded up to a
    multiple of 336.  -> (transposed, new_w, new_h, padded_h) in the landscape frame."""
    trans = width < height
    if trans:
        width, height = height, width
    ratio = width / height
    scale = 1
    while scale * np.ceil(scale / ratio) <= hd_num:
        scale += 1
    scale -= 1
    new_w = int(scale * CROP)
    new_h = int(new_w / ratio)
    return trans, new_w, new_h, int(np.ceil(new_h / CROP) * CROP)


class ToyPhi3VImageProcessor:
    def __init__(self, num_crops=PHI_NUM_CROPS):
        self.num_crops = num_crops
        self.num_img_tokens = 144
        self.image_mean, self.image_std = list(CLIP_MEAN), list(CLIP_STD)
        self.do_convert_rgb = True

    def __call__(self, images, return_tensors="pt"):
        pvs, sizes, ntok = [], [], []
        mean = torch.tensor(self.image_mean).view(3, 1, 1)
        std = torch.tensor(self.image_std).view(3, 1, 1)
        for im in images:
            im = im.convert("RGB")
            trans, new_w, new_h, pad_h = hd_size(im.size[0], im.size[1], self.num_crops)
            if trans:
                im = im.transpose(Image.TRANSPOSE)
            im = im.resize((new_w, new_h), Image.BILINEAR)
            top = (pad_h - new_h) // 2
            canvas = Image.new("RGB", (new_w, pad_h), (255, 255, 255))
            canvas.paste(im, (0, top))
            if trans:
                canvas = canvas.transpose(Image.TRANSPOSE)
            hd = (torch.tensor(np.asarray(canvas).astype(np.float32) / 255).permute(2, 0, 1) - mean) / std
            h, w = hd.shape[1], hd.shape[2]
            glob = F.interpolate(hd[None], size=(CROP, CROP), mode="bicubic")
            local = hd.reshape(1, 3, h // CROP, CROP, w // CROP, CROP).permute(0, 2, 4, 1, 3, 5).reshape(-1, 3, CROP, CROP)
            crops = torch.cat([glob, local], 0)
            if crops.shape[0] < self.num_crops + 1:
                crops = torch.cat([crops, torch.zeros(self.num_crops + 1 - crops.shape[0], 3, CROP, CROP)], 0)
            pvs.append(crops)
            sizes.append([h, w])
            ntok.append(int((h // CROP * (w // CROP) + 1) * 144 + 1 + (h // CROP + 1) * 12))
        return {"pixel_values": torch.stack(pvs), "image_sizes": torch.tensor(sizes, dtype=torch.long), "num_img_tokens": ntok}


class ToyPhi3VProcessor:
    """One prompt at a time, as the reference calls it (phi3processor.py:283-284)."""
    _tag = re.compile(r"<\|image_(\d+)\|>")

    def __init__(self, image_processor, tokenizer):
        self.image_processor, self.tokenizer = image_processor, tokenizer
        self.num_crops = image_processor.num_crops

    def __call__(self, text, images=None, return_tensors="pt"):
        from transformers.feature_extraction_utils import BatchFeature
        assert isinstance(text, str), "one prompt per call"
        img = self.image_processor(images, return_tensors=return_tensors) if images else None
        chunks = self._tag.split(text)                   # text, image number, text, ...
        ids = list(self.tokenizer(chunks[0]).input_ids)  # with BOS
        for k in range(1, len(chunks), 2):
            ids += [-int(chunks[k])] * img["num_img_tokens"][int(chunks[k]) - 1]
            ids += self.tokenizer(chunks[k + 1], add_special_tokens=False).input_ids
        input_ids = torch.tensor([ids], dtype=torch.long)
        data = {"input_ids": input_ids, "attention_mask": torch.ones_like(input_ids)}
        if img is not None:
            data["pixel_values"], data["image_sizes"] = img["pixel_values"], img["image_sizes"]
        return BatchFeature(data)


class TinyPhi3V(nn.Module):
    def __init__(self, vocab, hidden=32, vis=8, layers=2):
        super().__init__()
        from transformers import LlamaConfig, LlamaForCausalLM
        self.stem = nn.Conv2d(3, vis, kernel_size=14, stride=14)                       # 336 / 14 = 24 features a side
        self.pos = nn.Parameter(torch.randn(1, vis, 24, 24) * 0.1)
        self.tower = nn.Sequential(nn.Linear(vis, 2 * vis), nn.GELU(), nn.Linear(2 * vis, vis))
        self.glb_GN = nn.Parameter(torch.randn(1, 1, 4 * vis) * 0.5)
        self.sub_GN = nn.Parameter(torch.randn(1, 1, 1, 4 * vis) * 0.5)
        self.img_projection = nn.Sequential(nn.Linear(4 * vis, hidden), nn.GELU(), nn.Linear(hidden, hidden))
        self.lm = LlamaForCausalLM(LlamaConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=2 * hidden,
                                               num_hidden_layers=layers, num_attention_heads=2, num_key_value_heads=2,
                                               max_position_embeddings=2048, pad_token_id=0, bos_token_id=1, eos_token_id=0,
                                               initializer_range=0.15))   # default 0.02: logits and gradients ~ 0
        self.generation_config = SimpleNamespace(eos_token_id=0)

    # [n, 24, 24, C] -> [n, 12, 12, 4C]: the four neighbours of a 2 x 2 cell side by side
    @staticmethod
    def _merge(f):
        n, _, _, c = f.shape
        return f.reshape(n, 12, 2, 12, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(n, 12, 12, 4 * c)

    def image_features(self, pixel_values, image_sizes):
        B, n = pixel_values.shape[:2]
        f = self.stem(pixel_values.flatten(0, 1)) + self.pos                           # [B*n, C, 24, 24]
        f = f.permute(0, 2, 3, 1)
        f = f + self.tower(f)
        f = self._merge(f).reshape(B, n, 12, 12, -1)
        c4 = f.shape[-1]
        rows = []
        for b in range(B):
            h, w = int(image_sizes[b, 0]) // CROP, int(image_sizes[b, 1]) // CROP
            glb = torch.cat([f[b, :1], self.sub_GN.expand(1, 12, 1, c4)], dim=2).reshape(1, -1, c4)
            sub = f[b, 1:1 + h * w].reshape(1, h, w, 12, 12, c4).permute(0, 1, 3, 2, 4, 5).reshape(1, h * 12, w * 12, c4)
            sub = torch.cat([sub, self.sub_GN.expand(1, h * 12, 1, c4)], dim=2).reshape(1, -1, c4)
            rows.append(torch.cat([sub, self.glb_GN, glb], dim=1)[0])
        return self.img_projection(torch.cat(rows, 0))                                # [tokens of the batch, hidden]

    def _embed(self, input_ids, pixel_values, image_sizes):
        emb = self.lm.get_input_embeddings()(input_ids.clamp(min=0))
        where = input_ids < 0
        if pixel_values is not None and bool(where.any()):
            feats = self.image_features(pixel_values.to(emb.dtype), image_sizes)
            assert feats.shape[0] == int(where.sum()), (feats.shape, int(where.sum()))
            emb = emb.masked_scatter(where[..., None], feats)
        return emb

    def _lm(self, emb, attention_mask, **kw):
        pos = None
        if attention_mask is not None:
            pos = (attention_mask.long().cumsum(-1) - 1).clamp(min=0)
        return self.lm(inputs_embeds=emb, attention_mask=attention_mask, position_ids=pos, **kw)

    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, image_sizes=None, logits_to_keep=0, **unused):
        return self._lm(self._embed(input_ids, pixel_values, image_sizes), attention_mask, logits_to_keep=logits_to_keep)

    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, pixel_values=None, image_sizes=None, max_new_tokens=8,
                 do_sample=False, **unused):
        emb = self._embed(input_ids, pixel_values, image_sizes)
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        eos = self.generation_config.eos_token_id
        ids, done = input_ids, torch.zeros(input_ids.shape[0], dtype=torch.bool, device=input_ids.device)
        for _ in range(max_new_tokens):
            nxt = self._lm(emb, attention_mask, logits_to_keep=1).logits[:, -1].argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, eos), nxt)
            ids = torch.cat([ids, nxt[:, None]], 1)
            emb = torch.cat([emb, self.lm.get_input_embeddings()(nxt)[:, None]], 1)
            attention_mask = torch.cat([attention_mask, torch.ones_like(attention_mask[:, :1])], 1)
            done = done | (nxt == eos)
            if bool(done.all()):
                break
        return ids


def phi3v_processor(num_crops=PHI_NUM_CROPS):
    words = _vocabulary(PHI_SPECIAL)
    return ToyPhi3VProcessor(ToyPhi3VImageProcessor(num_crops), _tokenizer(words)), len(words)


def load_model_and_processor(model_name: str, device, seed: int = 0, dtype=torch.float32):
    if model_name != "synthetic/tiny-phi3v":
        raise ValueError(model_name)
    proc, n = phi3v_processor()
    torch.manual_seed(seed)
    model = TinyPhi3V(n).eval()
    model.requires_grad_(False)
    return model.to(dtype).to(device), proc
