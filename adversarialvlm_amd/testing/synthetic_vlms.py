"""Offline plugins for the two non-LLaVA model families whose ARCHITECTURE ships with the installed transformers:
random-init Llama-3.2-Vision (mllama) and Qwen2-VL models small enough for a CPU, with toy tokenizers.

    synthetic/tiny-mllama     MllamaForConditionalGeneration: 56x56 tiles of 14x14 patches, max 4 tiles, one
                              cross-attention layer; the processor is the REAL `MllamaProcessor` around the real
                              `MllamaImageProcessorPil` and a word-level `PreTrainedTokenizerFast`
    synthetic/tiny-qwen2vl    Qwen2VLForConditionalGeneration: patch 14 / merge 2 / temporal 2 vision tower, M-RoPE
                              text model; `Qwen2VLProcessor` itself cannot be constructed here (it insists on a
                              `BaseVideoProcessor`, which the installed transformers replaces by its torchvision
                              dummy - DESIGN.md section 2), so the processor is `ToyQwen2VLProcessor`: the real
                              `Qwen2VLImageProcessorPil` and a real `PreTrainedTokenizerFast`, joined by the one step
                              the HF class adds - each `<|image_pad|>` becomes grid_t*grid_h*grid_w / merge^2 copies

    synthetic/mllama-11b      the released Llama-3.2-11B-Vision ARCHITECTURE (MllamaConfig defaults with 560-pixel tiles: 32 + 8
                              vision layers of width 1280, 40 text layers of width 4096, eight cross-attention layers), random
                              init, fp16 - for end-to-end timing (tools/e2e_bench.py); the real processor geometry (tile 560, 4 tiles)
    synthetic/qwen2-vl-7b     the Qwen2-VL-7B ARCHITECTURE (28 layers of width 3584, 28 / 4 heads, 32-layer vision tower), random
                              init, bf16; pixel bounds 56^2 .. 28^2 * 1280

These are what puts the wiring of `attack_model.py:314-328` with `llama32processor.py:119-147,360-405`
(`pixel_values [B,1,4,3,T,T]`, `aspect_ratio_ids/mask`, `cross_attention_mask`) and
`qwen2VLprocessor.py:68-96,211-272` (`[B*n_patches, 1176]` + `image_grid_thw`) in front of a model's `forward`
without weights or a network (tests/test_gpu_e2e_families.py).  Phi-3.5-Vision is remote code: its interface twin lives in synthetic_phi3v.py.
"""
import torch

from ..plan import CLIP_MEAN, CLIP_STD
from ..processors.llama32processor import AdvMllamaInputs, DifferentiableMllamaImageProcessor  # noqa: F401  (registry looks them up here)
from ..processors.qwen2VLprocessor import AdvQwen2VLInputs, DifferentiableQwen2VLImageProcessor  # noqa: F401

# a closed word list (word-level tokenizers have no hashing): everything else is <unk>
_COMMON = ("describe item number in this picture what is shown region of the image sure here it of course answer scene please "
           "hi now t list a an and user assistant yes no cannot sorry").split()
_NUMBERS = [str(i) for i in range(0, 40)]


def _vocabulary(special):
    words = list(special)
    for w in _COMMON + _NUMBERS:
        for form in (w, w.capitalize(), w + ".", w + "?", w + ",", w + "!"):
            if form not in words:
                words.append(form)
    return words


def _word_level_tokenizer(words, **special):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    return PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", padding_side="left", **special)


# ----------------------------------------------------------------------------------------------- Llama-3.2-Vision
MLLAMA_SPECIAL = ["<pad>", "<|begin_of_text|>", "<|eot_id|>", "<|image|>", "<|start_header_id|>", "<|end_header_id|>", "<unk>"]
MLLAMA_TEMPLATE = ("{% for m in messages %}<|start_header_id|> {{ m['role'] }} <|end_header_id|> "
                   "{% for c in m['content'] %}{% if c['type'] == 'image' %}<|image|> {% else %}{{ c['text'] }} {% endif %}{% endfor %}"
                   "<|eot_id|> {% endfor %}{% if add_generation_prompt %}<|start_header_id|> assistant <|end_header_id|> {% endif %}")
MLLAMA_TILE, MLLAMA_MAX_TILES = 56, 4


def mllama_processor(tile=MLLAMA_TILE, max_tiles=MLLAMA_MAX_TILES, padding_side="left"):
    from transformers.models.mllama.image_processing_pil_mllama import MllamaImageProcessorPil
    from transformers.models.mllama.processing_mllama import MllamaProcessor
    words = _vocabulary(MLLAMA_SPECIAL)
    fast = _word_level_tokenizer(words, pad_token="<pad>", bos_token="<|begin_of_text|>", eos_token="<|eot_id|>",
                                 additional_special_tokens=["<|image|>"])
    fast.padding_side = padding_side
    ip = MllamaImageProcessorPil(size={"height": tile, "width": tile}, max_image_tiles=max_tiles, image_mean=list(CLIP_MEAN),
                                 image_std=list(CLIP_STD))
    return MllamaProcessor(image_processor=ip, tokenizer=fast, chat_template=MLLAMA_TEMPLATE), len(words)


def mllama_11b(device, seed=0):
    """Random-init model of the released 11B architecture, built on the device in fp16 (22 GB)."""
    from transformers import MllamaConfig, MllamaForConditionalGeneration
    from transformers.models.mllama.configuration_mllama import MllamaTextConfig, MllamaVisionConfig
    torch.manual_seed(seed)
    cfg = MllamaConfig(vision_config=MllamaVisionConfig(image_size=560), text_config=MllamaTextConfig(pad_token_id=0),
                       image_token_index=MLLAMA_SPECIAL.index("<|image|>"))
    with torch.device(device):
        model = MllamaForConditionalGeneration(cfg).to(torch.float16)
    model.eval().requires_grad_(False)
    with torch.no_grad():
        for p in model.parameters():
            if p.numel() <= 16 and p.abs().max() == 0:      # the zero-initialised gates would hide the vision path
                p.fill_(0.5)
    return model


def qwen2vl_7b(device, seed=0):
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    torch.manual_seed(seed)
    ids = {w: i for i, w in enumerate(QWEN_SPECIAL)}
    cfg = Qwen2VLConfig(
        vision_config=dict(depth=32, embed_dim=1280, hidden_size=3584, mlp_ratio=4, num_heads=16, in_channels=3, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2),
        text_config=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
                         num_key_value_heads=4, max_position_embeddings=32768, pad_token_id=ids["<pad>"], bos_token_id=None,
                         eos_token_id=ids["<|im_end|>"],
                         rope_parameters={"rope_type": "default", "mrope_section": [16, 24, 24], "rope_theta": 1000000.0}),
        image_token_id=ids["<|image_pad|>"], video_token_id=ids["<|video_pad|>"], vision_start_token_id=ids["<|vision_start|>"],
        vision_end_token_id=ids["<|vision_end|>"])
    with torch.device(device):
        model = Qwen2VLForConditionalGeneration(cfg).to(torch.bfloat16)
    model.eval().requires_grad_(False)
    return model


def mllama_model(vocab_words, tile=MLLAMA_TILE, max_tiles=MLLAMA_MAX_TILES, seed=0):
    from transformers import MllamaConfig, MllamaForConditionalGeneration
    from transformers.models.mllama.configuration_mllama import MllamaTextConfig, MllamaVisionConfig
    torch.manual_seed(seed)
    vc = MllamaVisionConfig(hidden_size=32, num_hidden_layers=2, num_global_layers=1, attention_heads=2, intermediate_size=64,
                            image_size=tile, patch_size=14, max_num_tiles=max_tiles, vision_output_dim=96,
                            intermediate_layers_indices=[0, 1],
                            supported_aspect_ratios=[[1, 1], [1, 2], [1, 3], [1, 4], [2, 1], [2, 2], [3, 1], [4, 1]])
    tc = MllamaTextConfig(vocab_size=vocab_words, hidden_size=32, num_hidden_layers=3, cross_attention_layers=[1],
                          num_attention_heads=2, num_key_value_heads=2, intermediate_size=64, max_position_embeddings=256,
                          pad_token_id=MLLAMA_SPECIAL.index("<pad>"), bos_token_id=MLLAMA_SPECIAL.index("<|begin_of_text|>"),
                          eos_token_id=MLLAMA_SPECIAL.index("<|eot_id|>"))
    cfg = MllamaConfig(vision_config=vc, text_config=tc, image_token_index=MLLAMA_SPECIAL.index("<|image|>"))
    model = MllamaForConditionalGeneration(cfg).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.requires_grad_(False)
            if p.abs().max() == 0:      # the zero-initialised gates would hide the vision path altogether
                p.normal_(0, 0.5)
    return model


# ------------------------------------------------------------------------------------------------------ Qwen2-VL
QWEN_SPECIAL = ["<pad>", "<|im_start|>", "<|im_end|>", "<|image_pad|>", "<|vision_start|>", "<|vision_end|>", "<|video_pad|>",
                "<unk>"]
QWEN_TEMPLATE = ("{% for m in messages %}<|im_start|> {{ m['role'] }} {% for c in m['content'] %}"
                 "{% if c['type'] == 'image' %}<|vision_start|> <|image_pad|> <|vision_end|> {% else %}{{ c['text'] }} {% endif %}"
                 "{% endfor %}<|im_end|> {% endfor %}{% if add_generation_prompt %}<|im_start|> assistant {% endif %}")
QWEN_MIN_PIXELS, QWEN_MAX_PIXELS = 56 * 56, 28 * 28 * 16


class ToyQwen2VLProcessor:
    """What `Qwen2VLProcessor.__call__` does for text + images (transformers processing_qwen2_vl.py), around the real
    image processor and a real fast tokenizer: image_processor(images) -> pixel_values, image_grid_thw; the i-th
    `<|image_pad|>` of the batch is repeated grid_i.prod() / merge^2 times; tokenizer(text, padding)."""
    image_token = "<|image_pad|>"

    def __init__(self, image_processor, tokenizer, chat_template):
        self.image_processor, self.tokenizer, self.chat_template = image_processor, tokenizer, chat_template
        self.image_token_id = tokenizer.convert_tokens_to_ids(self.image_token)

    def apply_chat_template(self, messages, add_generation_prompt=False):
        from jinja2 import Template
        return Template(self.chat_template).render(messages=messages, add_generation_prompt=add_generation_prompt)

    def __call__(self, text=None, images=None, padding=True, return_tensors="pt"):
        from transformers.feature_extraction_utils import BatchFeature
        img = self.image_processor(images=images, return_tensors=return_tensors)
        grid = img["image_grid_thw"]
        merge = self.image_processor.merge_size ** 2
        k, out_text = 0, []
        for t in text:
            parts = t.split(self.image_token)
            row = parts[0]
            for rest in parts[1:]:
                row += " ".join([self.image_token] * (int(grid[k].prod()) // merge)) + rest
                k += 1
            out_text.append(row)
        enc = dict(self.tokenizer(out_text, padding=padding, return_tensors=return_tensors))
        # transformers 5.x: the model derives its M-RoPE positions from per-token modality ids (0 text, 1 image) that
        # the processor returns beside input_ids (ProcessorMixin.create_mm_token_type_ids)
        enc["mm_token_type_ids"] = (enc["input_ids"] == self.image_token_id).to(enc["input_ids"].dtype)
        return BatchFeature({**enc, **img})


def qwen2vl_processor(min_pixels=None, max_pixels=None):
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    words = _vocabulary(QWEN_SPECIAL)
    fast = _word_level_tokenizer(words, pad_token="<pad>", eos_token="<|im_end|>",
                                 additional_special_tokens=["<|image_pad|>", "<|video_pad|>", "<|vision_start|>", "<|vision_end|>"])
    ip = Qwen2VLImageProcessorPil(patch_size=14, merge_size=2, temporal_patch_size=2, min_pixels=min_pixels or QWEN_MIN_PIXELS,
                                  max_pixels=max_pixels or QWEN_MAX_PIXELS, image_mean=list(CLIP_MEAN), image_std=list(CLIP_STD))
    return ToyQwen2VLProcessor(ip, fast, QWEN_TEMPLATE), len(words)


def qwen2vl_model(vocab_words, seed=0):
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    torch.manual_seed(seed)
    ids = {w: i for i, w in enumerate(QWEN_SPECIAL)}
    cfg = Qwen2VLConfig(
        vision_config=dict(depth=2, embed_dim=32, hidden_size=64, mlp_ratio=2, num_heads=2, in_channels=3, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2),
        text_config=dict(vocab_size=vocab_words, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                         num_key_value_heads=2, max_position_embeddings=512, pad_token_id=ids["<pad>"], bos_token_id=None,
                         eos_token_id=ids["<|im_end|>"],
                         rope_parameters={"rope_type": "default", "mrope_section": [2, 3, 3], "rope_theta": 10000.0}),
        image_token_id=ids["<|image_pad|>"], video_token_id=ids["<|video_pad|>"], vision_start_token_id=ids["<|vision_start|>"],
        vision_end_token_id=ids["<|vision_end|>"])
    model = Qwen2VLForConditionalGeneration(cfg).eval()
    model.requires_grad_(False)
    return model


# ---------------------------------------------------------------------------------------------- plugin entry point
def load_model_and_processor(model_name: str, device, seed: int = 0, dtype=torch.float32):
    if model_name == "synthetic/tiny-mllama":
        proc, n = mllama_processor()
        model = mllama_model(n, seed=seed)
    elif model_name == "synthetic/tiny-qwen2vl":
        proc, n = qwen2vl_processor()
        model = qwen2vl_model(n, seed=seed)
    elif model_name == "synthetic/mllama-11b":
        proc, _ = mllama_processor(tile=560, max_tiles=4)
        return mllama_11b(device, seed=seed), proc
    elif model_name == "synthetic/qwen2-vl-7b":
        proc, _ = qwen2vl_processor(min_pixels=56 * 56, max_pixels=28 * 28 * 1280)
        return qwen2vl_7b(device, seed=seed), proc
    else:
        raise ValueError(model_name)
    return model.to(dtype).to(device), proc
