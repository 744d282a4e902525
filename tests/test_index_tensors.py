"""CPU tier: the integer side tensors ("tile/index tensors bit-exact", north star).

1. Plan geometry and the plugins' `index_tensors()` against tests/golden/index_tensors.npz = outputs of the HF image
   processors themselves (MllamaImageProcessorPil, Qwen2VLImageProcessorPil; tests/golden/make_golden.py).
2. `get_inputs_train()` of AdvMllamaInputs - cached tokenisation + hand-assembled padding + index tensors from the
   plan - against ONE real MllamaProcessor call on the same prompts (padding=True), key by key, for both padding
   sides.  The processor is built offline: HF image processor with explicit parameters + a toy word-level tokenizer.
   (Qwen2VLProcessor cannot be constructed here - its video processor needs torchvision - and Phi-3.5's processor is
   remote code; their only side tensors, image_grid_thw and image_sizes, are covered by part 1 and by the
   reference captures of tests/test_gpu_processors.py.)
"""
import os
import random
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from PIL import Image

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "index_tensors.npz"))
CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


def _mllama_proc(tile, max_tiles):
    from adversarialvlm_amd.processors.llama32processor import DifferentiableMllamaImageProcessor
    orig = SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, size={"height": tile, "width": tile},
                           max_image_tiles=max_tiles, do_convert_rgb=True)
    return DifferentiableMllamaImageProcessor(orig, "cpu")


def test_golden_covers_enough_geometries():
    assert len(GOLD["mllama_geometry"]) >= 10 and len(GOLD["qwen_geometry"]) >= 10


@pytest.mark.parametrize("row", range(len(GOLD["mllama_geometry"])))
def test_mllama_index_tensors_equal_hf(row):
    H, W, tile, max_tiles = (int(v) for v in GOLD["mllama_geometry"][row])
    proc = _mllama_proc(tile, max_tiles)
    info = proc.plan_for(H, W).info
    want_mask = [int(v) for v in GOLD["mllama_aspect_ratio_mask"][row][:max_tiles]]
    assert int(info.aspect_ratio_id) == int(GOLD["mllama_aspect_ratio_ids"][row])
    assert int(info.num_tiles) == int(GOLD["mllama_num_tiles"][row])
    own = proc.index_tensors(H, W, 3)
    assert own["aspect_ratio_ids"].dtype == torch.long and own["aspect_ratio_ids"].tolist() == [[int(GOLD["mllama_aspect_ratio_ids"][row])]] * 3
    assert own["aspect_ratio_mask"].tolist() == [[want_mask]] * 3
    assert own["num_tiles"] == [[int(GOLD["mllama_num_tiles"][row])]] * 3
    assert tuple(int(v) for v in (info.out_shape[2], info.out_shape[4], info.out_shape[5])) == (max_tiles, tile, tile)


@pytest.mark.parametrize("row", range(len(GOLD["qwen_geometry"])))
def test_qwen_grid_equals_hf(row):
    from adversarialvlm_amd.processors.qwen2VLprocessor import DifferentiableQwen2VLImageProcessor
    H, W, lo, hi = (int(v) for v in GOLD["qwen_geometry"][row])
    orig = SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, patch_size=14, merge_size=2, temporal_patch_size=2,
                           min_pixels=lo, max_pixels=hi, do_convert_rgb=True)
    proc = DifferentiableQwen2VLImageProcessor(orig, "cpu")
    t, gh, gw, rows, cols = (int(v) for v in GOLD["qwen_grid_thw_rows_cols"][row])
    plan = proc.plan_for(H, W)
    assert (int(plan.info.grid_h), int(plan.info.grid_w)) == (gh, gw)
    assert plan.out_shape == (rows, cols)
    assert proc.index_tensors(H, W, 2)["image_grid_thw"].tolist() == [[t, gh, gw]] * 2


def test_phi3_image_sizes_are_the_hd_canvas():
    from adversarialvlm_amd.processors.phi3processor import DifferentiablePhi3VImageProcessor
    orig = SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, num_crops=6, do_convert_rgb=True)
    proc = DifferentiablePhi3VImageProcessor(orig, "cpu")
    own = proc.index_tensors(512, 512, 4)["image_sizes"]
    assert own.dtype == torch.long and own.tolist() == [[672, 672]] * 4          # 512^2 -> 672^2 HD canvas


# ---------------------------------------------------------------- get_inputs_train against one HF processor call
WORDS = ["<pad>", "<|begin_of_text|>", "<|eot_id|>", "<|image|>", "<|start_header_id|>", "<|end_header_id|>", "user", "assistant",
         "what", "is", "in", "this", "image", "sure", "here", "it", "describe", "the", "scene", "please", "hi", "now", "<unk>"]
TEMPLATE = ("{% for m in messages %}<|start_header_id|> {{ m['role'] }} <|end_header_id|> "
            "{% for c in m['content'] %}{% if c['type'] == 'image' %}<|image|> {% else %}{{ c['text'] }} {% endif %}{% endfor %}"
            "<|eot_id|> {% endfor %}{% if add_generation_prompt %}<|start_header_id|> assistant <|end_header_id|> {% endif %}")


def _toy_mllama_processor(tile, padding_side):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    from transformers.models.mllama.image_processing_pil_mllama import MllamaImageProcessorPil
    from transformers.models.mllama.processing_mllama import MllamaProcessor
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(WORDS)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<pad>", bos_token="<|begin_of_text|>", eos_token="<|eot_id|>",
                                   unk_token="<unk>", padding_side=padding_side, additional_special_tokens=["<|image|>"])
    ip = MllamaImageProcessorPil(size={"height": tile, "width": tile}, max_image_tiles=4, image_mean=CLIP_MEAN, image_std=CLIP_STD)
    return MllamaProcessor(image_processor=ip, tokenizer=fast, chat_template=TEMPLATE)


@pytest.mark.parametrize("padding_side", ["left", "right"])
@pytest.mark.parametrize("size", [(60, 90), (130, 40), (56, 56)])
def test_mllama_batches_equal_one_hf_processor_call(padding_side, size):
    from adversarialvlm_amd.processors.llama32processor import AdvMllamaInputs, DifferentiableMllamaImageProcessor
    tile = 56
    proc = _toy_mllama_processor(tile, padding_side)
    H, W = size
    image = Image.fromarray((np.random.default_rng(0).random((H, W, 3)) * 255).astype(np.uint8))
    questions = ["what is in this image", "describe the scene please", "hi", "describe this image now please"]
    adv = DifferentiableMllamaImageProcessor(proc.image_processor, "cpu")
    ip = AdvMllamaInputs(questions=questions, test_questions=["hi"], batch_size=5, original_image=image, processor=proc,
                         device="cpu", target_text="sure here it is", rng=random.Random(4))
    ip.bind_geometry(adv, H, W)
    for _ in range(3):                     # repeated calls hit the cache
        state = ip.rng.getstate()
        got = ip.get_inputs_train()
        ip.rng.setstate(state)
        batch_questions = ip.rng.choices(questions, k=5)          # what get_inputs_train just drew
        prompts = [ip._render_train(q, ip.target_text) for q in batch_questions]
        want = proc(text=prompts, images=[[image] for _ in prompts], padding=True, return_tensors="pt")
        assert set(got.keys()) == set(want.keys()) - {"pixel_values"}
        for k in got.keys():
            assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k
    assert got["cross_attention_mask"].shape[-1] == 4 and int(got["attention_mask"].min()) == 0      # rows really are padded
