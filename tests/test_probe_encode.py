"""CPU tier: the encode step of the batched generation probe (train_test._generate_batched -> ip._encode) for the
plugins whose HF processor takes images per SAMPLE.  ADVICE r02 (high): a flat image list made `MllamaProcessor`
raise for more than one prompt, so every run with --generation_probe and a Llama-3.2 model died at its first save
step.  Also the Qwen2-VL batch against one processor call (the processor is the toy joining of
the real image processor and a real fast tokenizer, since `Qwen2VLProcessor` does not construct without
torchvision - see DESIGN.md section 2)."""
import random

import numpy as np
import pytest
import torch
from PIL import Image

QUESTIONS = ["what is in this image", "describe the scene please", "hi", "describe this image now please"]


def _image(H, W):
    return Image.fromarray((np.random.default_rng(0).random((H, W, 3)) * 255).astype(np.uint8))


def test_mllama_probe_encodes_several_prompts():
    from adversarialvlm_amd.testing.synthetic_vlms import AdvMllamaInputs, mllama_processor
    proc, _ = mllama_processor()
    img = _image(60, 90)
    ip = AdvMllamaInputs(questions=QUESTIONS, test_questions=QUESTIONS, batch_size=2, original_image=img, processor=proc,
                         device="cpu", target_text="sure here it is")
    prompts = [ip._render_inference(q) for q in QUESTIONS[:3]]
    with pytest.raises(ValueError):                      # what the probe used to do
        proc(text=prompts, images=[img] * 3, padding=True, return_tensors="pt")
    enc = ip._encode(prompts, [img] * 3)
    assert enc["input_ids"].shape[0] == 3 and enc["pixel_values"].shape[:3] == (3, 1, 4)
    assert enc["cross_attention_mask"].shape[:2] == enc["input_ids"].shape
    assert int(enc["attention_mask"].min()) == 0          # rows of different length: left padding is really exercised
    one = ip._encode(prompts[:1], [img])                 # the serial probe and get_inputs_train use the same method
    assert torch.equal(one["aspect_ratio_ids"], enc["aspect_ratio_ids"][:1])


@pytest.mark.parametrize("size", [(60, 90), (130, 40), (56, 56)])
def test_qwen2vl_batches_equal_one_processor_call(size):
    """get_inputs_train() (cached tokenisation, hand-assembled left padding, image_grid_thw from the PLAN) against
    ONE call of the processor on the same prompts, key by key - qwen2VLprocessor.py:68-96 keeps whatever that call
    returns.  `mm_token_type_ids` (transformers 5.x) follows input_ids' padding."""
    from adversarialvlm_amd.testing.synthetic_vlms import (AdvQwen2VLInputs, DifferentiableQwen2VLImageProcessor,
                                                              qwen2vl_processor)
    proc, _ = qwen2vl_processor()
    H, W = size
    img = _image(H, W)
    adv = DifferentiableQwen2VLImageProcessor(proc.image_processor, "cpu")
    ip = AdvQwen2VLInputs(questions=QUESTIONS, test_questions=["hi"], batch_size=5, original_image=img, processor=proc,
                          device="cpu", target_text="sure here it is", rng=random.Random(4))
    ip.bind_geometry(adv, H, W)
    for _ in range(3):
        state = ip.rng.getstate()
        got = ip.get_inputs_train()
        ip.rng.setstate(state)
        drawn = ip.rng.choices(QUESTIONS, k=5)
        want = proc(text=[ip._render_train(q, ip.target_text) for q in drawn], images=[img] * 5)
        assert set(got.keys()) == set(want.keys()) - {"pixel_values"}
        for k in got.keys():
            assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k
    assert int(got["attention_mask"].min()) == 0 and int(got["mm_token_type_ids"].max()) == 1
    n_img = int((got["input_ids"][0] == proc.image_token_id).sum())
    t, gh, gw = got["image_grid_thw"][0].tolist()
    assert n_img == t * gh * gw // 4                      # one placeholder per merged 2x2 patch group


@pytest.mark.parametrize("size", [(60, 90), (130, 40), (56, 56), (336, 336), (100, 400)])
def test_phi3v_twin_batches_equal_the_reference_assembly(size):
    """get_inputs_train() of the Phi-3.5 plugin (cached tokenisation, image_sizes from the PLAN) against what the
    reference assembles per step - one processor call per prompt, pad_left, cat (phi3processor.py:275-311) - on the twin of
    the remote processor (testing/synthetic_phi3v.py): key by key.  The number of negative placeholder ids equals the
    plan's num_img_tokens, i.e. the HD geometry of the HIP path (C side) and of the twin's PIL path agree."""
    from adversarialvlm_amd.testing.synthetic_phi3v import (AdvPhiInputs, DifferentiablePhi3VImageProcessor, phi3v_processor)
    from adversarialvlm_amd.processors.phi3processor import pad_left
    proc, _ = phi3v_processor()
    H, W = size
    img = _image(H, W)
    adv = DifferentiablePhi3VImageProcessor(proc.image_processor, "cpu")
    ip = AdvPhiInputs(questions=QUESTIONS, test_questions=["hi"], batch_size=4, original_image=img, processor=proc,
                      device="cpu", target_text="sure here it is", rng=random.Random(4))
    assert ip.shift == 1                                   # BOS is the "extra" token of phi3processor.py:61
    ip.bind_geometry(adv, H, W)
    for _ in range(2):
        state = ip.rng.getstate()
        got = ip.get_inputs_train()
        ip.rng.setstate(state)
        drawn = ip.rng.choices(QUESTIONS, k=4)
        encs = [proc(ip._render_train(q, ip.target_text), [img], return_tensors="pt") for q in drawn]
        ids = pad_left([e.input_ids[0] for e in encs], proc.tokenizer.pad_token_id)
        want = {"input_ids": ids, "attention_mask": (ids != proc.tokenizer.pad_token_id).long(),
                "image_sizes": torch.cat([e.image_sizes for e in encs], 0)}
        assert set(got.keys()) == set(want.keys())
        for k in want:
            assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k
    info = adv.plan_for(H, W).info
    assert int((got["input_ids"][0] < 0).sum()) == int(info.num_img_tokens)
    assert tuple(encs[0].pixel_values.shape[1:]) == tuple(adv.plan_for(H, W).out_shape[1:])
