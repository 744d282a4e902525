"""Qwen2-VL plugin (reference: src/processors/qwen2VLprocessor.py)."""
import torch

from ..plan import Plan
from .base import AdvInputsBase, DifferentiableProcessorBase, chat_template_render


def load_model_and_processor(model_name, device):
    """qwen2VLprocessor.py:16-25 (bf16, left padding)."""
    from transformers import AutoProcessor, Qwen2VLForConditionalGeneration
    model = Qwen2VLForConditionalGeneration.from_pretrained(model_name, device_map=device, torch_dtype=torch.bfloat16)
    model.requires_grad_(False)
    processor = AutoProcessor.from_pretrained(model_name, padding_side="left")
    return model, processor


class AdvQwen2VLInputs(AdvInputsBase):
    refuses = ["As an AI language model, I cannot", "I'm sorry, but"]
    extra_token = "<|im_end|>\n"

    def _render_train(self, question, answer):
        return chat_template_render(self.processor, question, answer, image_first=True)

    def _render_inference(self, question):
        return chat_template_render(self.processor, question, None, image_first=True)


class DifferentiableQwen2VLImageProcessor(DifferentiableProcessorBase):
    """Smart resize to multiples of patch*merge, normalise, temporal duplication, 9-D patchify
    (qwen2VLprocessor.py:211-272)."""

    def __init__(self, orig_processor, device):
        super().__init__(orig_processor, device)
        self.patch_size = orig_processor.patch_size
        self.merge_size = orig_processor.merge_size
        # transformers 4.51 (the reference's pin) keeps min_pixels / max_pixels as attributes (qwen2VLprocessor.py:141-142);
        # 5.x moved them into size = {"shortest_edge": min_pixels, "longest_edge": max_pixels}
        size = getattr(orig_processor, "size", None) or {}
        self.min_pixels = getattr(orig_processor, "min_pixels", None) or size["shortest_edge"]
        self.max_pixels = getattr(orig_processor, "max_pixels", None) or size["longest_edge"]
        self.temporal_patch_size = orig_processor.temporal_patch_size

    def _make_plan(self, H, W):
        return Plan.qwen2vl(H, W, patch=self.patch_size, merge=self.merge_size, temporal=self.temporal_patch_size,
                            min_pixels=self.min_pixels, max_pixels=self.max_pixels, mean=self.mean, std=self.std)

    def _extras(self, plan):
        return {"num_tiles": [int(plan.info.num_tiles)]}

    def index_tensors(self, H, W, batch):
        """image_grid_thw [B,3] the HF processor would produce for this geometry."""
        i = self.plan_for(H, W).info
        return {"image_grid_thw": torch.tensor([[1, int(i.grid_h), int(i.grid_w)]] * batch, dtype=torch.long)}
