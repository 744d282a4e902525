"""Plugin registry: model name -> (load_model_and_processor, AdvInputs, DifferentiableImageProcessor).

Same map and the same `load_components` contract as the reference's
`src/processors/__init__.py:5-76` (ValueError for unknown names; a `processor_class` of None is
accepted and yields None, as for the reference's evaluation-only judge entry `google/gemma-3-12b-it`
- that entry itself is NOT registered here: the judge and the evaluation harness are out of scope,
SURVEY.md section 2 rows 9 and 13-14), resolved inside this package.  The random-init `synthetic/*` architectures the tests and the end-to-end benchmark use are
NOT part of this map: they live in `adversarialvlm_amd.testing` and register themselves when that package is imported
(directly, or by name through ADVX_PLUGIN_MODULES).
"""
import importlib
import os
from typing import Tuple

MODEL_MAP = {
    "microsoft/Phi-3.5-vision-instruct": {
        "module": "adversarialvlm_amd.processors.phi3processor",
        "input_class": "AdvPhiInputs",
        "processor_class": "DifferentiablePhi3VImageProcessor",
    },
    "Qwen/Qwen2-VL-2B-Instruct": {
        "module": "adversarialvlm_amd.processors.qwen2VLprocessor",
        "input_class": "AdvQwen2VLInputs",
        "processor_class": "DifferentiableQwen2VLImageProcessor",
    },
    "Qwen/Qwen2-VL-7B-Instruct": {
        "module": "adversarialvlm_amd.processors.qwen2VLprocessor",
        "input_class": "AdvQwen2VLInputs",
        "processor_class": "DifferentiableQwen2VLImageProcessor",
    },
    "alpindale/Llama-3.2-11B-Vision-Instruct": {
        "module": "adversarialvlm_amd.processors.llama32processor",
        "input_class": "AdvMllamaInputs",
        "processor_class": "DifferentiableMllamaImageProcessor",
    },
    "alpindale/Llama-3.2-11B-Vision": {
        "module": "adversarialvlm_amd.processors.llama32processor",
        "input_class": "AdvMllamaInputs",
        "processor_class": "DifferentiableMllamaImageProcessor",
    },
    "SinclairSchneider/Llama-Guard-3-11B-Vision": {
        "module": "adversarialvlm_amd.processors.llama32processor",
        "input_class": "AdvMllamaInputs",
        "processor_class": "DifferentiableMllamaImageProcessor",
    },
    "llava-hf/llava-1.5-7b-hf": {
        "module": "adversarialvlm_amd.processors.llavaprocessor",
        "input_class": "AdvLlavaInputs",
        "processor_class": "DifferentiableLlavaImageProcessor",
    },
}


def register(model_name: str, module: str, input_class: str, processor_class):
    """Add a model (e.g. a local checkpoint path or a tiny test config) to the map."""
    MODEL_MAP[model_name] = {"module": module, "input_class": input_class, "processor_class": processor_class}


def _import_plugin_modules():
    """ADVX_PLUGIN_MODULES = comma-separated module names; importing one registers its models (`register`).  How a process
    that is only handed a model NAME - a trainer run as a command - learns of models outside the map above, e.g. the random-init
    architectures of adversarialvlm_amd.testing that the tests and the end-to-end benchmark run around."""
    for name in filter(None, (m.strip() for m in os.environ.get("ADVX_PLUGIN_MODULES", "").split(","))):
        importlib.import_module(name)


def load_components(model_name: str) -> Tuple[object, object, object]:
    if model_name not in MODEL_MAP:
        _import_plugin_modules()
    if model_name not in MODEL_MAP:
        raise ValueError(f"Model {model_name} not found in MODEL_MAP. Please add it to the map.")
    info = MODEL_MAP[model_name]
    module = importlib.import_module(info["module"])
    load_model_and_processor = getattr(module, "load_model_and_processor")
    adv_inputs = getattr(module, info["input_class"])
    proc = getattr(module, info["processor_class"]) if info["processor_class"] is not None else None
    return load_model_and_processor, adv_inputs, proc
