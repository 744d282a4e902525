"""Test support that needs the product package but is not part of it: random-init model architectures with toy / word-level
tokenizers standing where the reference's `from_pretrained` checkpoints stand (there are no weights, tokenizers or
preprocessor configs offline).  Used by tests/, tests/golden/make_golden.py, tools/e2e_bench.py and bench.py's end-to-end child.

Importing this package registers the `synthetic/*` model names with the plugin registry (`processors.register`), exactly as
a user would register a local checkpoint.  Processes that reach the registry by NAME only (the trainers run as commands)
get it through `ADVX_PLUGIN_MODULES=adversarialvlm_amd.testing` (processors/__init__.py: load_components).

    synthetic/tiny-llava      CLIP-like 56x56/14 tower + 2-layer Llama (CPU-runnable)           testing/synthetic.py
    synthetic/llava-1.5-7b    LlavaConfig() defaults = CLIP-L/14-336 + Llama-7B                  testing/synthetic.py
    synthetic/tiny-mllama, synthetic/mllama-11b, synthetic/tiny-qwen2vl, synthetic/qwen2-vl-7b   testing/synthetic_vlms.py
    synthetic/tiny-phi3v      twin of the Phi-3.5-Vision remote-code INTERFACE                   testing/synthetic_phi3v.py
"""
from ..processors import register

_PKG = __name__

SYNTHETIC_MODELS = {
    "synthetic/tiny-llava": ("synthetic", "AdvLlavaInputs", "DifferentiableLlavaImageProcessor"),
    "synthetic/llava-1.5-7b": ("synthetic", "AdvLlavaInputs", "DifferentiableLlavaImageProcessor"),
    "synthetic/tiny-mllama": ("synthetic_vlms", "AdvMllamaInputs", "DifferentiableMllamaImageProcessor"),
    "synthetic/mllama-11b": ("synthetic_vlms", "AdvMllamaInputs", "DifferentiableMllamaImageProcessor"),
    "synthetic/tiny-qwen2vl": ("synthetic_vlms", "AdvQwen2VLInputs", "DifferentiableQwen2VLImageProcessor"),
    "synthetic/qwen2-vl-7b": ("synthetic_vlms", "AdvQwen2VLInputs", "DifferentiableQwen2VLImageProcessor"),
    "synthetic/tiny-phi3v": ("synthetic_phi3v", "AdvPhiInputs", "DifferentiablePhi3VImageProcessor"),
}


def register_synthetic_models():
    for name, (module, input_class, processor_class) in SYNTHETIC_MODELS.items():
        register(name, f"{_PKG}.{module}", input_class, processor_class)


register_synthetic_models()
