"""CPU oracle for the pixel-space hot path (TEST INFRASTRUCTURE ONLY).

This package restates, with stock torch-CPU / numpy operations, the arithmetic of the
reference's universal adversarial-image optimisation loop
(`src/attack_model.py:276-472`, `src/crossattack_models.py:301-517`) and of its four
differentiable image processors (`src/processors/*.py`).  It exists to CHECK the HIP
path; nothing under `adversarialvlm_amd/` may import it.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it.

Pinning status (see DESIGN.md "Oracle"):
  * LLaVA / Qwen2-VL / Phi-3.5 processors: pinned by captures from the imported
    reference classes (tests/golden/*.npz, generator tests/golden/make_golden.py).
  * Llama-3.2 (mllama) processor: pinned by captures from the imported reference class
    (mllama_reference.npz; the three integer helpers it asks transformers for are served
    from `image_processing_pil_mllama`, see make_golden.py); integer geometry additionally
    pinned by those helpers.
  * image_fit_loss / create_mask / get_loss: pinned by outputs of the reference's own
    functions (trainer_helpers.npz, suffix_loss.npz).
  * GaussianBlur / RandomResizedCrop: torchvision is absent -> restated from the
    published algorithm, "torchvision-unverified" (parity unpinned for these two ops).
  * In-kernel noise: `philox.py` restates Philox4x32-10 and the device's counter addressing,
    pinned by the Random123 known-answer vectors (tests/test_oracle_philox.py).
  * The reference has no tests or golden vectors of its own.
"""
