"""Diagnostic (development): run the dense / sparse trainings of test_accumulated_loss_never_mixes_accumulation_windows a few
times and print where two runs of the same training differ."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ADVX_PLUGIN_MODULES", "adversarialvlm_amd.testing")
import torch
import adversarialvlm_amd.testing  # noqa
from adversarialvlm_amd import attack_model
from test_gpu_e2e import _kw

tmp = tempfile.mkdtemp()
runs = []
for r in range(4):
    _, h = attack_model.train(**_kw(tmp, f"dense{r}", 8, grad_accum_steps=2, return_engine=True))
    runs.append(h)
    print("run", r, [(x["iteration"], repr(x["ce_loss"]), repr(x["image_loss"])) for x in h[:5]], flush=True)
keys = [k for k in runs[0][0] if isinstance(runs[0][0][k], (int, float))]
for r in range(1, 4):
    for it in range(8):
        for k in keys:
            if runs[r][it].get(k) != runs[0][it].get(k):
                print("DIFF run", r, "iteration", it, k, repr(runs[0][it].get(k)), repr(runs[r][it].get(k)))
eng = attack_model.train(**_kw(tmp, "dense_e", 8, grad_accum_steps=2, return_engine=True))[0]
print("mode", eng.mode)

# ---- is the VLM's forward itself reproducible?  (same inputs, same weights, 200 calls)
from adversarialvlm_amd.testing.synthetic import load_model_and_processor
from adversarialvlm_amd.processors import load_components
import random
dev = torch.device("cuda:0")
model, proc = load_model_and_processor("synthetic/tiny-llava", dev, seed=0)
_, AdvInputs, _ = load_components("synthetic/tiny-llava")
ip = AdvInputs(questions=["what is in the image", "describe the scene please", "hi"], test_questions=["t"], batch_size=4,
               original_image=None, processor=proc, device=dev, target_text="sure here it is", rng=random.Random(5))
inputs = ip.get_inputs_train()
pv = torch.rand(4, 3, 56, 56, device=dev)
vals = {}
for i in range(300):
    out = model(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"], pixel_values=pv)
    l = ip.get_loss(out.logits[:, :-1, :])
    key = (repr(float(l)), repr(float(out.logits.double().sum())))
    vals[key] = vals.get(key, 0) + 1
print("forward of the tiny VLM on identical inputs, 300 calls ->", vals)
