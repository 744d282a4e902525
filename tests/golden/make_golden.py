#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

The reference ships no tests or golden vectors, so the oracle is
pinned against outputs of the reference's own classes, captured here by importing them
from /root/reference/src, and against the stock torch calls the reference trainer
makes.  Only DATA (inputs + expected outputs) is written; no reference source text is
stored.  The reference tree does not exist on the GPU box - nothing at test time reads it.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz (+ cli_flags_reference.json)
    GOLDEN_OUT=/tmp/g python tests/golden/make_golden.py     # ... into another directory, to compare with the committed ones

Importability notes: llavaprocessor imports directly; qwen2VLprocessor /
phi3processor have unused torchvision imports, so a bare stub module is registered after
transformers' symbols are resolved; llama32processor additionally imports three integer
helpers from a transformers module that needs torchvision - the installed transformers ships
the same helpers in its PIL-backend module, which is registered under the expected name
(import_reference_mllama).  mllama_restated.npz (oracle output) is kept as a regression file.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("GOLDEN_OUT", HERE)        # GOLDEN_OUT=/tmp/x regenerates the fixtures somewhere else (to compare with the tree's)


def repo_root():
    """Where `adversarialvlm_amd/` lives (the trainer-run fixtures borrow its tiny random models): beside this file's tests/
    directory, else the working directory, else /root/repo - so a copy of this script run from elsewhere still finds it."""
    for cand in (os.path.dirname(os.path.dirname(HERE)), os.getcwd(), "/root/repo"):
        if os.path.isdir(os.path.join(cand, "adversarialvlm_amd")):
            return cand
    raise SystemExit("adversarialvlm_amd/ not found: run from the repository root")
REF = "/root/reference/src"

CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


def lcg_tensor(shape, salt):
    """Deterministic pseudo-random float32 in [-0.5, 0.5): exact integer arithmetic, so
    the tests can rebuild large upstream gradients instead of storing them."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    v = (i * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503)) % np.uint64(2 ** 32)
    v = (v * np.uint64(1664525) + np.uint64(1013904223)) % np.uint64(2 ** 32)
    return torch.from_numpy((v.astype(np.float64) / 2 ** 32 - 0.5).astype(np.float32).reshape(shape))


def meta():
    import transformers
    return dict(torch_version=torch.__version__, transformers_version=transformers.__version__,
                numpy_version=np.__version__)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    for k, v in meta().items():
        out["meta_" + k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print("wrote", name, {k: out[k].shape for k in out if not k.startswith("meta_")})


def import_reference():
    sys.path.insert(0, REF)
    import processors.llavaprocessor as llava
    # resolve what the other two modules import from transformers BEFORE the stub exists
    from transformers import AutoProcessor, Qwen2VLForConditionalGeneration, AutoModelForCausalLM  # noqa: F401
    from transformers.image_processing_utils import BatchFeature  # noqa: F401
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tv.transforms, tvt.functional = tvt, tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf})
    import processors.qwen2VLprocessor as qwen
    import processors.phi3processor as phi3
    return llava, qwen, phi3


def import_reference_mllama():
    """llama32processor imports three integer helpers from
    transformers.models.mllama.image_processing_mllama, a module that itself needs torchvision
    and therefore cannot be imported here; the SAME helpers (get_optimal_tiled_canvas,
    get_image_size_fit_to_canvas, pack_images) ship in the installed transformers under
    image_processing_pil_mllama.  Registering that installed module under the name the reference
    asks for (plus the bare torchvision stub: the reference imports torchvision and never calls
    it) makes `DifferentiableMllamaImageProcessor` importable.  Call after import_reference()."""
    import importlib
    from transformers import AutoProcessor, MllamaConfig, MllamaForConditionalGeneration, MllamaImageProcessor  # noqa: F401
    pil = importlib.import_module("transformers.models.mllama.image_processing_pil_mllama")
    sys.modules.setdefault("transformers.models.mllama.image_processing_mllama", pil)
    return importlib.import_module("processors.llama32processor")


def golden_llava(llava):
    cases = [("down", (3, 37, 91), (24, 32)), ("mixed", (3, 64, 40), (56, 56)),
             ("ident", (3, 48, 48), (48, 48)), ("up", (3, 20, 30), (50, 44))]
    arrays = {}
    for k, (name, ishape, (ch, cw)) in enumerate(cases):
        op = SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, do_convert_rgb=True,
                             crop_size={"height": ch, "width": cw})
        proc = llava.DifferentiableLlavaImageProcessor(op, "cpu")
        torch.manual_seed(100 + k)
        img = torch.rand(ishape, requires_grad=True)
        pv = proc.process(img)["pixel_values"]
        up = lcg_tensor(pv.shape, 7 + k)
        pv.backward(up)
        arrays.update({f"{name}_image": img, f"{name}_crop": np.array([ch, cw]),
                       f"{name}_pixel_values": pv, f"{name}_image_grad": img.grad,
                       f"{name}_salt": np.array(7 + k)})
    # full-size capture 512 -> 336 : checksums + sampled entries only
    op = SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, do_convert_rgb=True,
                         crop_size={"height": 336, "width": 336})
    proc = llava.DifferentiableLlavaImageProcessor(op, "cpu")
    img = lcg_tensor((3, 512, 512), 99) + 0.5
    img.requires_grad_(True)
    pv = proc.process(img)["pixel_values"]
    up = lcg_tensor(pv.shape, 98)
    pv.backward(up)
    idx = np.arange(0, pv.numel(), 997)
    gidx = np.arange(0, img.numel(), 1499)
    arrays.update(full_salt_image=np.array(99), full_salt_up=np.array(98),
                  full_pv_sum=pv.detach().double().sum(), full_pv_sumsq=(pv.detach().double() ** 2).sum(),
                  full_pv_idx=idx, full_pv_val=pv.detach().flatten()[idx],
                  full_grad_sum=img.grad.double().sum(), full_grad_idx=gidx,
                  full_grad_val=img.grad.flatten()[gidx])
    save("llava_reference.npz", **arrays)


def golden_qwen(qwen):
    arrays = {}
    cases = [("a", (3, 60, 90), 28 * 28 * 4, 28 * 28 * 64), ("b", (3, 50, 50), 56 * 56, 28 * 28 * 6),
             ("c", (3, 20, 33), 56 * 56, 28 * 28 * 1280)]
    for k, (name, ishape, minp, maxp) in enumerate(cases):
        op = SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, do_convert_rgb=True,
                             patch_size=14, merge_size=2, temporal_patch_size=2,
                             min_pixels=minp, max_pixels=maxp)
        proc = qwen.DifferentiableQwen2VLImageProcessor(op, "cpu")
        torch.manual_seed(200 + k)
        img = torch.rand(ishape, requires_grad=True)
        out = proc.process(img)
        pv = out["pixel_values"]
        up = lcg_tensor(pv.shape, 17 + k)
        pv.backward(up)
        arrays.update({f"{name}_image": img, f"{name}_minmax": np.array([minp, maxp]),
                       f"{name}_pixel_values": pv, f"{name}_num_tiles": np.array(out["num_tiles"]),
                       f"{name}_image_grad": img.grad, f"{name}_salt": np.array(17 + k),
                       f"{name}_optimal_size": np.array(proc._optimal_size(img))})
    # integer geometry sweep
    op = SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, do_convert_rgb=True, patch_size=14,
                         merge_size=2, temporal_patch_size=2, min_pixels=56 * 56, max_pixels=28 * 28 * 1280)
    proc = qwen.DifferentiableQwen2VLImageProcessor(op, "cpu")
    sizes = [(336, 336), (512, 512), (1352, 1988), (30, 40), (14, 14), (2000, 3000), (377, 1201),
             (42, 42), (70, 70), (1001, 999)]
    geo = [list(s) + list(proc._optimal_size(torch.empty(3, *s))) for s in sizes]
    arrays["geometry"] = np.array(geo, dtype=np.int64)
    save("qwen2vl_reference.npz", **arrays)


def golden_phi3(phi3):
    arrays = {}
    cases = [("wide", (3, 100, 150)), ("tall", (3, 150, 100)), ("square", (3, 120, 120))]
    for k, (name, ishape) in enumerate(cases):
        op = SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, do_convert_rgb=True,
                             num_crops=6, num_img_tokens=144)
        proc = phi3.DifferentiablePhi3VImageProcessor(op, "cpu")
        torch.manual_seed(300 + k)
        img = torch.rand(ishape, requires_grad=True)
        out = proc.process(img)
        pv = out["pixel_values"]
        up = lcg_tensor(pv.shape, 27 + k)
        pv.backward(up)
        flat = pv.detach().reshape(7, -1).double()
        idx = np.arange(0, pv.numel(), 1009)
        arrays.update({f"{name}_image": img, f"{name}_image_sizes": np.array(out["image_sizes"]),
                       f"{name}_num_img_tokens": np.array(out["num_img_tokens"]),
                       f"{name}_tile_sum": flat.sum(1), f"{name}_tile_sumsq": (flat ** 2).sum(1),
                       f"{name}_pv_idx": idx, f"{name}_pv_val": pv.detach().flatten()[idx],
                       f"{name}_image_grad": img.grad, f"{name}_salt": np.array(27 + k)})
    sizes = [(336, 336), (512, 512), (1352, 1988), (1988, 1352), (100, 900), (900, 100), (300, 301)]
    geo = []
    proc = phi3.DifferentiablePhi3VImageProcessor(
        SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, do_convert_rgb=True, num_crops=6,
                        num_img_tokens=144), "cpu")
    for s in sizes:
        out = proc.process(torch.full((3, *s), 0.5))
        geo.append(list(s) + list(out["image_sizes"][0]) + list(out["num_img_tokens"]))
    arrays["geometry"] = np.array(geo, dtype=np.int64)
    save("phi3_reference.npz", **arrays)


def golden_mllama_helpers():
    """Integer geometry from the installed transformers helpers the reference calls
    (llama32processor.py:262-277).  Data only."""
    import importlib
    m = importlib.import_module("transformers.models.mllama.image_processing_pil_mllama")
    sizes = [(336, 336), (512, 512), (1352, 1988), (1988, 1352), (100, 900), (900, 100), (560, 560),
             (561, 560), (1120, 1120), (2000, 500), (300, 2400), (10, 10), (1121, 1119), (700, 700)]
    rows = []
    for (h, w) in sizes:
        for (mt, ts) in [(4, 560), (4, 448), (6, 224)]:
            ch, cw = m.get_optimal_tiled_canvas(h, w, mt, ts)
            nh, nw = m.get_image_size_fit_to_canvas(h, w, ch, cw, ts)
            rows.append([h, w, mt, ts, int(ch), int(cw), int(nh), int(nw)])
    ids = m.convert_aspect_ratios_to_ids_np([[(1, 1)], [(2, 2)], [(1, 4)], [(4, 1)], [(2, 1)]], 4)
    save("mllama_helpers.npz", geometry=np.array(rows, dtype=np.int64), aspect_ids=ids,
         arrangements4=np.array(m.get_all_supported_aspect_ratios(4), dtype=np.int64))


def golden_closed_form():
    """Fixtures from the stock torch calls the trainer makes (attack_model.py:94-104, 184,
    216, 300, 340, 343-346, 366-373)."""
    torch.manual_seed(7)
    arrays = {}
    # tanh + image_fit_loss value/grad, expression restated from attack_model.py:94-104
    p = (torch.randn(3, 9, 11) * 2).requires_grad_(True)
    x0 = torch.rand(3, 9, 11) * 1.4 - 0.2
    eps = 0.5
    x = eps * torch.tanh(p)
    s = x0 + x
    loss = torch.mean(torch.relu(0.9 * torch.zeros_like(s) - s) ** 2 + torch.relu(s - 0.9 * torch.ones_like(s)) ** 2)
    loss.backward()
    arrays.update(fit_p=p, fit_x0=x0, fit_eps=np.array(eps), fit_x=x, fit_loss=loss, fit_p_grad=p.grad)
    # AdamW + StepLR trajectory
    q = torch.zeros(3, 5, 7, requires_grad=True)
    opt = torch.optim.AdamW([q], lr=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5)
    gs, ps, ms, vs, lrs = [], [], [], [], []
    for t in range(6):
        g = torch.randn(3, 5, 7) * (10.0 ** (t - 3))
        q.grad = g.clone()
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        st = opt.state[q]
        gs.append(g); ps.append(q.detach().clone()); ms.append(st["exp_avg"].clone()); vs.append(st["exp_avg_sq"].clone())
    arrays.update(adamw_g=torch.stack(gs), adamw_p=torch.stack(ps), adamw_m=torch.stack(ms),
                  adamw_v=torch.stack(vs), adamw_lr=np.array(lrs), adamw_lr0=np.array(1e-2),
                  adamw_step_size=np.array(2), adamw_gamma=np.array(0.5))
    # quantiser (tensor2pil -> PNG -> pil_to_tensor == uint8 truncation), incl. lattice points
    lattice = torch.arange(256, dtype=torch.float32) / 255
    sv = torch.cat([lattice, torch.rand(3 * 16 * 16 - 256) * 1.3 - 0.15]).reshape(3, 16, 16)
    qv = torch.tensor((sv.clamp(0, 1) * 255).numpy().astype(np.uint8).astype(np.float32) / 255)
    d = (qv - sv).abs()
    arrays.update(q_s=sv, q_q=qv, q_std=d.std(), q_mean=d.mean(), q_l1=d.sum())
    # grad norm
    gg = torch.randn(3, 8, 8)
    arrays.update(norm_g=gg, norm_val=gg.norm())
    save("closed_form.npz", **arrays)


def import_reference_trainer():
    """attack_model.py imports wandb (absent), two torchvision transform classes and names a
    wandb type in a signature; none of them is touched by the two pure helpers captured here
    (`image_fit_loss` :86-106, `create_mask` :66-84), so empty placeholders with those names are
    enough to import the module.  Call after import_reference() / import_reference_mllama()
    (processors/__init__ pulls every plugin in)."""
    import importlib
    tvt = sys.modules["torchvision.transforms"]
    for name in ("RandomResizedCrop", "GaussianBlur"):
        if not hasattr(tvt, name):
            setattr(tvt, name, type(name, (), {}))
    if "wandb" not in sys.modules:
        w = types.ModuleType("wandb")
        w.Table = type("Table", (), {})
        sys.modules["wandb"] = w
    return importlib.import_module("attack_model")


def golden_trainer_helpers(am):
    """Outputs of the reference trainer's own helper functions (not restatements)."""
    arrays = {}
    for k, (shape, lo, hi) in enumerate([((3, 9, 11), -0.6, 0.6), ((3, 16, 16), -1.0, 1.0), ((3, 5, 7), -0.05, 0.05)]):
        torch.manual_seed(500 + k)
        x0 = torch.rand(shape)
        x = (torch.rand(shape) * (hi - lo) + lo).requires_grad_(True)
        loss = am.image_fit_loss(x0, x, 0, 1)                       # the call attack_model.py:329 makes
        loss.backward()
        arrays.update({f"fit{k}_x0": x0, f"fit{k}_x": x.detach(), f"fit{k}_loss": loss.detach(), f"fit{k}_x_grad": x.grad})
    for k, (mt, ms, shape) in enumerate([("corner", 4, (3, 9, 11)), ("corner", 100, (3, 336, 336)), ("bottom_lines", 3, (3, 9, 11)),
                                         ("bottom_lines", 20, (3, 70, 100))]):
        m = am.create_mask(mt, ms, shape, "cpu")
        arrays.update({f"mask{k}_type": np.array(0 if mt == "corner" else 1), f"mask{k}_size": np.array(ms),
                       f"mask{k}_shape": np.array(shape), f"mask{k}_sum": m.double().sum(),
                       f"mask{k}_idx": torch.nonzero(m.flatten())[:: max(1, int(m.sum()) // 64)].flatten()})
    save("trainer_helpers.npz", **arrays)


def golden_suffix_loss(llava):
    """AdvLlavaInputs.update_target_tokens + get_loss (llavaprocessor.py:64-78) called on an
    instance whose constructor is bypassed (it needs a downloaded HF processor): the tokenizer is
    a two-line fake that returns fixed ids, everything else is the reference's own code.
    Captures the loss and d loss / d logits for [B, S, V] logits."""
    arrays = {}
    for k, (B, S, V, ids, shift) in enumerate([(3, 14, 50, [7, 11, 3, 29, 2], 1), (2, 20, 97, [5, 9, 9, 4, 31, 8, 2], 2),
                                               (4, 9, 33, [1, 2], 1)]):
        inst = object.__new__(llava.AdvLlavaInputs)
        inst.device = "cpu"
        inst.batch_size = B
        inst.shift = shift
        inst.extra_token = ""
        inst.target_text = "t"
        inst.processor = SimpleNamespace(tokenizer=lambda text, return_tensors, add_special_tokens, _ids=ids:
                                         SimpleNamespace(input_ids=torch.tensor([_ids])))
        inst.update_target_tokens()
        torch.manual_seed(600 + k)
        logits = (torch.randn(B, S, V) * 2.0).requires_grad_(True)
        loss = inst.get_loss(logits[:, :-1, :])                      # attack_model.py:325-327
        loss.backward()
        arrays.update({f"ce{k}_logits": logits.detach(), f"ce{k}_ids": np.array(ids), f"ce{k}_shift": np.array(shift),
                       f"ce{k}_target": inst.target, f"ce{k}_suffix_length": np.array(inst.suffix_length),
                       f"ce{k}_loss": loss.detach(), f"ce{k}_logits_grad": logits.grad})
    save("suffix_loss.npz", **arrays)


def golden_mllama_reference(mllama):
    """Captures from the reference's own DifferentiableMllamaImageProcessor
    (llama32processor.py:219-405): pixel_values, num_tiles and image.grad for a seeded upstream
    gradient over geometries that reach 1, 2, 3 and 4 tiles in every arrangement, plus one
    capture at the model's real tile size (560), stored as checksums and sampled entries."""
    arrays = {}
    cases = [("one", (3, 30, 27), 32), ("wide3", (3, 20, 90), 32), ("tall4", (3, 120, 25), 32), ("two_by_two", (3, 70, 100), 32),
             ("wide2", (3, 40, 75), 32), ("tall2", (3, 60, 33), 32), ("up", (3, 10, 12), 32)]
    for k, (name, ishape, tile) in enumerate(cases):
        op = SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, do_convert_rgb=True, do_normalize=True,
                             size={"height": tile, "width": tile}, max_image_tiles=4, rescale_factor=1 / 255)
        proc = mllama.DifferentiableMllamaImageProcessor(op, "cpu")
        torch.manual_seed(300 + k)
        img = torch.rand(ishape, requires_grad=True)
        out = proc.process(img)
        pv = out["pixel_values"]
        up = lcg_tensor(pv.shape, 31 + k)
        pv.backward(up)
        arrays.update({f"{name}_image": img, f"{name}_tile": np.array(tile), f"{name}_pixel_values": pv,
                       f"{name}_image_grad": img.grad, f"{name}_salt": np.array(31 + k),
                       f"{name}_num_tiles": np.array(int(out["num_tiles"]))})
    op = SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, do_convert_rgb=True, do_normalize=True,
                         size={"height": 560, "width": 560}, max_image_tiles=4, rescale_factor=1 / 255)
    proc = mllama.DifferentiableMllamaImageProcessor(op, "cpu")
    img = lcg_tensor((3, 512, 512), 77) + 0.5
    img.requires_grad_(True)
    out = proc.process(img)
    pv = out["pixel_values"]
    up = lcg_tensor(pv.shape, 76)
    pv.backward(up)
    idx = np.arange(0, pv.numel(), 4999)
    gidx = np.arange(0, img.numel(), 1499)
    arrays.update(full_salt_image=np.array(77), full_salt_up=np.array(76), full_num_tiles=np.array(int(out["num_tiles"])),
                  full_shape=np.array(pv.shape), full_pv_sum=pv.detach().double().sum(),
                  full_pv_sumsq=(pv.detach().double() ** 2).sum(), full_pv_idx=idx, full_pv_val=pv.detach().flatten()[idx],
                  full_grad_sum=img.grad.double().sum(), full_grad_idx=gidx, full_grad_val=img.grad.flatten()[gidx])
    save("mllama_reference.npz", **arrays)


def golden_mllama_restated():
    """NOT a reference capture: produced by the oracle's restatement of
    llama32processor.py:360-405 (module not importable here)."""
    sys.path.insert(0, repo_root())
    from oracle.processors import MllamaOracle
    arrays = {}
    for k, (name, ishape, tile) in enumerate([("a", (3, 40, 70), 32), ("b", (3, 90, 50), 32), ("c", (3, 30, 30), 32)]):
        proc = MllamaOracle(tile=tile, max_tiles=4)
        torch.manual_seed(400 + k)
        img = torch.rand(ishape, requires_grad=True)
        out = proc.process(img)
        pv = out["pixel_values"]
        up = lcg_tensor(pv.shape, 37 + k)
        pv.backward(up)
        arrays.update({f"{name}_image": img, f"{name}_tile": np.array(tile), f"{name}_pixel_values": pv,
                       f"{name}_num_tiles": np.array(out["num_tiles"]), f"{name}_image_grad": img.grad,
                       f"{name}_salt": np.array(37 + k)})
    save("mllama_restated.npz", **arrays)


MLLAMA_GEOMETRIES = [  # (H, W, tile, max_image_tiles)
    (336, 336, 560, 4), (512, 512, 560, 4), (300, 1000, 560, 4), (1000, 300, 560, 4), (700, 700, 560, 4), (200, 1500, 560, 4),
    (1352, 1988, 560, 4), (90, 70, 560, 4), (561, 560, 560, 4), (1120, 1120, 560, 4), (1121, 400, 560, 4), (2000, 2000, 560, 4),
    (560, 1680, 560, 4), (60, 90, 32, 4), (150, 500, 64, 4), (448, 448, 448, 4), (300, 1000, 224, 6), (900, 250, 224, 8),
]
QWEN_GEOMETRIES = [  # (H, W, min_pixels, max_pixels), patch 14 / merge 2 / temporal 2 as in the released configs
    (336, 336, 56 * 56, 28 * 28 * 1280), (512, 512, 56 * 56, 28 * 28 * 1280), (60, 90, 28 * 28, 28 * 28 * 16),
    (120, 150, 28 * 28, 28 * 28 * 36), (70, 70, 28 * 28, 28 * 28 * 36), (1352, 1988, 56 * 56, 28 * 28 * 1280),
    (40, 1000, 56 * 56, 28 * 28 * 1280), (1000, 40, 56 * 56, 28 * 28 * 1280), (28, 28, 56 * 56, 28 * 28 * 1280),
    (2000, 3000, 56 * 56, 28 * 28 * 1280), (337, 335, 56 * 56, 28 * 28 * 1280), (97, 130, 28 * 28, 28 * 28 * 64),
]


def golden_index_tensors():
    """the integer side tensors the HF image processors hand to the models, from the HF classes
    themselves (PIL backends, constructed offline with explicit parameters): Mllama aspect_ratio_ids /
    aspect_ratio_mask / num_tiles (the reference keeps them from processor(...) at llama32processor.py:119-147) and
    Qwen2-VL image_grid_thw (qwen2VLprocessor.py:68-96).  The plans' geometry and the plugins' index_tensors()
    must reproduce them exactly (tests/test_index_tensors.py)."""
    from PIL import Image
    from transformers.models.mllama.image_processing_pil_mllama import MllamaImageProcessorPil
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    ids, masks, tiles = [], [], []
    for H, W, tile, max_tiles in MLLAMA_GEOMETRIES:
        ip = MllamaImageProcessorPil(size={"height": tile, "width": tile}, max_image_tiles=max_tiles, image_mean=CLIP_MEAN,
                                     image_std=CLIP_STD)
        out = ip(images=[Image.fromarray(np.full((H, W, 3), 128, np.uint8))], return_tensors="pt")
        assert tuple(out["pixel_values"].shape) == (1, 1, max_tiles, 3, tile, tile)
        ids.append(int(out["aspect_ratio_ids"][0, 0]))
        m = out["aspect_ratio_mask"][0, 0].tolist()
        masks.append(m + [-1] * (8 - len(m)))
        tiles.append(int(out["num_tiles"][0][0]))
    grids = []
    for H, W, lo, hi in QWEN_GEOMETRIES:
        qp = Qwen2VLImageProcessorPil(patch_size=14, merge_size=2, temporal_patch_size=2, min_pixels=lo, max_pixels=hi)
        out = qp(images=[Image.fromarray(np.full((H, W, 3), 128, np.uint8))], return_tensors="pt")
        grids.append(out["image_grid_thw"][0].tolist() + [int(out["pixel_values"].shape[0]), int(out["pixel_values"].shape[1])])
    save("index_tensors.npz", mllama_geometry=np.asarray(MLLAMA_GEOMETRIES, np.int64), mllama_aspect_ratio_ids=np.asarray(ids, np.int64),
         mllama_aspect_ratio_mask=np.asarray(masks, np.int64), mllama_num_tiles=np.asarray(tiles, np.int64),
         qwen_geometry=np.asarray(QWEN_GEOMETRIES, np.int64), qwen_grid_thw_rows_cols=np.asarray(grids, np.int64))

def golden_full_size(llava, qwen, phi3, mllama):
    """Captures at BASELINE's two image sizes (336 x 336 synthetic, 512 x 512 `images/gray.png`) with the models' real
    processor parameters, for every family: pixel_values and image.grad of the reference classes, stored as checksums and
    sampled entries (inputs are rebuilt from their salts).  A file of its own, so the older fixtures regenerate bit for bit."""
    def ns(**kw):
        return SimpleNamespace(image_mean=CLIP_MEAN, image_std=CLIP_STD, do_convert_rgb=True, **kw)
    makers = {
        "llava": lambda: llava.DifferentiableLlavaImageProcessor(ns(crop_size={"height": 336, "width": 336}), "cpu"),
        "qwen2vl": lambda: qwen.DifferentiableQwen2VLImageProcessor(
            ns(patch_size=14, merge_size=2, temporal_patch_size=2, min_pixels=56 * 56, max_pixels=28 * 28 * 1280), "cpu"),
        "phi3": lambda: phi3.DifferentiablePhi3VImageProcessor(ns(num_crops=6, num_img_tokens=144), "cpu"),
        "mllama": lambda: mllama.DifferentiableMllamaImageProcessor(
            ns(do_normalize=True, size={"height": 560, "width": 560}, max_image_tiles=4, rescale_factor=1 / 255), "cpu"),
    }
    arrays, salt = {}, 500
    for fam, make in makers.items():
        for (H, W) in ((336, 336), (512, 512), (400, 600)):
            salt += 2
            proc = make()
            img = lcg_tensor((3, H, W), salt) + 0.5
            img.requires_grad_(True)
            out = proc.process(img)
            pv = out["pixel_values"]
            pv.backward(lcg_tensor(pv.shape, salt + 1))
            idx = np.arange(0, pv.numel(), 4999)
            gidx = np.arange(0, img.numel(), 499)
            k = f"{fam}_{H}x{W}"
            extra = [int(v) for key in ("num_tiles", "image_sizes", "num_img_tokens") if key in out and out[key] is not None
                     for v in np.asarray(out[key]).reshape(-1)]
            arrays.update({f"{k}_salt": np.array(salt), f"{k}_shape": np.array(pv.shape), f"{k}_ints": np.array(extra, dtype=np.int64),
                           f"{k}_pv_sum": pv.detach().double().sum(), f"{k}_pv_sumsq": (pv.detach().double() ** 2).sum(),
                           f"{k}_pv_idx": idx, f"{k}_pv_val": pv.detach().flatten()[idx],
                           f"{k}_grad_sum": img.grad.double().sum(), f"{k}_grad_sumsq": (img.grad.double() ** 2).sum(),
                           f"{k}_grad_idx": gidx, f"{k}_grad_val": img.grad.flatten()[gidx]})
    save("full_size_reference.npz", **arrays)

def golden_trainer_run(am, llava, qwen=None, mllama=None, phi3=None):
    """The reference's OWN `attack_model.train()` (attack_model.py:108-478), run here on the CPU for a few iterations around a
    tiny random LLaVA-architecture model: what it logs every iteration (loss, image loss, re-saved loss, quantise-error
    mean / std / L1, noise std, gradient norm, learning rate) and the images it writes.  This is the trainer loop itself -
    the order of operations, sigma_noise(t+1) = std(|q(s_t) - s_t|) through its PNG round trip, masked gradient, AdamW +
    StepLR cadence with gradient accumulation - not a restatement of it.

    What stands in, and why it does not touch the loop: `load_components` is pointed at the reference's own
    `AdvLlavaInputs` / `DifferentiableLlavaImageProcessor` and at a loader that returns the tiny random model with its toy
    processor (adversarialvlm_amd/testing/synthetic.py - there are no weights in the container); `wandb` is a recorder
    (its `log` calls ARE the capture); the question / answer pools are two neutral lines (the pools themselves are not part
    of the path and are not copied); the run happens in a scratch directory.  The noise is `torch.randn_like` on the global
    CPU generator, one draw per iteration and nothing else draws from it, so a test rebuilds it from the seed (its logged mean
    and std are kept as the check)."""
    import random
    import shutil
    import tempfile

    from PIL import Image

    sys.path.insert(0, repo_root())
    from adversarialvlm_amd.testing import synthetic

    class Recorder(types.ModuleType):
        def __init__(self):
            super().__init__("wandb")
            self.rows = []
            self.Table = type("Table", (), {"__init__": lambda self, *a, **k: None, "add_data": lambda self, *a: None})
            self.Image = lambda *a, **k: None

        def init(self, **kw):
            self.rows = []

        def log(self, d):
            self.rows.append({k: (float(v) if (torch.is_tensor(v) or isinstance(v, (int, float, np.floating))) else None)
                              for k, v in d.items()})

        def finish(self):
            pass

    rec = Recorder()
    sys.modules["wandb"] = rec
    am.wandb = rec
    q, a = types.ModuleType("questions"), types.ModuleType("answers")
    q.not_safe_questions = ["what is shown here", "describe the scene please", "list the items in this picture"]
    q.questions = ["describe this image", "what is in this picture", "hi", "what is shown in region 3 of the image"]
    q.not_safe_questions_test = ["hi", "what is in this picture"]
    a.answers, a.adv_answers = ["sure here it is", "of course the answer is"], ["yes here is the list"]
    sys.modules["questions"], sys.modules["answers"] = q, a

    from adversarialvlm_amd.testing import synthetic_vlms

    def loader(model_name, device):
        model, proc = synthetic.load_model_and_processor("synthetic/tiny-llava", device, seed=0)
        return model, proc

    def load_mllama(model_name, device):
        return synthetic_vlms.load_model_and_processor("synthetic/tiny-mllama", device, seed=2)

    def load_qwen(model_name, device):
        model, proc = synthetic_vlms.load_model_and_processor("synthetic/tiny-qwen2vl", device, seed=3)
        # transformers 4.51 (the reference's pin) kept the pixel bounds as attributes; 5.x keeps them in `size`
        proc.image_processor.min_pixels, proc.image_processor.max_pixels = synthetic_vlms.QWEN_MIN_PIXELS, synthetic_vlms.QWEN_MAX_PIXELS
        return model, proc
    table = {"tiny": (loader, llava.AdvLlavaInputs, llava.DifferentiableLlavaImageProcessor)}
    if phi3 is not None:
        # Phi-3.5-Vision's processor and model are remote code: the reference's OWN plugin pair (AdvPhiInputs with its
        # batch_processing / pad_left, DifferentiablePhi3VImageProcessor) runs here around the interface twin of
        # adversarialvlm_amd/testing/synthetic_phi3v.py.  The plugin moves every batch `.to("cuda:0")` (phi3processor.py:284,
        # :302): in this container without a GPU that move is made the identity it would be on a one-device host.
        from adversarialvlm_amd.testing import synthetic_phi3v
        _cuda_moves_are_identity_without_a_gpu()
        table["tiny-phi3v"] = ((lambda name, dev: synthetic_phi3v.load_model_and_processor("synthetic/tiny-phi3v", dev, seed=4)),
                               phi3.AdvPhiInputs, phi3.DifferentiablePhi3VImageProcessor)
    if mllama is not None:
        table["tiny-mllama"] = (load_mllama, mllama.AdvMllamaInputs, mllama.DifferentiableMllamaImageProcessor)
    if qwen is not None:
        table["tiny-qwen2vl"] = (load_qwen, qwen.AdvQwen2VLInputs, qwen.DifferentiableQwen2VLImageProcessor)
    am.load_components = lambda name: table[name]

    runs = [("a", dict(grad_accum_steps=1, mask_type="corner", mask_size=30, scheduler_step_size=2, scheduler_gamma=0.5,
                       start_from_white=False), (3, 64, 48), 5, 11),
            ("b", dict(grad_accum_steps=2, mask_type=None, mask_size=None, scheduler_step_size=100, scheduler_gamma=1.0,
                       start_from_white=False), (3, 56, 56), 6, 12),
            ("c", dict(grad_accum_steps=1, mask_type="bottom_lines", mask_size=20, scheduler_step_size=100, scheduler_gamma=1.0,
                       start_from_white=True), (3, 40, 70), 3, 13),
            # BASELINE configs[2] at the reference's level: a Llama-3.2-Vision architecture, tanh + localized patch (corner mask);
            # batch 1 (the reference hands the HF Mllama processor a flat image list: transformers 5.x takes it for one prompt)
            ("d", dict(grad_accum_steps=1, mask_type="corner", mask_size=40, scheduler_step_size=100, scheduler_gamma=1.0,
                       start_from_white=False, model_name="tiny-mllama", batch_size=1), (3, 60, 90), 4, 14),
            ("e", dict(grad_accum_steps=2, mask_type=None, mask_size=None, scheduler_step_size=100, scheduler_gamma=1.0,
                       start_from_white=False, model_name="tiny-qwen2vl", batch_size=1), (3, 60, 90), 4, 15),
            # the draws of the loop: prompts sampled with replacement from the pool (--prompt list) and a target drawn per
            # iteration (--target_text_random: multi-answer supervision), both from the global `random` stream (:283-292)
            ("f", dict(grad_accum_steps=1, mask_type=None, mask_size=None, scheduler_step_size=100, scheduler_gamma=1.0,
                       start_from_white=False, batch_size=3, prompt="list", target_text_random=True), (3, 56, 56), 5, 16),
            ("h", dict(grad_accum_steps=1, mask_type="corner", mask_size=40, scheduler_step_size=100, scheduler_gamma=1.0,
                       start_from_white=False, model_name="tiny-phi3v", batch_size=2), (3, 60, 90), 3, 18),
            # Qwen2-VL with prompts of different lengths in one batch: the REAL fast tokenizer's left padding, `mm_token_type_ids`
            # padded with it, three images in one processor call (qwen2VLprocessor.py:68-96)
            ("j", dict(grad_accum_steps=1, mask_type=None, mask_size=None, scheduler_step_size=100, scheduler_gamma=1.0,
                       start_from_white=False, model_name="tiny-qwen2vl", batch_size=3, prompt="list"), (3, 60, 90), 3, 19),
            # --restart_num: the reference clamps and re-quantises a LOCAL x every restart_num iterations (:447-457), which the next
            # iteration overwrites from p - a no-op on the trajectory (Q5); run i is run a with restart_num 2
            ("i", dict(grad_accum_steps=1, mask_type="corner", mask_size=30, scheduler_step_size=2, scheduler_gamma=0.5,
                       start_from_white=False, restart_num=2), (3, 64, 48), 5, 11),
            # BASELINE configs[0] as written: tanh-clamp attack, 1 prompt, 2 PGD steps, on the CPU - the reference's own run of it
            ("g", dict(grad_accum_steps=1, mask_type=None, mask_size=None, scheduler_step_size=100, scheduler_gamma=1.0,
                       start_from_white=False, batch_size=1), (3, 56, 56), 2, 17)]
    arrays = {}
    cwd = os.getcwd()
    for name, kw, ishape, iters, seed in runs:
        kw = dict(kw)
        model_name, batch = kw.pop("model_name", "tiny"), kw.pop("batch_size", 2)
        prompt, ttr = kw.pop("prompt", "describe this image"), kw.pop("target_text_random", False)
        restart = kw.pop("restart_num", 0)
        if model_name not in table:
            continue
        tmp = tempfile.mkdtemp()
        os.chdir(tmp)
        try:
            rng = np.random.default_rng(seed)
            img = (rng.random((ishape[1], ishape[2], 3)) * 255).astype(np.uint8)
            img[: ishape[1] // 4, : ishape[2] // 3] = 0          # zeros: the default mask (x_0 != 0) must leave them alone
            Image.fromarray(img).save("in.png")
            random.seed(seed)
            torch.manual_seed(seed)
            am.train(exp_name="run", img_orig="in.png", prompt=prompt, target_text="sure here it is",
                     model_name=model_name, lr=1e-2, num_iterations=iters, save_steps=2, batch_size=batch, restart_num=restart,
                     clamp_method="tanh", epsilon=0.5, sigma=1e-3, target_text_random=ttr, **kw)
            per_iter = [r for r in rec.rows if "loss_resaved" in r]
            assert len(per_iter) == iters
            arrays[f"{name}_log_keys"] = np.array(sorted({k for r in rec.rows for k in r}))      # every key the run handed to wandb.log
            arrays[f"{name}_accumulated_loss"] = np.array([r["accumulated_loss"] for r in rec.rows if "accumulated_loss" in r])
            keys = ["loss", "image_loss", "loss_resaved", "resave_error_mean", "resave_error_std", "resave_error_l1", "noise_mean",
                    "noise_std", "adversarial_mean", "adversarial_std", "lr", "grad norm", "global_iteration"]
            for k in keys:
                arrays[f"{name}_{k.replace(' ', '_')}"] = np.array([r[k] for r in per_iter], dtype=np.float64)
            arrays[f"{name}_image"] = img
            arrays[f"{name}_seed"] = np.array(seed)
            arrays[f"{name}_model"], arrays[f"{name}_batch"] = np.array(model_name), np.array(batch)
            arrays[f"{name}_prompt"], arrays[f"{name}_target_random"] = np.array(prompt), np.array(int(ttr))
            arrays[f"{name}_restart"] = np.array(restart)
            arrays[f"{name}_questions"] = np.array(q.not_safe_questions + q.questions)        # the pool as train() forms it (:144)
            arrays[f"{name}_answers"] = np.array(a.answers + a.adv_answers)                   # (:147-148)
            arrays[f"{name}_iters"] = np.array(iters)
            arrays[f"{name}_accum"] = np.array(kw["grad_accum_steps"])
            arrays[f"{name}_sched"] = np.array([kw["scheduler_step_size"], kw["scheduler_gamma"]], dtype=np.float64)
            arrays[f"{name}_mask"] = np.array([{"corner": 0, "bottom_lines": 1, None: -1}[kw["mask_type"]], kw["mask_size"] or 0])
            arrays[f"{name}_white"] = np.array(int(kw["start_from_white"]))
            arrays[f"{name}_final"] = np.fromfile(os.path.join("runs", "run", "optimized_image_iter_final.bin"), dtype=np.float32)
            arrays[f"{name}_final_png"] = np.asarray(Image.open(os.path.join("runs", "run", "optimized_image_iter_final.png")).convert("RGB"))
            arrays[f"{name}_mask_sum"] = np.array(float(torch.load(os.path.join("runs", "run", "mask.pt")).sum()))
            arrays[f"{name}_files"] = np.array(sorted(f for f in os.listdir(os.path.join("runs", "run"))))
            import csv
            with open(os.path.join("runs", "run", "test_results_iter_0.csv"), newline="", encoding="utf-8") as fcsv:
                arrays[f"{name}_probe0"] = np.array([row for row in csv.reader(fcsv)])     # the generation probe at iteration 0 (train_test.py)
            arrays[f"{name}_probe0_stats"] = np.array([[r.get(k) for k in ("test_target_first_word_acc", "test_target_acc", "test_refuse_count",
                                                                           "test_total_questions")]
                                                       for r in rec.rows if "test_target_acc" in r][0], dtype=np.float64)
            # the batch the reference's AdvLlavaInputs assembles for this prompt (llavaprocessor.py:80-108), for the id layout
            load_fn, AdvCls, _ = table[model_name]
            _, proc = load_fn(model_name, "cpu")
            ip = AdvCls(questions=["describe this image"], test_questions=["hi"], batch_size=batch,
                        original_image=Image.fromarray(img), processor=proc, device="cpu", target_text="sure here it is")
            random.seed(1234)
            enc = ip.get_inputs_train()
            for key in enc.keys():
                if key != "pixel_values":
                    arrays[f"{name}_in_{key}"] = enc[key]
            arrays[f"{name}_suffix"] = np.array([ip.suffix_length, ip.shift])
            inf = ip.get_inputs_inference(Image.fromarray(img), question="what is in this picture")     # :110-133 of the plugins
            for key in inf.keys():
                if key != "pixel_values":
                    arrays[f"{name}_inf_{key}"] = inf[key]
        finally:
            os.chdir(cwd)
            shutil.rmtree(tmp, ignore_errors=True)
    save("trainer_run_reference.npz", **arrays)

def _cuda_moves_are_identity_without_a_gpu():
    """The Phi-3.5 plugin moves every batch `.to("cuda:0")` (phi3processor.py:284,:302): in this container without a GPU that move
    is made the identity it would be on a one-device host."""
    from transformers.feature_extraction_utils import BatchFeature
    if not torch.cuda.is_available() and not getattr(BatchFeature.to, "_cpu_only_container", False):
        plain_to = BatchFeature.to

        def to(self, *args, **kwargs):
            args = tuple("cpu" if (isinstance(a, str) and a.startswith("cuda")) else a for a in args)
            return plain_to(self, *args, **kwargs)
        to._cpu_only_container = True
        BatchFeature.to = to


def golden_cross_trainer_run(llava, qwen, mllama, phi3=None):
    """The reference's OWN `crossattack_models.train()` (crossattack_models.py:124-519) on the CPU: (x1) two tiny random
    LLaVA models with weights 0.7 / 0.3, batch 2, gradient accumulation 2, StepLR decay, a corner mask; (x2) one model of each
    family whose architecture ships with transformers - LLaVA, Llama-3.2-Vision, Qwen2-VL (adversarialvlm_amd/processors/
    synthetic*.py) - behind the reference's own AdvLlavaInputs / AdvMllamaInputs / AdvQwen2VLInputs and Differentiable*Processor
    classes, batch 1 (the reference hands the HF Mllama processor a flat image list, which transformers 5.x accepts for one
    prompt only).  Stand-ins as in golden_trainer_run: loader, wandb recorder, neutral question pools, scratch directory.
    Per iteration and model the loop draws `torch.randn_like(pixel_values_i)` from the global CPU generator, in model order,
    after the models were built in model order - a test rebuilds the draws from that."""
    import importlib
    import random
    import shutil
    import tempfile

    from PIL import Image

    sys.path.insert(0, repo_root())
    from adversarialvlm_amd.testing import synthetic, synthetic_vlms

    import_reference_trainer()                      # placeholders for wandb / torchvision names
    cm = importlib.import_module("crossattack_models")

    class Recorder(types.ModuleType):
        def __init__(self):
            super().__init__("wandb")
            self.rows = []
            self.Table = type("Table", (), {"__init__": lambda self, *a, **k: None, "add_data": lambda self, *a: None})
            self.Image = lambda *a, **k: None

        def init(self, **kw):
            self.rows = []

        def log(self, d):
            self.rows.append({k: (float(v) if (torch.is_tensor(v) or isinstance(v, (int, float, np.floating))) and not isinstance(v, bool)
                                  else None) for k, v in d.items()})

        def finish(self):
            pass

    rec = Recorder()
    sys.modules["wandb"] = rec
    cm.wandb = rec
    q, a = types.ModuleType("questions"), types.ModuleType("answers")
    q.not_safe_questions = ["what is shown here", "describe the scene please", "list the items in this picture"]
    q.questions = ["describe this image", "what is in this picture", "hi", "what is shown in region 3 of the image"]
    q.not_safe_questions_test = ["hi"]
    a.answers, a.adv_answers = ["sure here it is", "of course the answer is"], ["yes here is the list"]
    sys.modules["questions"], sys.modules["answers"] = q, a

    def load_qwen(name, dev):
        model, proc = synthetic_vlms.load_model_and_processor("synthetic/tiny-qwen2vl", dev, seed=3)
        # transformers 4.51 (the reference's pin) kept the pixel bounds as attributes; 5.x keeps them in `size`
        proc.image_processor.min_pixels, proc.image_processor.max_pixels = synthetic_vlms.QWEN_MIN_PIXELS, synthetic_vlms.QWEN_MAX_PIXELS
        return model, proc

    table = {
        "tiny-llava-0": (lambda name, dev: synthetic.load_model_and_processor("synthetic/tiny-llava", dev, seed=0),
                         llava.AdvLlavaInputs, llava.DifferentiableLlavaImageProcessor),
        "tiny-llava-1": (lambda name, dev: synthetic.load_model_and_processor("synthetic/tiny-llava", dev, seed=1),
                         llava.AdvLlavaInputs, llava.DifferentiableLlavaImageProcessor),
        "tiny-mllama": (lambda name, dev: synthetic_vlms.load_model_and_processor("synthetic/tiny-mllama", dev, seed=2),
                        mllama.AdvMllamaInputs, mllama.DifferentiableMllamaImageProcessor),
        "tiny-qwen2vl": (load_qwen, qwen.AdvQwen2VLInputs, qwen.DifferentiableQwen2VLImageProcessor),
    }
    if phi3 is not None:
        from adversarialvlm_amd.testing import synthetic_phi3v
        _cuda_moves_are_identity_without_a_gpu()
        table["tiny-phi3v"] = ((lambda name, dev: synthetic_phi3v.load_model_and_processor("synthetic/tiny-phi3v", dev, seed=4)),
                               phi3.AdvPhiInputs, phi3.DifferentiablePhi3VImageProcessor)
    cm.load_components = lambda name: table[name]

    runs = [("x1", ["tiny-llava-0", "tiny-llava-1"], dict(batch_size=2, grad_accum_steps=2, scheduler_step_size=1, scheduler_gamma=0.5,
                                                         mask_type="corner", mask_size=30, model_weights=[0.7, 0.3]), (64, 48), 6, 21),
            ("x2", ["tiny-llava-0", "tiny-mllama", "tiny-qwen2vl"],
             dict(batch_size=1, grad_accum_steps=1, scheduler_step_size=100, scheduler_gamma=0.9, mask_type=None, mask_size=None,
                  model_weights=None), (60, 90), 3, 22),
            # the draws of the cross loop (:303-321): a coin per iteration, a refusal per model below refuse_prob, else ONE target
            # for all models; prompts sampled from the pool
            ("x3", ["tiny-llava-0", "tiny-llava-1"],
             dict(batch_size=2, grad_accum_steps=1, scheduler_step_size=100, scheduler_gamma=0.9, mask_type=None, mask_size=None,
                  model_weights=[1.0, 0.5], prompt="list", target_text_random=True, DPO_flag=True, refuse_prob=0.5), (56, 56), 6, 23),
            # BASELINE configs[3] by name: Phi-3.5-Vision + Qwen2-VL + Llama-3.2-Vision, the weights of scripts/attacks/attack_cross.sh
            ("x4", ["tiny-phi3v", "tiny-qwen2vl", "tiny-mllama"],
             dict(batch_size=1, grad_accum_steps=1, scheduler_step_size=100, scheduler_gamma=0.9, mask_type=None, mask_size=None,
                  model_weights=[0.2, 0.8, 1.6]), (60, 90), 3, 24)]
    arrays = {}
    cwd = os.getcwd()
    for name, names, kw, (H, W), iters, seed in runs:
        if any(m not in table for m in names):
            continue
        kw = dict(kw)
        prompt, ttr = kw.pop("prompt", "describe this image"), kw.pop("target_text_random", False)
        dpo, refuse_prob = kw.pop("DPO_flag", False), kw.pop("refuse_prob", 0.1)
        tmp = tempfile.mkdtemp()
        os.chdir(tmp)
        try:
            rng = np.random.default_rng(seed)
            img = (rng.random((H, W, 3)) * 255).astype(np.uint8)
            img[: H // 4, : W // 3] = 0
            Image.fromarray(img).save("in.png")
            random.seed(seed)
            torch.manual_seed(seed)
            cm.train(exp_name="run", img_orig="in.png", prompt=prompt, target_text="sure here it is",
                     model_names=names, lr=1e-2, num_iterations=iters, save_steps=2, restart_num=0, clamp_method="tanh",
                     # Q3: the amplitude of the perturbation is `attack_norm`, --epsilon only reaches the log (:277, :329);
                     # --sigma is not used either, the first noise level is a literal 0.001 (:299): both set to values that would show
                     epsilon=0.3, sigma=5e-3, attack_norm=0.4,
                     start_from_white=False, target_text_random=ttr, DPO_flag=dpo, refuse_prob=refuse_prob, **kw)
            per_iter = [r for r in rec.rows if "loss_per_iteration" in r]
            assert len(per_iter) == iters
            arrays[f"{name}_log_keys"] = np.array(sorted({k for r in rec.rows for k in r}))
            arrays[f"{name}_accumulated_loss"] = np.array([r["accumulated_loss"] for r in rec.rows if "accumulated_loss" in r])
            keys = ["loss_per_iteration", "img_loss", "loss_resaved", "resave_error_mean", "resave_error_std", "resave_error_l1",
                    "noise_std", "adversarial_mean", "adversarial_std", "lr", "grad_norm", "global_iteration"]
            for k in keys:
                arrays[f"{name}_{k}"] = np.array([r[k] for r in per_iter], dtype=np.float64)
            arrays[f"{name}_model_losses"] = np.array([[r[f"loss_{i}_{mn}"] for i, mn in enumerate(names)] for r in per_iter])
            arrays[f"{name}_names"] = np.array(names)
            arrays[f"{name}_prompt"], arrays[f"{name}_target_random"] = np.array(prompt), np.array(int(ttr))
            arrays[f"{name}_dpo"] = np.array([float(dpo), refuse_prob])
            arrays[f"{name}_seed"] = np.array(seed)
            arrays[f"{name}_questions"] = np.array(q.not_safe_questions + q.questions)
            arrays[f"{name}_answers"] = np.array(a.answers + a.adv_answers)
            arrays[f"{name}_image"] = img
            arrays[f"{name}_iters"], arrays[f"{name}_batch"], arrays[f"{name}_accum"] = np.array(iters), np.array(kw["batch_size"]), np.array(kw["grad_accum_steps"])
            arrays[f"{name}_sched"] = np.array([kw["scheduler_step_size"], kw["scheduler_gamma"]], dtype=np.float64)
            arrays[f"{name}_mask"] = np.array([{"corner": 0, "bottom_lines": 1, None: -1}[kw["mask_type"]], kw["mask_size"] or 0])
            arrays[f"{name}_weights"] = np.array(kw["model_weights"] or [1.0] * len(names))
            arrays[f"{name}_final"] = np.fromfile(os.path.join("runs", "run", "optimized_image_iter_final.bin"), dtype=np.float32)
            arrays[f"{name}_files"] = np.array(sorted(os.listdir(os.path.join("runs", "run"))))
        finally:
            os.chdir(cwd)
            shutil.rmtree(tmp, ignore_errors=True)
    save("cross_trainer_run_reference.npz", **arrays)

def golden_cli_flags(am):
    """The command lines of the two trainers as the reference's OWN `main()` functions build them (attack_model.py:482-519,
    crossattack_models.py:527-575): every option with its type, default, choices and action, read off the parser object at the
    moment `parse_args` is called.  The free-text defaults of --exp_name / --img_orig / --prompt / --target_text are research
    content and are not stored (this package's defaults for them are neutral on purpose)."""
    import argparse
    import importlib
    import json
    cm = importlib.import_module("crossattack_models")

    class Caught(Exception):
        pass

    def grab(module):
        plain = argparse.ArgumentParser.parse_args

        def parse_args(self, *a, **k):
            raise Caught(self)
        argparse.ArgumentParser.parse_args = parse_args
        try:
            module.main()
        except Caught as c:
            parser = c.args[0]
        finally:
            argparse.ArgumentParser.parse_args = plain
        rows = []
        for act in parser._actions:
            if not act.option_strings or act.dest == "help":
                continue
            free_text = act.dest in ("exp_name", "img_orig", "prompt", "target_text")
            rows.append(dict(flag=act.option_strings[0], dest=act.dest, action=type(act).__name__,
                             type=getattr(act.type, "__name__", None) if act.type is not None else None,
                             default=None if free_text else act.default, default_stored=not free_text,
                             choices=list(act.choices) if act.choices else None, nargs=act.nargs))
        return rows
    import inspect

    def signature(fn):
        return [[n, None if prm.default is inspect.Parameter.empty else prm.default, prm.default is not inspect.Parameter.empty]
                for n, prm in inspect.signature(fn).parameters.items()]
    from processors import llavaprocessor, llama32processor, phi3processor, qwen2VLprocessor
    classes = {}
    for mod, adv, diff in ((llavaprocessor, "AdvLlavaInputs", "DifferentiableLlavaImageProcessor"),
                           (llama32processor, "AdvMllamaInputs", "DifferentiableMllamaImageProcessor"),
                           (phi3processor, "AdvPhiInputs", "DifferentiablePhi3VImageProcessor"),
                           (qwen2VLprocessor, "AdvQwen2VLInputs", "DifferentiableQwen2VLImageProcessor")):
        for cname in (adv, diff):
            cls = getattr(mod, cname)
            classes[cname] = {"init": signature(cls.__init__),
                              "methods": sorted(n for n, v in vars(cls).items() if callable(v) and not n.startswith("_"))}
            if hasattr(cls, "refuses"):      # the refusal prefixes are part of the plugin contract: count and digest, not the text
                import hashlib
                classes[cname]["refuses"] = [len(cls.refuses), hashlib.sha256("\n".join(cls.refuses).encode()).hexdigest()]
    import processors as ref_registry
    model_map = {k: [v["module"].split(".")[-1], v["input_class"], v["processor_class"]] for k, v in ref_registry.MODEL_MAP.items()}
    # the launch scripts: which trainer each one starts and which options it passes (names only; the values are run settings
    # and free text).  Lines that are commented out do not count.
    import re
    scripts = {}
    sdir = os.path.join(os.path.dirname(REF), "scripts", "attacks")
    for fn in sorted(os.listdir(sdir)):
        if not fn.endswith(".sh"):
            continue
        live = [ln for ln in open(os.path.join(sdir, fn)).read().splitlines() if not ln.lstrip().startswith("#")]
        text = "\n".join(ln.split(" #")[0] for ln in live)
        entry = re.findall(r"src/([\w-]+)\.py", text)
        flags = re.findall(r"(?<![\w-])(--[A-Za-z_]\w*)", text)
        if entry:
            scripts[fn] = {"entry": entry[0], "flags": sorted(set(flags)),
                           "entry_in_tree": os.path.exists(os.path.join(REF, entry[0] + ".py"))}
    data = {"attack_model": grab(am), "crossattack_models": grab(cm), "meta": meta(), "launch_scripts": scripts,
            "train_signatures": {"attack_model": signature(am.train), "crossattack_models": signature(cm.train)},
            "plugin_classes": classes, "model_map": model_map}
    with open(os.path.join(OUT, "cli_flags_reference.json"), "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print("wrote cli_flags_reference.json", {k: len(v) for k, v in data.items() if k != "meta"})


def main():
    """python tests/golden/make_golden.py [--only full_size]   (--only: just that fixture file, the others stay untouched)"""
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present: fixtures can only be regenerated in the build container")
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    if only not in (None, "full_size", "trainer_run", "cross_trainer_run", "cli_flags"):
        raise SystemExit("--only knows: full_size, trainer_run, cross_trainer_run, cli_flags")
    if only is None:
        golden_index_tensors()        # before the torchvision stub of import_reference() exists
    llava, qwen, phi3 = import_reference()
    mllama = import_reference_mllama()
    if only in (None, "full_size"):
        golden_full_size(llava, qwen, phi3, mllama)
    if only == "trainer_run":
        golden_trainer_run(import_reference_trainer(), llava, qwen, mllama, phi3)
    if only == "cross_trainer_run":
        golden_cross_trainer_run(llava, qwen, mllama, phi3)
    if only == "cli_flags":
        golden_cli_flags(import_reference_trainer())
    if only is not None:
        return
    golden_llava(llava)
    golden_qwen(qwen)
    golden_phi3(phi3)
    golden_suffix_loss(llava)
    golden_mllama_helpers()
    golden_mllama_reference(mllama)
    golden_trainer_helpers(import_reference_trainer())
    golden_closed_form()
    golden_mllama_restated()
    golden_cli_flags(import_reference_trainer())
    golden_trainer_run(import_reference_trainer(), llava, qwen, mllama, phi3)      # last: these replace the wandb placeholder by a recorder
    golden_cross_trainer_run(llava, qwen, mllama, phi3)


if __name__ == "__main__":
    main()
