#!/usr/bin/env python3
"""Soak of the trainers on one GPU box (development tool): two data-parallel ranks sharing the GPU (gloo for the host
side, the peer exchange for the image gradient) run the single-model trainer on a 512 x 512 image with blur 9 and a
random-resized crop for --iters iterations with replica checks; then one rank runs the cross-model trainer (two tiny
models, blur 5 + crop).  Passes if nothing hangs, no replica check fires and the losses stay finite.

    python tools/soak_dp.py --iters 300
"""
import argparse
import os
import socket
import sys
import tempfile
import time

os.environ.setdefault("ADVX_PLUGIN_MODULES", "adversarialvlm_amd.testing")     # the synthetic/* models, also in the spawned ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402
from PIL import Image  # noqa: E402


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def common(tmp, name, iters):
    return dict(exp_name=name, img_orig=os.path.join(tmp, "gray512.png"), prompt="list", target_text="sure here it is",
                lr=1e-2, num_iterations=iters, save_steps=max(50, iters // 4), batch_size=8, grad_accum_steps=1,
                scheduler_step_size=100, scheduler_gamma=0.9, restart_num=0, mask_type=None, mask_size=None,
                clamp_method="tanh", epsilon=0.5, sigma=1e-3, start_from_white=False, target_text_random=False,
                base_path=tmp, seed=11, log_every=50)


def rank_main(rank, world, port, tmp, iters, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo")
    from adversarialvlm_amd import attack_model
    t0 = time.time()
    eng, hist = attack_model.train(model_name="synthetic/tiny-llava", use_gaussian_blur=True, gblur_kernel_size=9, gblur_sigma=3.0,
                                   use_local_crop=True, exchange_transport="peer", replica_check_every=25, return_engine=True,
                                   **common(tmp, "soak_dp", iters))
    losses = [h["loss"] if isinstance(h, dict) and "loss" in h else float("nan") for h in (hist or [])]
    out[rank] = dict(seconds=time.time() - t0, mode=eng.mode, finite=bool(torch.isfinite(eng.p).all()),
                     last_loss=losses[-1] if losses else None, timed_out=bool(eng.peer.timed_out()) if eng.peer else None)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="advx_soak_")
    Image.fromarray(np.full((512, 512, 3), 128, np.uint8)).save(os.path.join(tmp, "gray512.png"))
    out = mp.Manager().dict()
    mp.spawn(rank_main, args=(2, free_port(), tmp, args.iters, out), nprocs=2, join=True)
    for r in range(2):
        print(f"dp rank {r}: {out[r]}", flush=True)
        assert out[r]["finite"] and not out[r]["timed_out"], out[r]
    from adversarialvlm_amd import crossattack_models
    from adversarialvlm_amd.processors import load_components
    load, AdvInputs, DiffProc = load_components("synthetic/tiny-llava")
    comps = {"synthetic/tiny-llava": (load, AdvInputs, DiffProc),
             "synthetic/tiny-llava-b": (lambda name, device: load("synthetic/tiny-llava", device, seed=1), AdvInputs, DiffProc)}
    t0 = time.time()
    kw = common(tmp, "soak_cross", args.iters)
    kw.pop("batch_size")
    eng, hist = crossattack_models.train(model_names=["synthetic/tiny-llava", "synthetic/tiny-llava-b"], batch_size=4,
                                         model_weights=[0.2, 0.8], use_gaussian_blur=True, gblur_kernel_size=5,
                                         use_local_crop=True, return_engine=True, DPO_flag=False, components=comps, **kw)
    print(f"cross: {time.time() - t0:.1f} s, mode {eng.mode}, finite {bool(torch.isfinite(eng.p).all())}", flush=True)
    assert torch.isfinite(eng.p).all()
    print("soak ok")


if __name__ == "__main__":
    main()
