"""Random geometries through the HOST side of the C ABI (no GPU): plan creation for the four processor kinds, the
integer description, every tap table (forward and transposed), the layout index maps and the composed-crop predicate and
row lengths for random windows.  Meant to run under tools/asan_host.sh (ASan + UBSan on the host code); on its own it
checks that every call returns ADVX_OK or a clean error and that table rows stay inside their source.
    python tools/fuzz_host_geometry.py [cases] [seed]"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adversarialvlm_amd import _lib as L  # noqa: E402
from adversarialvlm_amd import ops  # noqa: E402
from adversarialvlm_amd.plan import Plan  # noqa: E402


def one(rng):
    H, W = (rng.choice([rng.randint(1, 64), rng.randint(28, 700), rng.randint(300, 1400)]) for _ in range(2))
    kind = rng.randrange(4)
    if kind == 0:
        plan = Plan.llava(H, W, rng.choice([336, 224, rng.randint(8, 400)]), rng.choice([336, 224, rng.randint(8, 400)]))
    elif kind == 1:
        plan = Plan.mllama(H, W, tile=rng.choice([560, 448, 56, 14 * rng.randint(1, 30)]), max_tiles=rng.randint(1, 4))
    elif kind == 2:
        plan = Plan.phi3(H, W, num_crops=rng.choice([4, 6, 16, rng.randint(1, 16)]))
    else:
        lo = 56 * 56 * rng.choice([1, 1, 4])
        plan = Plan.qwen2vl(H, W, min_pixels=lo, max_pixels=max(lo, 28 * 28 * rng.choice([16, 256, 1280])))
    info = plan.info
    assert plan.out_numel > 0 and int(info.n_stage) >= 1
    for st in range(int(info.n_stage)):
        for axis in (0, 1):
            for tr in (False, True):
                start, count, w = plan.taps(st, axis, tr)
                assert (count >= 0).all() and (start >= 0).all() and (count <= w.shape[1]).all()
                assert np.isfinite(w).all()
        s = plan.stage(st)
        for _ in range(4):
            c, y, x = rng.randrange(3), rng.randrange(int(s.can_h)), rng.randrange(int(s.can_w))
            for idx in plan.out_index(st, c, y, x):
                assert 0 <= idx < plan.out_numel
    for _ in range(3):
        h, w = rng.randint(1, H), rng.randint(1, W)
        win = (rng.randint(0, H - h), rng.randint(0, W - w), h, w)
        if ops.crop_composes(plan, H, W, win):
            f, t = ops.crop_compose_strides(plan, H, W, win)
            assert min(f + t) >= 1
    return kind


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261004
    rng = random.Random(seed)
    done, refused = [0, 0, 0, 0], 0
    for i in range(n):
        try:
            done[one(rng)] += 1
        except L.AdvxError:
            refused += 1        # a geometry the library declines with an error code is fine; a crash is not
    print(f"fuzz_host_geometry: {sum(done)} plans built (llava/mllama/phi3/qwen2vl = {done}), {refused} declined with an error, seed {seed}")


if __name__ == "__main__":
    main()
