"""Development tools: set the library's tuning switches from the environment.

    ADVX_TUNE="8=0,6=2" python tools/generic_bench.py     # advx_set_tuning(8, 0); advx_set_tuning(6, 2)

(the numbers are the ADVX_TUNE_* codes of include/advx.h).  Tools only - the package never reads this variable.
"""
import os


def apply_env_tuning():
    spec = os.environ.get("ADVX_TUNE", "").strip()
    if not spec:
        return []
    from adversarialvlm_amd import _lib
    lib, done = _lib.load(), []
    for item in spec.split(","):
        what, value = (int(t) for t in item.split("="))
        _lib.check(lib.advx_set_tuning(what, value), "advx_set_tuning")
        done.append((what, value))
    return done
