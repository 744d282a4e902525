"""Build libadvx_hip.so in-tree with hipcc for gfx950 (no JIT cache, no CPU variant)."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = [os.path.join(CSRC, "advx.hip")]
PUBLIC_HEADER = os.path.join(os.path.dirname(_HERE), "include", "advx.h")


def headers():
    """Every header the library is compiled from: all of csrc/*.h (found, not listed - a new header cannot be
    forgotten) and the public include/advx.h."""
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [PUBLIC_HEADER]


OUT = os.path.join(_HERE, "libadvx_hip.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in SOURCES + headers())


def build_library(force=False, verbose=False):
    """Compile the HIP library if it is missing or older than its sources."""
    if not force and not _stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libadvx_hip.so cannot be built (there is no CPU fallback)")
    # compile beside the target and rename: concurrent builders (one rank per GPU) can never
    # leave a half-written library behind, and a process that has the old one mapped keeps it
    tmp = f"{OUT}.{os.getpid()}.tmp"
    cmd = [hipcc] + FLAGS + ["-o", tmp] + SOURCES
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, OUT)
    return OUT
