"""
Build recipe of libsvmc.so (the only native artefact of the package): hipcc, gfx950 only, in-tree.

    python -m stochvolmodels_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libsvmc.so")
SOURCES = ("svmc_runtime.hip", "svmc_kernels.hip", "svmc_analytic.hip", "svmc_chain.hip", "svmc_comm.hip")
HEADERS = ("svmc_internal.h", "svmc_models.h", "svmc_rng.h", "svmc_math.h", "svmc_log_table.h")
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def flags() -> list:
    # --align-all-blocks=4: every basic block starts 16-byte aligned.  The stepping loop's time depends on where its
    # header falls in instruction memory: instruction-for-instruction identical builds measured 3.80 ms (header at
    # +0x6e0) and 4.08 ms (+0x6ec: the 8-byte encodings straddle the fetch granule) on the same box; with aligned
    # blocks every build gets the fast placement (tools/ubench/ab_kernel.py).
    return ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-fno-gpu-rdc",
            "-mllvm", "--align-all-blocks=4", "-I" + INCLUDE, "-I" + CSRC, "-DSVMC_BUILDING=1", "-Wall",
            "-Wno-unused-function"]


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(INCLUDE, "svmc.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc()] + flags() + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
