"""
Build recipe of libsvmc.so (the only native artefact of the package): hipcc, gfx950 only, in-tree.

    python -m stochvolmodels_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libsvmc.so")
# written beside the library by every build: the instruction histogram of the stepping kernels' time loops, read off the
# compiler's own assembly of THIS build, and the library's sha256 -- bench.py prices its roofline from it and marks the
# line stale when the library it loaded is not the one the histogram describes
ISA_JSON = os.path.join(PKG, "libsvmc.isa.json")
ISA_KERNELS = ("logsv_rng_kernel", "logsv_chain_rng_kernel", "heston_rng_kernelILi0", "heston_rng_kernelILi1", "heston_rng_kernelILi2",
               "logsv_rng_lat_kernelILi4ELi4ELi256", "logsv_chain_rng_lat_kernelILi4ELi4ELi256")
SOURCES = ("svmc_runtime.hip", "svmc_kernels.hip", "svmc_analytic.hip", "svmc_chain.hip", "svmc_comm.hip", "svmc_multi.hip")
HEADERS = ("svmc_internal.h", "svmc_models.h", "svmc_rng.h", "svmc_math.h", "svmc_log_table.h", "svmc_icdf_table.h", "svmc_black.h", "svmc_ode.h", "svmc_dop853.h")
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
            "-Wno-unused-function", "-Wno-pass-failed"]


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(INCLUDE, "svmc.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def file_sha256(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def kernel_metadata(asm_dir: str) -> dict:
    """{kernel symbol: {vgpr, sgpr, scratch_bytes, lds_bytes}} of every kernel of this build, read off the amdhsa metadata the
    compiler appends to its assembly -- a spill (scratch_bytes > 0) is visible in the build, not only in a profile"""
    import glob
    import re
    out = {}
    for path in sorted(glob.glob(os.path.join(asm_dir, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
        text = open(path).read()
        start = text.find("amdhsa.kernels:")
        if start < 0:
            continue
        for block in re.split(r"\n  - ", text[start:])[1:]:
            def field(key, block=block):
                m = re.search(r"\." + key + r":\s*(\S+)", block)
                return m.group(1) if m else None
            name = field("name")
            if name is None:
                continue
            out[name] = {"vgpr": int(field("vgpr_count") or 0), "sgpr": int(field("sgpr_count") or 0),
                         "scratch_bytes": int(field("private_segment_fixed_size") or 0),
                         "lds_bytes": int(field("group_segment_fixed_size") or 0)}
    return out


def write_isa_json(asm_path: str) -> None:
    """per-kernel instruction histogram of the time loops (tools/isa_histogram.py) + the library's hash"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import isa_histogram
    finally:
        sys.path.pop(0)
    text = open(asm_path).read()
    kernels = {}
    for name in ISA_KERNELS:
        try:
            kernels[name] = isa_histogram.analyse(text, name)
        except (SystemExit, ValueError) as exc:             # a kernel renamed away: leave it out, bench.py says so
            kernels[name] = {"error": str(exc)}
    with open(ISA_JSON, "w") as fh:
        json.dump({"lib_sha256": file_sha256(LIB), "flags": flags(), "kernels": kernels,
                   "metadata": kernel_metadata(os.path.dirname(asm_path))}, fh, indent=1)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale() and os.path.exists(ISA_JSON):
        return LIB
    # --save-temps: the device assembly of this very compilation is kept for the histogram (temporaries in a scratch dir)
    with tempfile.TemporaryDirectory(prefix="svmc_build_") as tmp:
        cmd = [hipcc()] + flags() + ["--save-temps"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
        if verbose and res.stderr:
            print(res.stderr)
        asm = os.path.join(tmp, "svmc_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
        try:
            write_isa_json(asm)
        except Exception as exc:                             # the histogram is measurement metadata, never a build failure
            if verbose:
                print("isa histogram skipped:", exc)
            if os.path.exists(ISA_JSON):
                os.remove(ISA_JSON)
            with open(ISA_JSON, "w") as fh:
                json.dump({"lib_sha256": file_sha256(LIB), "flags": flags(), "kernels": {}, "error": str(exc)}, fh)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
