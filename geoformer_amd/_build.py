"""Build helpers: compile the HIP sources under ``csrc/`` into one C-ABI shared library.

The library (``geoformer_amd/lib/libgeoformer_hip.so``) is built in-tree with ``hipcc
--offload-arch=gfx950`` so it travels to the GPU box with the repo snapshot.  No torch
types cross this boundary: the entry points are plain ``extern "C"`` functions declared
in ``include/geoformer_hip.h``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libgeoformer_hip.so")
INCLUDE_DIR = os.path.join(REPO_DIR, "include")

ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm toolchain to build libgeoformer_hip.so)")


def _sources():
    return sorted(
        os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith(".hip")
    )


def _headers():
    hs = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith((".h", ".hpp"))]
    hs += [os.path.join(INCLUDE_DIR, f) for f in os.listdir(INCLUDE_DIR) if f.endswith(".h")]
    return hs


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """Compile every ``csrc/*.hip`` for gfx950 and link ``libgeoformer_hip.so``.

    Incremental: an object is rebuilt only when its source or any header is newer.
    Returns the path of the shared library.
    """
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = _headers()
    srcs = _sources()
    flags = [
        f"--offload-arch={ARCH}",
        "-O3",
        "-fPIC",
        "-std=c++17",
        "-ffp-contract=off",  # every fused multiply-add in the kernels is an explicit fmaf()
        "-fno-gpu-rdc",
        "-Wno-unused-result",
        f"-I{INCLUDE_DIR}",
        f"-I{CSRC_DIR}",
    ]
    # per-file extra device flags (none at present).  ROCm 7.2's gfx950 backend was seen folding a chain of
    # "bits = f & open; open &= ~bits" updates into v_bitop3_b32 with wrong truth tables (round 5, in a kernel that is no
    # longer in the tree: HISTORY.md 7); `check_bitop3()` below lists the kernels whose ISA holds the instruction.
    extra = {}
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def _cc(job):
        s, o = job
        cmd = [hipcc, *flags, *extra.get(os.path.basename(s), []), "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(_cc, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
