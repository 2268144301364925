"""Builds libhdrnet_amd.so (the C-ABI library, include/hdrnet_amd.h) for gfx950.

hipcc cross-compiles without a GPU.  The library is built IN-TREE
(hdrnet_amd/lib/libhdrnet_amd.so) so that it travels with the repository snapshot to
the GPU box and shows up as a loaded in-tree .so of the test processes.

    python -m hdrnet_amd.build [--force] [--verbose]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libhdrnet_amd.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]

# (source, extra flags).  generic_kernels.hip is the bit-exact path: no FMA contraction.
SOURCES = [
    ("capi.hip", []),
    ("generic_kernels.hip", ["-ffp-contract=off"]),
    # The blend is written on explicit 2-wide vectors (v_pk_fma_f32); LLVM's SLP pass on top
    # of that pairs unrelated scalars ACROSS pixels and pays for it in v_mov shuffles
    # (measured: 947 -> 646 ISA lines, 82 -> 55 VGPRs with it off).
    ("apply_fwd_rows.hip", ["-fno-slp-vectorize"]),
    ("apply_fwd_variants.hip", ["-fno-slp-vectorize"]),
    ("apply_fwd_io.hip", ["-fno-slp-vectorize"]),
    ("apply_bwd_rows.hip", ["-fno-slp-vectorize"]),
    ("slice_fwd_rows.hip", ["-fno-slp-vectorize"]),
    ("grid_grad_mfma.hip", ["-fno-slp-vectorize"]),
    ("guide_grad.hip", ["-fno-slp-vectorize"]),
    ("resize_bilinear.hip", []),
]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _deps() -> List[str]:
    out = [os.path.join(INCLUDE, "hdrnet_amd.h"), os.path.abspath(__file__)]
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h")):
            out.append(os.path.join(CSRC, f))
    return out


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def _run(cmd: List[str], verbose: bool) -> None:
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("command failed: %s\n%s%s" % (" ".join(cmd), res.stdout, res.stderr))
    if verbose and (res.stdout or res.stderr):
        print(res.stdout + res.stderr, flush=True)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link libhdrnet_amd.so.  Returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    cc = hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    objdir = os.path.join(LIB_DIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src, extra in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [cc, *COMMON, *extra, "-I", CSRC, "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out:
            print(out, flush=True)
    tmp = LIB_PATH + ".tmp"
    _run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp, *objs], verbose)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(path)
