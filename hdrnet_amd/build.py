"""Builds libhdrnet_amd.so (the C-ABI library, include/hdrnet_amd.h) for gfx950.

hipcc cross-compiles without a GPU.  The library is built IN-TREE
(hdrnet_amd/lib/libhdrnet_amd.so) so that it travels with the repository snapshot to
the GPU box and shows up as a loaded in-tree .so of the test processes.

    python -m hdrnet_amd.build [--force] [--verbose] [--tools]
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
# Tools build (include/hdrnet_amd_tools.h): the same sources with -DHDRNET_TOOLS_BUILD plus the
# benchmark-only kernel variants / memory skeletons.  Never loaded by the product path.
TOOLS_LIB_PATH = os.path.join(LIB_DIR, "libhdrnet_amd_tools.so")
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
    ("apply_fwd_seg.hip", ["-fno-slp-vectorize"]),
    # + MFMA results in VGPRs (gfx950's register file is unified): the guide network's 4x4x4 blocks are consumed by
    # the VALU at once, and the AGPR form costs a v_accvgpr_read per element
    ("apply_fwd_io.hip", ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]),
    ("apply_bwd_rows.hip", ["-fno-slp-vectorize"]),
    ("apply_vjp_seg.hip", ["-fno-slp-vectorize"]),
    ("slice_fwd_rows.hip", ["-fno-slp-vectorize"]),
    ("grid_grad_mfma.hip", ["-fno-slp-vectorize"]),
    ("guide_grad.hip", ["-fno-slp-vectorize"]),
    ("resize_bilinear.hip", []),
    ("coeff_net.hip", []),
    ("coeff_net_train.hip", []),
    ("metrics.hip", []),
]
TOOLS_ONLY_SOURCES = [
    ("apply_fwd_variants.hip", ["-fno-slp-vectorize"]),
    ("pyramid_onepass.hip", ["-fno-slp-vectorize"]),
]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


_FLAG_OK: dict = {}


def _flag_supported(cc: str, flags: List[str]) -> bool:
    """Does this hipcc accept `flags`?  (-mllvm options are hidden LLVM knobs: an older or newer ROCm LLVM that does not
    know one rejects the whole compile.  Probed once per process on an empty translation unit.)"""
    key = tuple(flags)
    if key not in _FLAG_OK:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            with open(src, "w") as fh:
                fh.write("__global__ void probe() {}\n")
            res = subprocess.run([cc, f"--offload-arch={ARCH}", "-c", src, "-o", os.path.join(d, "probe.o"), *flags],
                                 capture_output=True, text=True)
            _FLAG_OK[key] = res.returncode == 0
    return _FLAG_OK[key]


# flags a compile can do without (tuning only): dropped when the compiler does not know them
OPTIONAL_FLAG_GROUPS = [["-mllvm", "-amdgpu-mfma-vgpr-form"]]


def _usable_flags(cc: str, extra: List[str]) -> List[str]:
    out = list(extra)
    for group in OPTIONAL_FLAG_GROUPS:
        n = len(group)
        for i in range(len(out) - n + 1):
            if out[i:i + n] == group and not _flag_supported(cc, group):
                del out[i:i + n]
                break
    return out


def _deps() -> List[str]:
    out = [os.path.join(INCLUDE, "hdrnet_amd.h"), os.path.abspath(__file__)]
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h")):
            out.append(os.path.join(CSRC, f))
    return out


def source_digest(tools: bool = False) -> str:
    """sha256 over everything the library is built from: csrc/*, the public header, this file (the
    flags) and which build it is.  Written next to the .so after a successful link."""
    import hashlib

    h = hashlib.sha256(b"tools" if tools else b"product")
    for d in sorted(_deps()):
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _digest_path(tools: bool) -> str:
    return (TOOLS_LIB_PATH if tools else LIB_PATH) + ".digest"


def is_stale(tools: bool = False) -> bool:
    """True unless the .so exists AND was built from exactly the sources in the tree -- decided from a
    content digest, not from mtimes (a git checkout or a copy without -t perturbs those)."""
    path = TOOLS_LIB_PATH if tools else LIB_PATH
    if not os.path.exists(path):
        return True
    try:
        with open(_digest_path(tools)) as fh:
            return fh.read().strip() != source_digest(tools)
    except OSError:
        return True


def _run(cmd: List[str], verbose: bool) -> None:
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("command failed: %s\n%s%s" % (" ".join(cmd), res.stdout, res.stderr))
    if verbose and (res.stdout or res.stderr):
        print(res.stdout + res.stderr, flush=True)


def _sweep_stale(objdir: str) -> None:
    """Remove what a KILLED build left behind (the `finally` below never ran for it): per-process objects
    `*.o.<pid>` in `objdir` and `*.tmp` link outputs / digests beside the library.  Called with the build lock held,
    so nothing found here belongs to a live build."""
    for d, pred in ((objdir, lambda f: ".o." in f), (LIB_DIR, lambda f: f.endswith(".tmp") or ".digest." in f)):
        try:
            names = os.listdir(d)
        except OSError:
            continue
        for f in names:
            path = os.path.join(d, f)
            if pred(f) and os.path.isfile(path):
                try:
                    os.remove(path)
                except OSError:
                    pass


def build(force: bool = False, verbose: bool = False, tools: bool = False) -> str:
    """Compile every HIP source for gfx950 and link libhdrnet_amd.so (tools=True:
    libhdrnet_amd_tools.so).  Returns its path.

    Safe against concurrent callers (every rank of `bench.py --gpus N` loads the library): the
    build runs under an exclusive file lock, staleness is re-checked once the lock is held, and
    objects / the link output go to per-process temporary names before an atomic rename."""
    lib_path = TOOLS_LIB_PATH if tools else LIB_PATH
    if not force and not is_stale(tools):
        return lib_path
    cc = hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    import fcntl

    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale(tools):  # another process built it while we waited
                return lib_path
            digest = source_digest(tools)  # of the sources as they are read now
            objdir = os.path.join(LIB_DIR, "obj_tools" if tools else "obj")
            os.makedirs(objdir, exist_ok=True)
            _sweep_stale(objdir)
            tag = ".%d" % os.getpid()
            define = ["-DHDRNET_TOOLS_BUILD"] if tools else []
            objs = []
            procs = []
            for src, extra in SOURCES + (TOOLS_ONLY_SOURCES if tools else []):
                obj = os.path.join(objdir, src.replace(".hip", ".o") + tag)
                cmd = [cc, *COMMON, *define, *_usable_flags(cc, extra), "-I", CSRC, "-I", INCLUDE, "-c",
                       os.path.join(CSRC, src), "-o", obj]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
                objs.append(obj)
            try:
                for cmd, p in procs:
                    out, _ = p.communicate()
                    if p.returncode != 0:
                        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), out))
                    if verbose and out:
                        print(out, flush=True)
                tmp = lib_path + tag + ".tmp"
                _run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp, *objs], verbose)
                os.replace(tmp, lib_path)
                with open(_digest_path(tools) + tag, "w") as fh:
                    fh.write(digest + "\n")
                os.replace(_digest_path(tools) + tag, _digest_path(tools))
            finally:
                for o in objs:
                    if os.path.exists(o):
                        os.remove(o)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib_path


if __name__ == "__main__":
    for _tools in ([False, True] if "--tools" in sys.argv else [False]):
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv,
                    tools=_tools))
