"""The C-ABI library builds for gfx950 (hipcc cross-compiles without a GPU), loads, and
exports every symbol include/hdrnet_amd.h declares.  No compute call is made here."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "hdrnet_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(hdrnet_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def libpath():
    from hdrnet_amd import build
    return build.build()


def test_header_declares_the_four_ops():
    names = declared_functions()
    for n in ("hdrnet_bilateral_slice_apply_f32", "hdrnet_bilateral_slice_apply_grad_f32",
              "hdrnet_bilateral_slice_f32", "hdrnet_bilateral_slice_grad_f32"):
        assert n in names


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    for n in declared_functions():
        assert hasattr(lib, n), f"{n} declared in include/hdrnet_amd.h but not exported"


def test_binding_table_matches_header(libpath):
    from hdrnet_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    # argument counts of the binding table == parameter counts in the header
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^)]*)\)", src)
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else len(params.split(","))
        assert n == len(args), (name, n, len(args))


def test_library_is_gfx950_only_and_in_tree(libpath):
    assert os.path.commonpath([ROOT, libpath]) == ROOT
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={libpath}"], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets
    else:
        blob = open(libpath, "rb").read()
        assert b"gfx950" in blob
        for other in (b"gfx942", b"gfx90a", b"sm_80"):
            assert other not in blob


def test_version_and_error_text_without_gpu(libpath):
    lib = ctypes.CDLL(libpath)
    lib.hdrnet_version.restype = ctypes.c_int
    lib.hdrnet_last_error.restype = ctypes.c_char_p
    assert lib.hdrnet_version() >= 100
    # Argument validation happens before any HIP call: safe without a GPU.
    rc = lib.hdrnet_bilateral_slice_apply_f32(None, None, None, None, 1, 4, 4, 0, 2, 2, 3, 3, 1, None)
    assert rc == 1
    assert b"grid extents" in lib.hdrnet_last_error()
    rc = lib.hdrnet_bilateral_slice_f32(None, None, None, 1, 4, 4, 2, 2, 2, 0, None)
    assert rc == 1
    # Zero-sized image: legal no-op (the reference loops over an empty shape).
    assert lib.hdrnet_bilateral_slice_apply_f32(None, None, None, None, 0, 4, 4, 2, 2, 2, 3, 3, 1, None) == 0
