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


def test_new_entry_points_validate_without_gpu(libpath):
    """Argument validation of the guide-network / pyramid / resize entry points happens before any
    HIP call, like the four ops': error codes and messages are testable on a CPU box."""
    lib = ctypes.CDLL(libpath)
    lib.hdrnet_last_error.restype = ctypes.c_char_p
    LL, I, P, SZ = ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    lib.hdrnet_pointwise_guide_grad_workspace_bytes.restype = SZ
    lib.hdrnet_pointwise_guide_grad_workspace_bytes.argtypes = [LL, I, I]
    lib.hdrnet_input_moments_workspace_bytes.restype = SZ
    lib.hdrnet_input_moments_workspace_bytes.argtypes = [LL, I]
    # unsupported widths report "no workspace" = unsupported, never a bogus size
    assert lib.hdrnet_pointwise_guide_grad_workspace_bytes(1000, 3, 5) == 0
    assert lib.hdrnet_pointwise_guide_grad_workspace_bytes(1000, 2, 16) == 0
    assert lib.hdrnet_input_moments_workspace_bytes(1000, 2) == 0
    assert lib.hdrnet_input_moments_workspace_bytes(0, 3) == 0
    lib.hdrnet_pointwise_guide_grad_f32.argtypes = [P] * 6 + [I] + [P] * 2 + [LL, I, I, P, SZ, P]
    rc = lib.hdrnet_pointwise_guide_grad_f32(None, None, None, None, None, None, 0, None, None, 16, 3, 16, None, 0, None)
    assert rc == 1 and b"null buffer" in lib.hdrnet_last_error()
    rc = lib.hdrnet_pointwise_guide_grad_f32(None, None, None, None, None, None, 0, None, None, -1, 3, 16, None, 0, None)
    assert rc == 1 and b"bad sizes" in lib.hdrnet_last_error()
    lib.hdrnet_input_moments_f32.argtypes = [P, LL, I, P, P, P, SZ, P]
    assert lib.hdrnet_input_moments_f32(None, 16, 2, None, None, None, 0, None) == 1
    assert b"Cin in {1,3}" in lib.hdrnet_last_error()
    lib.hdrnet_resize_bilinear_f32.argtypes = [P, P] + [I] * 6 + [P]
    assert lib.hdrnet_resize_bilinear_f32(None, None, 1, 0, 4, 2, 2, 3, None) == 1
    assert b"bad extents" in lib.hdrnet_last_error()
    assert lib.hdrnet_resize_bilinear_f32(None, None, 1, 4, 4, 0, 2, 3, None) == 0  # empty output: no-op
    lib.hdrnet_bilateral_slice_apply_upadd_f32.argtypes = [P] * 4 + [I, I, P] + [I] * 9 + [P, P, I, P]
    rc = lib.hdrnet_bilateral_slice_apply_upadd_f32(None, None, None, None, 2, 2, None, 1, 4, 4, 2, 2, 2, 3, 3, 1,
                                                    None, None, 0, None)
    assert rc == 1 and b"either a guide map or the guide network" in lib.hdrnet_last_error()
    rc = lib.hdrnet_bilateral_slice_apply_upadd_f32(None, None, None, None, 0, 2, None, 1, 4, 4, 2, 2, 2, 3, 3, 1,
                                                    None, None, 0, None)
    assert rc == 1 and b"coarse extents" in lib.hdrnet_last_error()
    # the ..._ex twins of the guide-network entry points (round 5): the sigmoid is chosen by HDRNET_GUIDE_SIGMOID_FAST in
    # `flags` -- any other bit is refused before anything else is looked at
    U = ctypes.c_uint
    lib.hdrnet_bilateral_slice_apply_upadd_f32_ex.argtypes = [P] * 4 + [I, I, P] + [I] * 9 + [P, P, I, U, P]
    rc = lib.hdrnet_bilateral_slice_apply_upadd_f32_ex(None, None, None, None, 2, 2, None, 1, 4, 4, 2, 2, 2, 3, 3, 1,
                                                       None, None, 0, 0x40000, None)
    assert rc == 1 and b"HDRNET_GUIDE_SIGMOID_FAST" in lib.hdrnet_last_error()
    rc = lib.hdrnet_bilateral_slice_apply_upadd_f32_ex(None, None, None, None, 2, 2, None, 1, 4, 4, 2, 2, 2, 3, 3, 1,
                                                       None, None, 0, 0x10000, None)
    assert rc == 1 and b"either a guide map or the guide network" in lib.hdrnet_last_error()  # flag accepted
    lib.hdrnet_bilateral_slice_apply_nnguide_f32_ex.argtypes = [P] * 6 + [I] * 10 + [U, P]
    rc = lib.hdrnet_bilateral_slice_apply_nnguide_f32_ex(None, None, None, None, None, None, 1, 4, 4, 2, 2, 2, 3, 3, 1, 16,
                                                         0x1, None)
    assert rc == 1 and b"unknown flags" in lib.hdrnet_last_error()
    rc = lib.hdrnet_bilateral_slice_apply_nnguide_f32_ex(None, None, None, None, None, None, 1, 4, 4, 2, 2, 2, 3, 3, 1, 16,
                                                         0x10000, None)
    assert rc == 1 and b"null buffer" in lib.hdrnet_last_error()
    lib.hdrnet_bilateral_slice_apply_io_ex.argtypes = [P] * 4 + [I] * 10 + [ctypes.c_float, I] + [P] * 2 + [I, P, U, P]
    rc = lib.hdrnet_bilateral_slice_apply_io_ex(None, None, None, None, 1, 4, 4, 2, 2, 2, 3, 3, 1, 0, 1.0, 0, None, None, 0,
                                                None, 0x50000, None)
    assert rc == 1 and b"unknown flags" in lib.hdrnet_last_error()
    # HDRNET_GUIDE_RELU_PRESCALED (0x20000): the arrays of hdrnet_guide_nn_prescale_f32 -- a guide NETWORK of three input
    # channels with 16-B aligned arrays, or the call is refused; the helper itself validates before it launches
    F = ctypes.c_float
    rc = lib.hdrnet_bilateral_slice_apply_nnguide_f32_ex(0x1000, 0x1000, 0x1000, 0x1000, 0x1000, None, 1, 4, 4, 2, 2, 2, 1, 1,
                                                         1, 16, 0x20000, None)
    assert rc == 1 and b"HDRNET_GUIDE_RELU_PRESCALED needs a guide network with Cin = 3" in lib.hdrnet_last_error()
    rc = lib.hdrnet_bilateral_slice_apply_nnguide_f32_ex(0x1000, 0x1000, 0x1004, 0x1000, 0x1000, None, 1, 4, 4, 2, 2, 2, 3, 3,
                                                         1, 16, 0x30000, None)
    assert rc == 1 and b"16-B aligned guide_conv1" in lib.hdrnet_last_error()
    rc = lib.hdrnet_bilateral_slice_apply_io_ex(0x1000, 0x1000, 0x1000, 0x1000, 1, 4, 4, 2, 2, 2, 3, 3, 1, 0, 1.0, 0, None,
                                                None, 0, None, 0x20000, None)  # a guide MAP: nothing to prescale
    assert rc == 1 and b"HDRNET_GUIDE_RELU_PRESCALED" in lib.hdrnet_last_error()
    lib.hdrnet_guide_nn_prescale_f32.argtypes = [P, P, I, I, F, P, P, P]
    assert lib.hdrnet_guide_nn_prescale_f32(0x1000, 0x1000, 16, 1, 1.0, 0x1000, 0x1000, None) == 1
    assert b"Cin = 3" in lib.hdrnet_last_error()
    assert lib.hdrnet_guide_nn_prescale_f32(0x1000, 0x1000, 16, 3, 0.0, 0x1000, 0x1000, None) == 1
    assert b"x_max" in lib.hdrnet_last_error()
    assert lib.hdrnet_guide_nn_prescale_f32(0x1000, 0x1000, 16, 3, 1.0, 0x1004, 0x1000, None) == 1
    assert b"16-B aligned output" in lib.hdrnet_last_error()
    assert lib.hdrnet_guide_nn_prescale_f32(None, 0x1000, 16, 3, 1.0, 0x1000, 0x1000, None) == 1
    assert b"null buffer" in lib.hdrnet_last_error()


def test_curves_entry_points_validate_without_gpu(libpath):
    lib = ctypes.CDLL(libpath)
    lib.hdrnet_last_error.restype = ctypes.c_char_p
    LL, I, P, SZ, F = ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_float
    lib.hdrnet_curves_guide_grad_workspace_bytes.restype = SZ
    lib.hdrnet_curves_guide_grad_workspace_bytes.argtypes = [LL, I, I]
    assert lib.hdrnet_curves_guide_grad_workspace_bytes(1000, 3, 8) == 0   # the reference has 16 knots
    assert lib.hdrnet_curves_guide_grad_workspace_bytes(1000, 1, 16) == 0
    lib.hdrnet_curves_guide_grad_f32.argtypes = [P] * 7 + [I] + [P] * 4 + [LL, I, I, P, SZ, P]
    rc = lib.hdrnet_curves_guide_grad_f32(*([None] * 7), 0, *([None] * 4), 10, 3, 16, None, 0, None)
    assert rc == 1 and b"null buffer" in lib.hdrnet_last_error()
    lib.hdrnet_bilateral_slice_apply_io_curves.argtypes = [P] * 3 + [I] * 10 + [F, I] + [P] * 4 + [I, P, P]
    rc = lib.hdrnet_bilateral_slice_apply_io_curves(None, None, None, 1, 4, 4, 2, 2, 2, 3, 3, 1, 0, 1.0, 0,
                                                    None, None, None, None, 0, None, None)
    assert rc == 1 and b"curve knots" in lib.hdrnet_last_error()
    rc = lib.hdrnet_bilateral_slice_apply_io_curves(None, None, None, 1, 4, 4, 2, 2, 2, 3, 3, 1, 5, 1.0, 0,
                                                    None, None, None, None, 16, None, None)
    assert rc == 1 and b"dtype" in lib.hdrnet_last_error()
    rc = lib.hdrnet_bilateral_slice_apply_io_curves(None, None, None, 1, 4, 4, 2, 2, 2, 3, 3, 1, 0, 1.0, 0,
                                                    None, None, None, None, 16, None, None)
    assert rc == 1 and b"null buffer" in lib.hdrnet_last_error()
    # the prepared cell tables (round 5): size query, the prepare entry point's checks, the forward's check of the buffer
    lib.hdrnet_curves_guide_prepared_bytes.restype = SZ
    lib.hdrnet_curves_guide_prepared_bytes.argtypes = [I]
    assert lib.hdrnet_curves_guide_prepared_bytes(3) == (3 * 64 * 4 + 3 * 4 + 4) * 4
    assert lib.hdrnet_curves_guide_prepared_bytes(1) == 0
    lib.hdrnet_curves_guide_prepare_f32.argtypes = [P, P, I, I, P, SZ, ctypes.POINTER(I), P]
    us = I(7)
    assert lib.hdrnet_curves_guide_prepare_f32(0x1000, 0x1000, 17, 3, 0x1000, 1 << 20, ctypes.byref(us), None) == 1
    assert b"1 .. 16 knots" in lib.hdrnet_last_error()
    assert lib.hdrnet_curves_guide_prepare_f32(0x1000, 0x1000, 16, 1, 0x1000, 1 << 20, ctypes.byref(us), None) == 1
    assert lib.hdrnet_curves_guide_prepare_f32(None, 0x1000, 16, 3, 0x1000, 1 << 20, ctypes.byref(us), None) == 1
    assert b"null buffer" in lib.hdrnet_last_error()
    assert lib.hdrnet_curves_guide_prepare_f32(0x1000, 0x1000, 16, 3, 0x1000, 64, ctypes.byref(us), None) == 1
    assert b"hdrnet_curves_guide_prepared_bytes" in lib.hdrnet_last_error()
    assert lib.hdrnet_curves_guide_prepare_f32(0x1000, 0x1000, 16, 3, 0x1004, 1 << 20, ctypes.byref(us), None) == 1
    assert lib.hdrnet_curves_guide_prepare_f32(0x1000, 0x1000, 16, 3, 0x1000, 1 << 20, None, None) == 1  # where to report?
    assert b"usable" in lib.hdrnet_last_error()
    lib.hdrnet_bilateral_slice_apply_io_curves_prepared.argtypes = [P] * 3 + [I] * 10 + [F, I] + [P] * 4 + [I, P, P, P]
    rc = lib.hdrnet_bilateral_slice_apply_io_curves_prepared(0x1000, 0x1000, 0x1000, 1, 4, 4, 2, 2, 2, 3, 3, 1, 0, 1.0, 0,
                                                             0x1000, 0x1000, 0x1000, 0x1000, 16, 0x1004, None, None)
    assert rc == 1 and b"prepared curves tables" in lib.hdrnet_last_error()
    rc = lib.hdrnet_bilateral_slice_apply_io_curves_prepared(0x1000, 0x1000, 0x1000, 1, 4, 4, 2, 2, 2, 3, 3, 1, 0, 1.0, 0,
                                                             0x1000, 0x1000, 0x1000, 0x1000, 20, 0x1000, None, None)
    assert rc == 1 and b"prepared curves tables" in lib.hdrnet_last_error()
