"""TEST INFRASTRUCTURE ONLY -- ctypes front-ends of the two CPU checkers.

``port()`` -> liboracle.so (C99 restatement), ``ref()`` -> _ref/libhdrnet_ref.so
(the reference's own .cc files compiled unchanged).  Both expose the same seven
functions of the path (reference: hdrnet/ops/bilateral_slice_apply.cc:24-259,
hdrnet/ops/bilateral_slice.cc:25-168) over NHWC float32 numpy arrays.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT_SO = os.path.join(_HERE, "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libhdrnet_ref.so")
_FP = ctypes.POINTER(ctypes.c_float)


def build(verbose: bool = False) -> None:
    """Run oracle/Makefile (liboracle.so always; _ref when /root/reference exists)."""
    res = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout + res.stderr)
    if res.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + res.stdout + res.stderr)


def _stale(so: str, *srcs: str) -> bool:
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in srcs)


def have_ref() -> bool:
    return os.path.exists(_REF_SO)


def _f32(a, name: str) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError(f"{name} must be C-contiguous")
    return a


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_FP)


class Oracle:
    """Same call surface for the port (prefix ``oracle_``) and the reference build
    (prefix ``ref_``)."""

    def __init__(self, lib: ctypes.CDLL, kind: str):
        self.lib = lib
        self.kind = kind  # "port" | "reference"

    # ---- threads (port only; the reference code is serial) -------------------
    def set_threads(self, n: int) -> int:
        if self.kind == "port":
            self.lib.oracle_set_threads(ctypes.c_int(n))
            return int(self.lib.oracle_max_threads())
        return 1

    # ---- helpers --------------------------------------------------------------
    @staticmethod
    def _apply_dims(grid, guide, inp, has_offset) -> Tuple[int, ...]:
        if grid.ndim != 5 or guide.ndim != 3 or inp.ndim != 4:
            raise ValueError("grid must be 5-D, guide 3-D, input 4-D")
        B, GH, GW, GD, C = grid.shape
        _, H, W = guide.shape
        Cin = inp.shape[3]
        Cj = Cin + (1 if has_offset else 0)
        if inp.shape[:3] != guide.shape or guide.shape[0] != B:
            raise ValueError("Input and guide size should match / batch sizes should match.")
        if C % Cj:
            raise ValueError("grid channels not divisible by input channels (+1)")
        return B, H, W, GH, GW, GD, Cin, C // Cj

    @staticmethod
    def _slice_dims(grid, guide) -> Tuple[int, ...]:
        if grid.ndim != 5 or guide.ndim != 3:
            raise ValueError("grid must be 5-D, guide 3-D")
        B, GH, GW, GD, C = grid.shape
        _, H, W = guide.shape
        if guide.shape[0] != B:
            raise ValueError("batch sizes should match")
        return B, H, W, GH, GW, GD, C

    # ---- BilateralSliceApply ----------------------------------------------------
    def bilateral_slice_apply(self, grid, guide, inp, has_offset: bool = True) -> np.ndarray:
        grid, guide, inp = _f32(grid, "grid"), _f32(guide, "guide"), _f32(inp, "input")
        B, H, W, GH, GW, GD, Cin, Cout = self._apply_dims(grid, guide, inp, has_offset)
        out = np.empty((B, H, W, Cout), np.float32)
        if out.size == 0:
            return out
        fn = self.lib.oracle_bilateral_slice_apply if self.kind == "port" else self.lib.ref_bilateral_slice_apply
        fn(_p(grid), _p(guide), _p(inp), _p(out), B, H, W, GH, GW, GD, Cin, Cout, int(bool(has_offset)))
        return out

    def bilateral_slice_apply_rows(self, grid, guide_rows, inp_rows, frame_height: int, y0: int,
                                   has_offset: bool = True) -> np.ndarray:
        """The forward on rows y0 .. y0 + rows - 1 of frames `frame_height` rows high (port only):
        the checker for hdrnet_bilateral_slice_apply_rows_f32."""
        if self.kind != "port":
            raise NotImplementedError("the reference's op has no row-split form")
        grid, guide_rows, inp_rows = _f32(grid, "grid"), _f32(guide_rows, "guide"), _f32(inp_rows, "input")
        B, rows, W, GH, GW, GD, Cin, Cout = self._apply_dims(grid, guide_rows, inp_rows, has_offset)
        if y0 < 0 or y0 + rows > frame_height:
            raise ValueError("row band outside the frame")
        out = np.empty((B, rows, W, Cout), np.float32)
        if out.size:
            self.lib.oracle_bilateral_slice_apply_rows(_p(grid), _p(guide_rows), _p(inp_rows), _p(out), B,
                                                       int(frame_height), int(y0), rows, W, GH, GW, GD, Cin,
                                                       Cout, int(has_offset))
        return out

    def bilateral_slice_apply_grad(self, grid, guide, inp, dout, has_offset: bool = True,
                                   want=("grid", "guide", "input")):
        grid, guide, inp, dout = (_f32(grid, "grid"), _f32(guide, "guide"),
                                  _f32(inp, "input"), _f32(dout, "dout"))
        B, H, W, GH, GW, GD, Cin, Cout = self._apply_dims(grid, guide, inp, has_offset)
        if dout.shape != (B, H, W, Cout):
            raise ValueError("dout shape mismatch")
        ho = int(bool(has_offset))
        dgrid = np.zeros_like(grid) if "grid" in want else None
        dguide = np.zeros_like(guide) if "guide" in want else None
        dinput = np.zeros_like(inp) if "input" in want else None
        if guide.size:
            if self.kind == "port":
                L = self.lib
                if dgrid is not None:
                    L.oracle_bilateral_slice_apply_grid_grad(_p(guide), _p(inp), _p(dout), _p(dgrid),
                                                             B, H, W, GH, GW, GD, Cin, Cout, ho)
                if dguide is not None:
                    L.oracle_bilateral_slice_apply_guide_grad(_p(grid), _p(guide), _p(inp), _p(dout), _p(dguide),
                                                              B, H, W, GH, GW, GD, Cin, Cout, ho)
                if dinput is not None:
                    L.oracle_bilateral_slice_apply_input_grad(_p(grid), _p(guide), _p(dout), _p(dinput),
                                                              B, H, W, GH, GW, GD, Cin, Cout, ho)
            else:
                self.lib.ref_bilateral_slice_apply_grad(_p(grid), _p(guide), _p(inp), _p(dout),
                                                        _p(dgrid), _p(dguide), _p(dinput),
                                                        B, H, W, GH, GW, GD, Cin, Cout, ho)
        return dgrid, dguide, dinput

    # ---- BilateralSlice ---------------------------------------------------------
    def bilateral_slice(self, grid, guide) -> np.ndarray:
        grid, guide = _f32(grid, "grid"), _f32(guide, "guide")
        B, H, W, GH, GW, GD, C = self._slice_dims(grid, guide)
        out = np.empty((B, H, W, C), np.float32)
        if out.size == 0:
            return out
        fn = self.lib.oracle_bilateral_slice if self.kind == "port" else self.lib.ref_bilateral_slice
        fn(_p(grid), _p(guide), _p(out), B, H, W, GH, GW, GD, C)
        return out

    def bilateral_slice_grad(self, grid, guide, dout, want=("grid", "guide")):
        grid, guide, dout = _f32(grid, "grid"), _f32(guide, "guide"), _f32(dout, "dout")
        B, H, W, GH, GW, GD, C = self._slice_dims(grid, guide)
        if dout.shape != (B, H, W, C):
            raise ValueError("dout shape mismatch")
        dgrid = np.zeros_like(grid) if "grid" in want else None
        dguide = np.zeros_like(guide) if "guide" in want else None
        if guide.size:
            if self.kind == "port":
                if dgrid is not None:
                    self.lib.oracle_bilateral_slice_grid_grad(_p(guide), _p(dout), _p(dgrid),
                                                              B, H, W, GH, GW, GD, C)
                if dguide is not None:
                    self.lib.oracle_bilateral_slice_guide_grad(_p(grid), _p(guide), _p(dout), _p(dguide),
                                                               B, H, W, GH, GW, GD, C)
            else:
                self.lib.ref_bilateral_slice_grad(_p(grid), _p(guide), _p(dout), _p(dgrid), _p(dguide),
                                                  B, H, W, GH, GW, GD, C)
        return dgrid, dguide


_port: Optional[Oracle] = None
_ref: Optional[Oracle] = None


def port() -> Oracle:
    """The C99 restatement (builds it on first use if missing/stale)."""
    global _port
    if _port is None:
        if _stale(_PORT_SO, os.path.join(_HERE, "bilateral_oracle.c")):
            build()
        lib = ctypes.CDLL(_PORT_SO)
        lib.oracle_max_threads.restype = ctypes.c_int
        _port = Oracle(lib, "port")
    return _port


def ref() -> Oracle:
    """The reference's own CPU code (prebuilt, or built here if /root/reference exists)."""
    global _ref
    if _ref is None:
        if not os.path.exists(_REF_SO) and os.path.isdir("/root/reference/hdrnet/ops"):
            build()
        if not os.path.exists(_REF_SO):
            raise FileNotFoundError(
                f"{_REF_SO} missing and /root/reference not available to build it")
        _ref = Oracle(ctypes.CDLL(_REF_SO), "reference")
    return _ref
