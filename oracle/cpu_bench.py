"""TEST INFRASTRUCTURE ONLY -- the CPU-baseline legs `bench.py` reports beside the GPU number
("CPU baseline beside it").  Nothing under hdrnet_amd/ imports this.

Legs, all on the SAME synthetic workload as the GPU metric (BilateralSliceApply forward,
3 -> 3 channels with offset, U[0,1) inputs, seed 1234):

  reference_1thread    oracle/_ref (the reference's bilateral_slice_apply.cc compiled unchanged,
                       -O2, serial) on whole frames, ONE core.  This is `cpu_baseline.value`.
  reference_nproc      P independent processes of the same, each given whole frames (the reference
                       code is serial and a frame cannot be split by rows without changing
                       scale_y); aggregate MP/s over the wall-clock of the slowest process.
  numpy_jax            the numpy restatement of jax/bilateral_slice.py:299-380 (oracle/jax_np.py)
                       + the reference test's einsum apply (hdrnet_ops_jax_tf2_test.py), on a row
                       window of the frame; numpy threads as configured by the host.

    python -m oracle.cpu_bench --worker H W GH GW GD frames     # one process of the nproc leg
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(H, W, GH, GW, GD, seed=1234):
    rng = np.random.default_rng(seed)
    grid = rng.random((1, GH, GW, GD, 12), dtype=np.float32)
    guide = rng.random((1, H, W), dtype=np.float32)
    inp = rng.random((1, H, W, 3), dtype=np.float32)
    return grid, guide, inp


def _impl():
    from . import cpu_oracle
    if cpu_oracle.have_ref():
        return cpu_oracle.ref(), "reference"
    return cpu_oracle.port(), "port"


def worker(H, W, GH, GW, GD, frames):
    impl, kind = _impl()
    if kind == "port":
        impl.set_threads(1)
    grid, guide, inp = _inputs(H, W, GH, GW, GD, seed=1234 + os.getpid() % 1000)
    t = time.perf_counter()
    for _ in range(frames):
        impl.bilateral_slice_apply(grid, guide, inp, True)
    return time.perf_counter() - t


def leg_single(H, W, GH, GW, GD, budget_s):
    impl, kind = _impl()
    if kind == "port":
        impl.set_threads(1)
    grid, guide, inp = _inputs(H, W, GH, GW, GD)
    hp = max(8, H // 16)
    t = time.perf_counter()
    impl.bilateral_slice_apply(grid, guide[:, :hp], inp[:, :hp], True)
    per_px = (time.perf_counter() - t) / (hp * W)
    frames = int(max(1, min(8, budget_s / max(per_px * H * W, 1e-9))))
    t = time.perf_counter()
    for _ in range(frames):
        impl.bilateral_slice_apply(grid, guide, inp, True)
    dt = time.perf_counter() - t
    return {"value": round(frames * H * W / 1e6 / dt, 4), "unit": "MP/s", "cores": 1, "kind": kind,
            "sample": f"{frames} frame(s) of {W}x{H}, grid {GH}x{GW}x{GD}x12, "
                      + ("oracle/_ref (reference bilateral_slice_apply.cc compiled unchanged, -O2, serial)"
                         if kind == "reference" else "oracle C port, 1 thread"),
            "seconds": round(dt, 2)}


def leg_nproc(H, W, GH, GW, GD, procs, frames=1, timeout_s=120):
    cmd = [sys.executable, "-m", "oracle.cpu_bench", "--worker"] + [str(v) for v in (H, W, GH, GW, GD, frames)]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    t = time.perf_counter()
    ps = [subprocess.Popen(cmd, cwd=_ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
          for _ in range(procs)]
    inner = []
    for p in ps:
        try:
            out, _ = p.communicate(timeout=timeout_s)
            inner.append(float(out.strip().splitlines()[-1]))
        except Exception:  # noqa: BLE001
            p.kill()
    wall = time.perf_counter() - t
    if not inner:
        return None
    slowest = max(inner)
    return {"value": round(len(inner) * frames * H * W / 1e6 / slowest, 3), "unit": "MP/s",
            "processes": len(inner), "frames_per_process": frames,
            "sample": f"{len(inner)} processes x {frames} whole frame(s) of {W}x{H}; aggregate over the slowest "
                      f"process's compute time ({slowest:.2f} s; {wall:.2f} s incl. process start and input generation)",
            "kind": _impl()[1]}


def leg_numpy_jax(H, W, GH, GW, GD, rows=None, budget_px=1.1e6):
    from . import jax_np
    grid, guide, inp = _inputs(H, W, GH, GW, GD)
    nr = int(max(8, min(H, budget_px // W))) if rows is None else rows
    r0 = (H - nr) // 2
    g = guide[0, r0:r0 + nr]
    x = inp[0, r0:r0 + nr]
    t = time.perf_counter()
    coeffs = jax_np.bilateral_slice(grid[0], g, rows=(r0, r0 + nr, H))          # [nr, W, 12]
    co = coeffs.reshape(nr, W, 3, 4)
    out = np.einsum("hwij,hwj->hwi", co[..., :3], x, dtype=np.float32) + co[..., 3]
    dt = time.perf_counter() - t
    try:
        import threadpoolctl
        nthreads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:  # noqa: BLE001
        nthreads = None
    assert out.shape == (nr, W, 3)
    return {"value": round(nr * W / 1e6 / dt, 3), "unit": "MP/s", "numpy_threads": nthreads,
            "sample": f"rows {r0}..{r0 + nr - 1} of one {W}x{H} frame ({nr * W / 1e6:.2f} MP), "
                      "oracle/jax_np.bilateral_slice (numpy restatement of jax/bilateral_slice.py) + einsum apply",
            "seconds": round(dt, 2)}


def run(H, W, GH, GW, GD, budget_s=8.0, procs=None):
    """All legs; the single-thread reference figure stays the headline (rounds stay comparable)."""
    res = leg_single(H, W, GH, GW, GD, budget_s)
    ncpu = os.cpu_count() or 1
    res["host_cpus"] = ncpu
    legs = {}
    try:
        # capped at 64 whole-image processes: each takes ~16 s of CPU and ~0.7 GB at 4K; 256 of them would not
        # fit the "default bench.py finishes within minutes" budget.  The cap is stated in the leg.
        cap = 64
        p = procs or max(1, min(ncpu, cap))
        legs["reference_nproc"] = leg_nproc(H, W, GH, GW, GD, p)
        legs["reference_nproc"]["cap"] = (f"{p} of {ncpu} host CPUs used (capped at {cap} processes to bound run time "
                                          "and memory)" if p < ncpu else f"all {ncpu} host CPUs used")
    except Exception as e:  # noqa: BLE001
        legs["reference_nproc"] = {"error": str(e)}
    try:
        legs["numpy_jax"] = leg_numpy_jax(H, W, GH, GW, GD)
    except Exception as e:  # noqa: BLE001
        legs["numpy_jax"] = {"error": str(e)}
    res["legs"] = legs
    return res


if __name__ == "__main__":
    if len(sys.argv) >= 8 and sys.argv[1] == "--worker":
        print(worker(*[int(v) for v in sys.argv[2:8]]))
    else:
        print(json.dumps(run(2160, 3840, 16, 16, 8), indent=1))
