#!/usr/bin/env python3
"""bench.py -- BilateralSliceApply forward throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k|1080p|1080p_b4|hdrp] [--split rows]
    python bench.py --workload train_1080p_b4 [--gpus N]    BASELINE config #4 as stated: training step, 4 x 1080p per GPU,
                                                            ONE flat-bucket gradient all-reduce over the N ranks
    python bench.py --workload hdrp_u16 [--gpus N]          BASELINE config #5 as stated: uint16 / 32767 wire format,
                                                            4000x3000, grid 32x32x8x12, one image per GPU

`--gpus N` with N > 1 from a bare shell re-launches itself as N ranks under
torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1); launched by the driver under
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.

Metric (BASELINE.json): megapixels/s of BilateralSliceApply forward @4K (3840x2160 fp32 NHWC,
grid 16x16x8x12, batch 1), reported with the fraction of the MI355X HBM roofline; for N > 1
each rank (one process per GPU, launched by torch.distributed.run) processes its OWN frames
-- the path shards by image with no data-path collective -- so scaling is
"weak" and `value` is the whole-job aggregate.

A "step" is one pass of the hot path over one frame: ONE kernel launch through the C-ABI
(hdrnet_bilateral_slice_apply_f32) on the current HIP stream.  Inputs are resident in HBM
before the timed region.  Steps rotate over enough independent buffer sets that the
working set exceeds the 256 MiB Infinity Cache + L2 (232 MB per 4K frame x 3 sets), so the
number is an HBM number, not a cache number; the cache-resident rate is reported separately
under "extra".

Protocol (round 2's short event-timed default is gone): an untimed, disclosed PRE-ROLL (>= 250 ms of
launches, `preroll_launches` in the JSON) puts the device into its SUSTAINED state -- the clock ramp from
idle takes ~25 ms, and on many boxes the first slow episode of the power manager arrives 50-100 ms after
the load starts (profiles/r02/exp35; with a 50-ms pre-roll a K = 20 window sat right on that edge and read
38.6 or 50 us depending on a few hundred microseconds of host timing, profiles/r03/k20_*.txt) -- then W
warm-up launches, then EXACTLY K launches timed on the WALL clock between barrier +
torch.cuda.synchronize() pairs -- always the wall clock, whatever K is; `value` = MP/s over that region,
max over ranks.  Defaults K = 2000, W = 1000.  HIP events on the launch stream bracket the same K launches and give the
`roofline` block its average kernel duration.  Under SUSTAINED load some boxes alternate between a fast
state and a ~17 % slower one in episodes of 50-200 ms (profiles/r02/exp35): the `sustained` block
(N = 1) reports the mean and the spread of 100-launch windows over a further 0.5 s, and
`roofline.frac_sustained` is the roofline fraction at that mean; its `power` entry is the socket power and shader
clock read beside it (the kernel runs at the package power cap).  The `pipelined` block (N = 1) is the same frames
issued round-robin on TWO streams -- the next frame's pipeline fill under the previous frame's drain, what a
pipeline of independent frames gets (hdrnet_amd.runtime.FramePipeline) -- reported beside the headline, never as
`value`.

Adds to the JSON line:
  roofline     -- algorithmic bytes / average kernel duration (HIP events on the launch
                  stream around the timed region) vs the 8 TB/s HBM3E peak; `traffic` = HBM bytes
                  per launch from the committed rocprofv3 PMC passes, only if they were collected
                  on the SAME kernel sources (digest of hdrnet_amd/csrc), else null
  sustained    -- 0.5 s of back-to-back launches after the timed region: mean, fastest / slowest
                  100-launch window, fraction of windows > 8 % slower than the fastest
  cpu_baseline -- the reference's own CPU op (oracle/_ref, kind "reference") on ONE core (value),
                  plus legs: nproc whole-image processes of it, and the numpy restatement of
                  jax/bilateral_slice.py; bounded samples, rank 0, N = 1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
CACHE_BYTES = 256 * 2 ** 20 + 32 * 2 ** 20  # Infinity Cache + aggregate L2

WORKLOADS = {
    # name: (B, H, W, GH, GW, GD, description)   B = images per launch (per GPU)
    "4k": (1, 2160, 3840, 16, 16, 8, "BilateralSliceApply fwd 3840x2160 fp32 NHWC, grid 16x16x8x12, batch=1/GPU"),
    "1080p": (1, 1080, 1920, 16, 16, 8, "BilateralSliceApply fwd 1920x1080 fp32 NHWC, grid 16x16x8x12, batch=1/GPU"),
    "1080p_b4": (4, 1080, 1920, 16, 16, 8,
                 "BilateralSliceApply fwd 4 x 1920x1080 fp32 NHWC per launch (config #4's per-GPU batch), grid 16x16x8x12"),
    "hdrp": (1, 3000, 4000, 32, 32, 8, "BilateralSliceApply fwd 4000x3000 fp32 NHWC, grid 32x32x8x12, 1 image/GPU"),
}


# The two multi-GPU configurations of BASELINE.json AS STATED (configs[3], configs[4]); bench lines of their own
# (main_train / main_hdrp_u16), never the headline.
EXTRA_WORKLOADS = {
    "train_1080p_b4": "training step fwd+bwd+Adam, HDRNetPointwiseNNGuide (coefficient network {bn}; the guide network's "
                      "own batch norm in training mode either way), 4 x 1920x1080 per GPU, "
                      "gradient all-reduce = ONE flat fp32 bucket (RCCL over xGMI)",
    "hdrp_u16": "BilateralSliceApply fwd, HDR+ wire format uint16 / 32767 -> fp32, 4000x3000, grid 32x32x8x12, "
                "1 image/GPU",
}


def algorithmic_bytes(B, H, W, GH, GW, GD, Cin=3, Cout=3, has_offset=True):
    """4*B*[H*W*(1 + Cin + Cout) + GH*GW*GD*Cout*Cj]."""
    Cj = Cin + int(has_offset)
    return 4 * B * (H * W * (1 + Cin + Cout) + GH * GW * GD * Cout * Cj)


# The translation units the measured forward kernel and its C-ABI entry point are compiled from; the digest follows
# their quoted #includes.  (Until late in round 4 the digest covered every file under csrc/, so adding an unrelated
# kernel file -- the coefficient network -- invalidated a traffic record of an unchanged forward.)
DIGEST_UNITS = ("apply_fwd_seg.hip", "apply_fwd_rows.hip", "generic_kernels.hip", "capi.hip")


def source_digest():
    """sha256 over the sources the benchmarked forward is built from: DIGEST_UNITS in hdrnet_amd/csrc, the closure of
    their quoted #includes (csrc/*.h, include/*.h) and hdrnet_amd/build.py (the compiler flags).  Stamps a PMC
    record with the code it was measured on (the GPU box has no .git)."""
    import hashlib
    import re
    csrc = os.path.join(ROOT, "hdrnet_amd", "csrc")
    seen, todo = {}, [os.path.join(csrc, u) for u in DIGEST_UNITS]
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen or not os.path.exists(f):
            continue
        with open(f, "rb") as fh:
            seen[f] = fh.read()
        for inc in re.findall(rb'^\s*#\s*include\s+"([^"]+)"', seen[f], flags=re.M):
            todo.append(os.path.join(os.path.dirname(f), inc.decode()))
    flags = os.path.join(ROOT, "hdrnet_amd", "build.py")
    if os.path.exists(flags):
        with open(flags, "rb") as fh:
            seen[flags] = fh.read()
    h = hashlib.sha256()
    for f in sorted(seen, key=lambda x: os.path.relpath(x, ROOT)):
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(seen[f])
    return h.hexdigest()[:16]


def measured_traffic(workload, kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*/traffic.json:
    FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md section HBM, calibrated on this access
    pattern in profiles/r02/README.md) + WRITE_SIZE, KiB -> B).  bench.py cannot run the profiler
    on itself, so this is the per-launch figure of the same command under `rocprofv3 --pmc`
    (tools/collect_profiles.sh) -- accepted ONLY if the record's source digest equals the digest
    of the kernel sources being benched; otherwise (None, reason)."""
    import glob
    digest = source_digest()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json")), reverse=True):
        try:
            for rec in json.load(open(f)):
                if rec.get("workload") == workload and rec.get("kernel") == kernel:
                    if rec.get("source_digest") == digest:
                        return dict(rec, source=os.path.relpath(f, ROOT)), None
                    stale = os.path.relpath(f, ROOT)
        except (OSError, ValueError):
            pass
    return None, (f"PMC record in {stale} was measured on other kernel sources" if stale
                  else "no PMC record for this workload / kernel")


def make_sets(dev, nsets, B, H, W, GH, GW, GD, seed, smooth_guide=False):
    """grid, guide, input ~ U[0,1) (the reference's own test inputs, hdrnet_ops_test.py:283).
    smooth_guide: a luminance-like low-pass ramp + 2 % noise instead (the
    z-gather locality of a real image; neighbouring pixels then share their LDS reads)."""
    gen = torch.Generator(device=dev).manual_seed(seed)
    sets = []
    for i in range(nsets):
        grid = torch.rand((B, GH, GW, GD, 12), device=dev, generator=gen)
        if smooth_guide:
            yy = torch.linspace(0, 1, H, device=dev)[:, None]
            xx = torch.linspace(0, 1, W, device=dev)[None, :]
            guide = 0.5 + 0.4 * torch.sin(5.0 * xx + 3.0 * yy + i) * torch.cos(2.0 * yy - xx)
            guide = (guide[None] + 0.02 * torch.randn((B, H, W), device=dev, generator=gen)).clamp(0, 1).contiguous()
        else:
            guide = torch.rand((B, H, W), device=dev, generator=gen)
        inp = torch.rand((B, H, W, 3), device=dev, generator=gen)
        out = torch.empty((B, H, W, 3), device=dev)
        sets.append((grid, guide, inp, out))
    return sets


_BOUND = {}


def run_steps(lib, sets, dims, stream, n, start=0, band=None):
    """n launches through the C-ABI over the rotating buffer sets.  The argument tuples are bound
    once per buffer ring (no tensor attribute look-ups on the launch path): at 1080p the kernel
    (12 us) is shorter than a naive Python launch loop.  The cache is keyed on the buffers' device
    addresses (and holds the ring), so a recycled `id()` can never alias another ring.
    band = (y0, rows): launch only that band of every frame through the row-split entry point
    (hdrnet_bilateral_slice_apply_rows_f32; `--split rows`)."""
    key = (tuple(t.data_ptr() for st in sets for t in st), dims, stream, band)
    ent = _BOUND.get(key)
    if ent is None:
        import ctypes
        B, H, W, GH, GW, GD = dims
        c_int, c_vp = ctypes.c_int, ctypes.c_void_p
        if band is None:
            fn = lib.hdrnet_bilateral_slice_apply_f32
            fixed = tuple(c_int(v) for v in (B, H, W, GH, GW, GD, 3, 3, 1)) + (c_vp(stream),)
            calls = [tuple(c_vp(t.data_ptr()) for t in st) + fixed for st in sets]
        else:
            assert B == 1, "--split rows: one frame per launch"
            y0, rows = band
            fn = lib.hdrnet_bilateral_slice_apply_rows_f32
            fixed = tuple(c_int(v) for v in (B, H, y0, rows, W, GH, GW, GD, 3, 3, 1)) + (c_vp(stream),)
            calls = [(c_vp(g.data_ptr()), c_vp(gu.data_ptr() + 4 * y0 * W), c_vp(i.data_ptr() + 12 * y0 * W),
                      c_vp(o.data_ptr() + 12 * y0 * W)) + fixed for (g, gu, i, o) in sets]
        _BOUND.clear()
        ent = _BOUND[key] = (fn, calls, sets)  # holds the ring
    fn, calls, _ = ent
    ns = len(calls)
    for k in range(start, start + n):
        rc = fn(*calls[k % ns])
        if rc != 0:
            raise RuntimeError(f"{fn.__name__} rc={rc}: {lib.hdrnet_last_error().decode()}")


_JSON_FD = None


def protect_stdout():
    """The driver reads ONE JSON line from stdout.  Libraries write there too -- RCCL prints a version banner from C
    stdio when its first communicator comes up, after Python's own buffers have been flushed -- so the process's fd 1
    is pointed at stderr for everything else and the line goes to the original stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(result):
    line = (json.dumps(result) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def timed(step_fn, steps, dist_on, dev):
    """Barrier + sync, K launches (also bracketed by HIP events on the launch stream), sync + barrier.
    Returns (wall seconds, event seconds)."""
    from hdrnet_amd import dist as hd
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    if dist_on:
        hd.barrier()
    torch.cuda.synchronize(dev)
    trace = os.environ.get("HDRNET_BENCH_TRACE") == "1" and steps <= 64  # diagnostics: per-launch events
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps)] if trace else None
    ev0.record()  # (before the wall clock starts: recording an event is not one of the K steps)
    t0 = time.perf_counter()
    if trace:
        for k in range(steps):
            step_fn(1, k)
            evs[k].record()
    else:
        step_fn(steps, 0)
    th = time.perf_counter()
    ev1.record()
    # The closing synchronize is entered with the work already done: a blocking wait wakes up 30-60 us late,
    # which is 5 % of a 20-launch region (profiles/r03/bench_steps20.txt: 41.5 us wall vs 38.9 us events per
    # launch); polling the closing event first costs one core for the length of the region.  What is left of the closing
    # synchronize once the event has fired is ~14-15 us of the call itself, with or without hipDeviceScheduleSpin
    # (tools/exp/spin_probe.py, round 5); the region's other ~9 us are the first launch reaching an idle device.
    while not ev1.query():
        pass
    tq = time.perf_counter()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    if dist_on:
        hd.barrier()
    if os.environ.get("HDRNET_BENCH_TRACE") == "2":  # diagnostics: where the host's share of a short region goes
        print("host trace: launch loop %.1f us, until the closing event fired %.1f us, closing synchronize %.1f us; "
              "events %.1f us" % ((th - t0) * 1e6, (tq - t0) * 1e6, (t1 - tq) * 1e6, ev0.elapsed_time(ev1) * 1e3),
              file=sys.stderr)
    if trace:
        marks = [ev0] + evs
        print("trace: host loop %.0f us; per-launch us: %s" % (
            (th - t0) * 1e6, " ".join("%.0f" % (marks[k].elapsed_time(marks[k + 1]) * 1e3) for k in range(steps))),
            file=sys.stderr)
    return t1 - t0, ev0.elapsed_time(ev1) * 1e-3


def cpu_baseline(H, W, GH, GW, GD):
    """The reference's CPU op timed on this host's cores on bounded samples (oracle/cpu_bench.py):
    value = one core; legs = nproc whole-image processes, numpy restatement of the JAX twin."""
    from oracle import cpu_bench
    return cpu_bench.run(H, W, GH, GW, GD)


def device_index(local_rank, device_count):
    """One process per GPU: rank -> cuda:<local_rank>; on a box with fewer GPUs than ranks (the 1-GPU development box)
    the ranks wrap around and share devices."""
    return local_rank % max(int(device_count), 1)


def launch_command(gpus, port, argv):
    """The command `python bench.py --gpus N` becomes from a bare shell -- the driver's own launch line:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ..."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: become N ranks under torch.distributed.run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(launch_command(args.gpus, port, sys.argv[1:]), env=env)


class _PowerProbe:
    """Shader clock and socket power of one GPU through librocm_smi64 (one sysfs read per call).  Reported beside
    the sustained block because this kernel runs at the package power cap (profiles/r03/power/):
    the numbers say which regime the box was in.  Every failure (no library, no permission) just drops the block."""

    def __init__(self, index):
        import ctypes

        class Freq(ctypes.Structure):
            _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32),
                        ("current", ctypes.c_uint32), ("frequency", ctypes.c_uint64 * 33)]

        self.ct = ctypes
        self.lib = ctypes.CDLL("librocm_smi64.so" if not os.path.exists("/opt/rocm/lib/librocm_smi64.so")
                               else "/opt/rocm/lib/librocm_smi64.so")
        if self.lib.rsmi_init(ctypes.c_uint64(0)):
            raise RuntimeError("rsmi_init failed")
        self.i = ctypes.c_uint32(index)
        self.f, self.p, self.t = Freq(), ctypes.c_uint64(0), ctypes.c_int(0)
        self.sclk, self.power = [], []

    def cap_w(self):
        cap = self.ct.c_uint64(0)
        return None if self.lib.rsmi_dev_power_cap_get(self.i, 0, self.ct.byref(cap)) else cap.value / 1e6

    def sample(self):
        if not self.lib.rsmi_dev_gpu_clk_freq_get(self.i, 0, self.ct.byref(self.f)) and self.f.current < 33:
            self.sclk.append(self.f.frequency[self.f.current] / 1e6)
        if not self.lib.rsmi_dev_power_get(self.i, self.ct.byref(self.p), self.ct.byref(self.t)):
            self.power.append(self.p.value / 1e6)

    def summary(self):
        if not self.sclk or not self.power:
            return None
        return {"socket_w_mean": round(sum(self.power) / len(self.power), 1), "socket_w_max": round(max(self.power), 1),
                "cap_w": self.cap_w(), "sclk_mhz_mean": round(sum(self.sclk) / len(self.sclk)),
                "sclk_mhz_min": round(min(self.sclk)), "sclk_mhz_max": round(max(self.sclk)),
                "samples": len(self.power), "how": "librocm_smi64, read by the launching thread every 4 windows"}


def sustained(step_fn, dev, est_us, seconds=0.5, window=100, smi_index=None):
    """Mean and spread of the per-launch time over `seconds` of back-to-back launches (event windows)."""
    n_win = max(4, int(seconds / (window * est_us * 1e-6)))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
    probe = None
    if smi_index is not None:
        try:
            probe = _PowerProbe(smi_index)
        except Exception:  # noqa: BLE001
            probe = None
    ev[0].record()
    for w in range(n_win):
        step_fn(window, w * window)
        ev[w + 1].record()
        if probe and w % 4 == 3 and w > n_win // 10:  # the device is busy from the first window on (the host
            try:                                        # runs ahead of it); two sysfs reads every 4 windows
                probe.sample()
            except Exception:  # noqa: BLE001
                probe = None
    torch.cuda.synchronize(dev)
    us = [ev[w].elapsed_time(ev[w + 1]) / window * 1e3 for w in range(n_win)]
    lo = min(us)
    out = {"launches": n_win * window, "us_per_launch_mean": round(sum(us) / len(us), 3),
           "window_us_min": round(lo, 3), "window_us_max": round(max(us), 3),
           "slow_window_fraction": round(sum(1 for x in us if x > 1.08 * lo) / len(us), 3),
           "window_launches": window}
    if probe and probe.summary():
        out["power"] = probe.summary()
    return out


def pipelined(lib, sets, dims, dev, nstreams=2, launches=2000, rounds=3):
    """Independent frames issued round-robin on `nstreams` HIP streams: the next frame's pipeline fill runs under the
    previous frame's drain (one stream serialises consecutive launches).  A serving-pipeline figure
    (hdrnet_amd.runtime.FramePipeline), reported BESIDE the headline, never as `value`; wall clock, best of `rounds`."""
    import ctypes
    B, H, W, GH, GW, GD = dims
    if len(sets) < nstreams + 1:
        return None
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    fixed = tuple(ctypes.c_int(v) for v in (B, H, W, GH, GW, GD, 3, 3, 1))
    calls = [tuple(ctypes.c_void_p(t.data_ptr()) for t in sets[k % len(sets)]) + fixed +
             (ctypes.c_void_p(streams[k % nstreams].cuda_stream),) for k in range(len(sets) * nstreams)]
    fn = lib.hdrnet_bilateral_slice_apply_f32

    def run(n):
        for k in range(n):
            if fn(*calls[k % len(calls)]):
                raise RuntimeError(lib.hdrnet_last_error().decode())

    run(launches // 2)
    torch.cuda.synchronize(dev)
    best = None
    for _ in range(rounds):
        t0 = time.perf_counter()
        run(launches)
        torch.cuda.synchronize(dev)
        us = (time.perf_counter() - t0) / launches * 1e6
        best = us if best is None else min(best, us)
    return best


def extra_ops(lib, dev, budget_s=0.3):
    """Short measurements of the OTHER rows of the scope table for the driver's line (`extra`), taken AFTER the headline's
    timed region and its sustained / pipelined blocks, on the same stream: config #2 (one 1080p frame), config #5 as
    stated (uint16 / 32767, 4000x3000, grid 32x32x8x12), the guide network fused into the 4K forward (config #3's hot
    kernel), all three gradients of a 4K frame (config #4's kernels at 4K), and the same backward on a luma_bins = 16
    grid.  Each: rotating buffer sets larger than the Infinity Cache, 10 warm-up launches, then launches for about
    `budget_s` / 2 seconds bracketed by HIP events on the launch stream; `frac` = SURVEY.md section 8d's algorithmic
    bytes over that time over 8 TB/s.  tools/op_bench.py times the same entry points with 5 x 50 launches
    (profiles/r06/ops_*.txt)."""
    import ctypes
    c_int, c_vp, c_f, c_sz, c_u = ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_size_t, ctypes.c_uint
    stream = torch.cuda.current_stream(dev).cuda_stream
    gen = torch.Generator(device=dev).manual_seed(4321)
    out = {}

    def ints(*v):
        return tuple(c_int(x) for x in v)

    def measure(name, fn, calls, abytes, est_us):
        n = len(calls)
        lib.hdrnet_enable_kernel_names(1)
        if fn(*calls[0]):
            raise RuntimeError(lib.hdrnet_last_error().decode())
        kernel = lib.hdrnet_last_kernel().decode()
        lib.hdrnet_enable_kernel_names(0)
        for k in range(10):
            fn(*calls[k % n])
        steps = int(min(4000, max(30, 0.5 * budget_s / (est_us * 1e-6))))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for k in range(steps):
            if fn(*calls[k % n]):
                raise RuntimeError(lib.hdrnet_last_error().decode())
        e1.record()
        torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) * 1e3 / steps
        out[name] = {"kernel": kernel, "us": round(us, 2), "frac": round(abytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                     "algorithmic_bytes": abytes, "launches": steps}

    def nsets_for(abytes):
        return max(3, -(-int(CACHE_BYTES * 1.5) // abytes))

    # config #2: one 1080p frame
    B, H, W, GH, GW, GD = WORKLOADS["1080p"][:6]
    ab = algorithmic_bytes(B, H, W, GH, GW, GD)
    sets = make_sets(dev, nsets_for(ab), B, H, W, GH, GW, GD, seed=2)
    calls = [(c_vp(g.data_ptr()), c_vp(gu.data_ptr()), c_vp(i.data_ptr()), c_vp(o.data_ptr())) +
             ints(B, H, W, GH, GW, GD, 3, 3, 1) + (c_vp(stream),) for (g, gu, i, o) in sets]
    measure("fwd_1080p (config #2)", lib.hdrnet_bilateral_slice_apply_f32, calls, ab, 11.0)
    del sets, calls

    # config #5 as stated: uint16 / 32767 -> f32, 4000x3000, grid 32x32x8x12
    B, H, W, GH, GW, GD = 1, 3000, 4000, 32, 32, 8
    ab = B * (H * W * 22 + 4 * GH * GW * GD * 12)
    sets = [(torch.rand((B, GH, GW, GD, 12), device=dev, generator=gen), torch.rand((B, H, W), device=dev, generator=gen),
             torch.randint(0, 32768, (B, H, W, 3), device=dev, generator=gen, dtype=torch.int32).to(torch.uint16),
             torch.empty((B, H, W, 3), device=dev)) for _ in range(nsets_for(ab))]
    calls = [(c_vp(g.data_ptr()), c_vp(gu.data_ptr()), c_vp(i.data_ptr()), c_vp(o.data_ptr())) +
             ints(B, H, W, GH, GW, GD, 3, 3, 1, 2) + (c_f(32767.0), c_int(0), None, None, c_int(0), None, c_vp(stream))
             for (g, gu, i, o) in sets]
    measure("fwd_hdrp_u16 (config #5)", lib.hdrnet_bilateral_slice_apply_io, calls, ab, 51.0)
    del sets, calls

    # 4K: the guide network fused into the forward, all three gradients (luma_bins = 8 and 16)
    B, H, W, GH, GW = WORKLOADS["4k"][:5]
    npx = B * H * W
    conv1 = (torch.randn((16, 4), device=dev, generator=gen) * 0.8).contiguous()
    conv2 = (torch.randn((17,), device=dev, generator=gen) * 0.5).contiguous()
    pconv1, pconv2 = torch.empty_like(conv1), torch.empty_like(conv2)
    from hdrnet_amd import _lib
    if lib.hdrnet_guide_nn_prescale_f32(c_vp(conv1.data_ptr()), c_vp(conv2.data_ptr()), 16, 3, c_f(65536.0),
                                        c_vp(pconv1.data_ptr()), c_vp(pconv2.data_ptr()), c_vp(stream)):
        raise RuntimeError(lib.hdrnet_last_error().decode())
    for GD in (8, 16):
        gridb = 4 * B * GH * GW * GD * 12
        ab_bwd = 4 * npx * 7 + 4 * npx * 4 + 2 * gridb  # section 8d: 44 B/px + grid in, dgrid out
        sets = make_sets(dev, nsets_for(4 * npx * 11), B, H, W, GH, GW, GD, seed=3 + GD)
        dout = [torch.randn((B, H, W, 3), device=dev, generator=gen) for _ in sets]
        if GD == 8:
            calls = [(c_vp(g.data_ptr()), c_vp(i.data_ptr()), c_vp(pconv1.data_ptr()), c_vp(pconv2.data_ptr()),
                      c_vp(o.data_ptr()), None) + ints(B, H, W, GH, GW, GD, 3, 3, 1, 16) +
                     (c_u(_lib.GUIDE_SIGMOID_FAST | _lib.GUIDE_RELU_PRESCALED), c_vp(stream)) for (g, gu, i, o) in sets]
            measure("fwd_4k_nnguide (config #3's hot kernel)", lib.hdrnet_bilateral_slice_apply_nnguide_f32_ex, calls,
                    4 * npx * 6 + gridb, 41.0)
        wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(B, H, W, GH, GW, GD, 3, 3, 1)
        ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
        dg = torch.empty((B, GH, GW, GD, 12), device=dev)
        dgu = torch.empty((B, H, W), device=dev)
        calls = [(c_vp(g.data_ptr()), c_vp(gu.data_ptr()), c_vp(i.data_ptr()), c_vp(d.data_ptr()), c_vp(dg.data_ptr()),
                  c_vp(dgu.data_ptr()), c_vp(o.data_ptr())) + ints(B, H, W, GH, GW, GD, 3, 3, 1) +
                 (c_vp(ws.data_ptr()), c_sz(wsb), c_vp(stream)) for (g, gu, i, o), d in zip(sets, dout)]
        measure("bwd_4k_all_three" + ("" if GD == 8 else "_luma_bins_16"), lib.hdrnet_bilateral_slice_apply_grad_f32, calls,
                ab_bwd, 108.0 if GD == 8 else 150.0)
        del sets, dout, calls, ws, dg, dgu
    return out


def preroll(step_fn, sync_fn, min_seconds=0.25, chunk=64, max_launches=100000):
    """Untimed launches until >= min_seconds have passed (device out of the idle power state)."""
    n = 0
    t0 = time.perf_counter()
    while n < max_launches:
        step_fn(chunk, n)
        sync_fn()
        n += chunk
        if time.perf_counter() - t0 >= min_seconds:
            break
    return n, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--workload", default="4k", choices=sorted(WORKLOADS) + sorted(EXTRA_WORKLOADS))
    ap.add_argument("--split", default="images", choices=["images", "rows"],
                    help="images (default): every rank slices its OWN frames, weak scaling.  rows: every frame "
                         "is split into N row bands, one per rank (hdrnet_bilateral_slice_apply_rows_f32), strong "
                         "scaling -- the total work is fixed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the two-stream `pipelined` block: its overlapping launches have longer per-kernel durations "
                         "and would be averaged into a rocprofv3 --kernel-trace --stats summary of this command")
    ap.add_argument("--extra", action="store_true",
                    help="also time the cache-resident rate and 1080p (same kernel name at other "
                         "sizes: keep off when collecting rocprofv3 --stats for the roofline line)")
    ap.add_argument("--no-extra", action="store_true", help=argparse.SUPPRESS)  # old spelling, no-op
    ap.add_argument("--no-extra-ops", action="store_true",
                    help="skip `extra.ops`: the short timings of the other configs / the backward after the headline "
                         "(other kernels: keep them out of a rocprofv3 --stats summary of this command)")
    ap.add_argument("--batch-norm", action="store_true",
                    help="train_1080p_b4: the model WITH batch norm in training mode (rounds 1-3 benched this graph).  "
                         "The default is without, as every training script of the reference runs it (scripts/*/*.sh: "
                         "--nobatch_norm; hdrnet/bin/train.py:244 batch_norm=False)")
    ap.add_argument("--no-batch-norm", action="store_true", help=argparse.SUPPRESS)  # the default since round 4; accepted
    ap.add_argument("--force-collective", action="store_true",
                    help="train_1080p_b4 at N = 1: create a one-rank RCCL communicator and issue the flat-bucket all-reduce "
                         "in every step all the same (the multi-rank structure with the real collective kernel on one GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: create the (one-rank) process group all the same, so that the barrier and the "
                         "max-over-ranks of the timed region run through RCCL on the device exactly as at N = 8 "
                         "(the dry run of what the driver launches; tests/test_gpu_bench_dryrun.py)")
    ap.add_argument("--stub-cpu", action="store_true", help=argparse.SUPPRESS)  # tests: gloo + a stub timed body
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    protect_stdout()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world  # n_gpus is what the process group says
    if args.stub_cpu:
        if args.workload == "train_1080p_b4":
            return main_train(args, rank, world, local_rank, stub=True)
        return stub_main(args, rank, world)
    if args.workload == "train_1080p_b4":
        return main_train(args, rank, world, local_rank)
    if args.workload == "hdrp_u16":
        return main_hdrp_u16(args, rank, world, local_rank)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    # One process per GPU.  (On a box with fewer GPUs than ranks -- the 1-GPU development box -- the
    # ranks share devices; RCCL refuses two ranks on one device, so that smoke test of the N > 1 path
    # sets HDRNET_BENCH_BACKEND=gloo.  The driver's 8-GPU runs use the default: nccl = RCCL.)
    dev = torch.device("cuda", device_index(local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    single = bool(args.force_dist) and world == 1
    dist_on = world > 1 or single
    from hdrnet_amd import dist as hd
    backend = os.environ.get("HDRNET_BENCH_BACKEND", "nccl")
    if dist_on:
        hd.init(backend=backend, device=dev, single=single)  # RCCL over xGMI; control plane only (barrier, max-time)

    from hdrnet_amd import _lib
    lib = _lib.load()  # raises loudly if the HIP library is missing

    B, H, W, GH, GW, GD, desc = WORKLOADS[args.workload]
    dims = (B, H, W, GH, GW, GD)
    frame_bytes = algorithmic_bytes(B, H, W, GH, GW, GD)
    band = None
    if args.split == "rows":
        # strong scaling: rank r slices rows [y0, y0 + rows) of EVERY frame; the grid (96 KiB) is replicated
        y0, y1 = hd.row_range(H, rank, world)
        band = (y0, y1 - y0)
    rows_here = band[1] if band else H
    abytes = algorithmic_bytes(B, rows_here, W, GH, GW, GD)  # per launch on this rank
    nsets = max(3, -(-int(CACHE_BYTES * 1.5) // abytes))
    sets = make_sets(dev, nsets, B, H, W, GH, GW, GD, seed=1234 + (0 if band else rank))
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step(n, k0):
        run_steps(lib, sets, dims, stream, n, start=k0, band=band)

    lib.hdrnet_enable_kernel_names(1)
    step(1, 0)
    kernel = lib.hdrnet_last_kernel().decode()
    lib.hdrnet_enable_kernel_names(0)  # no bookkeeping on the launch path from here on
    n_pre, pre_s = preroll(step, lambda: torch.cuda.synchronize(dev))
    step(args.warmup, 0)
    wall, gpu_s = timed(step, args.steps, dist_on, dev)

    wall_max, gpu_max = hd.max_over_ranks([wall, gpu_s], device=dev if backend == "nccl" else torch.device("cpu"))

    # `value`: always the wall clock of the K launches (max over ranks).  images: every rank did K frames
    # of its own; rows: the N ranks together did K frames.
    mp_per_step = B * H * W / 1e6
    value = (1 if band else world) * args.steps * mp_per_step / wall_max
    # ONE clock for the line: `value`, `ms_per_step` and `roofline.achieved` / `.frac` all come from the wall clock of the
    # K timed launches (the events of the same region -- 5-8 % shorter at K = 20, where the host's launch
    # and wake-up latencies are a visible share of 0.8 ms -- used to be `frac` beside a wall-clock `value`).  The event
    # quotient stays in the line under its own name.
    avg_kernel_s = gpu_max / args.steps  # events bracket K back-to-back launches of ONE kernel
    wall_per_launch_s = wall_max / args.steps
    achieved = abytes / wall_per_launch_s / 1e9
    achieved_events = abytes / avg_kernel_s / 1e9
    traffic, traffic_note = measured_traffic(args.workload, kernel) if not band else (None, "row-split launch")

    result = {
        "metric": "megapixels/sec BilateralSliceApply fwd @4K" if args.workload == "4k"
                  else f"megapixels/sec BilateralSliceApply fwd @{args.workload}",
        "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(wall_max / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "strong" if band else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "clock": "wall",
        "preroll_launches": n_pre, "preroll_ms": round(pre_s * 1e3, 1),
        "config": {"workload": desc, "images_per_gpu_per_step": B if not band else round(1.0 / world, 4),
                   "layout": "NHWC fp32",
                   "has_offset": True, "rotating_buffer_sets": nsets,
                   "working_set_MB": round(nsets * frame_bytes / 1e6, 1),
                   "parallelism": (f"row-split x{world} (one frame over all ranks)" if band
                                   else f"image-shard x{world}"),
                   "kernel": kernel,
                   # the communicator behind the barrier / max-over-ranks of the timed region (None: N = 1, no group)
                   "process_group": ({"backend": backend, "world": world, "device": str(dev)} if dist_on else None)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4),
                     "traffic": (traffic or {}).get("bytes_per_launch"),
                     "traffic_source": (traffic or {}).get("source") or traffic_note,
                     "source_digest": source_digest(),
                     "algorithmic_bytes_per_launch": abytes,
                     "wall_us_per_launch": round(wall_per_launch_s * 1e6, 3),
                     "timing": "achieved = algorithmic bytes per launch / (wall clock of the K timed launches / K): the "
                               "clock of `value`; frac_events = the same bytes over HIP events on the launch stream "
                               "around the same K launches / K; frac_sustained = over the mean of 100-launch event "
                               "windows during a further 0.5 s (N = 1)",
                     "frac_events": round(achieved_events / HBM_PEAK_GBPS, 4),
                     "event_us_per_launch": round(avg_kernel_s * 1e6, 3)},
    }

    if world == 1:
        # ~0.5 s more of back-to-back launches in 100-launch HIP-event windows: what a long-running caller
        # gets on this box, and how much it wanders (profiles/r02/exp35).
        sus = sustained(step, dev, est_us=avg_kernel_s * 1e6, smi_index=dev.index or 0)
        result["sustained"] = sus
        result["roofline"]["frac_sustained"] = round(
            abytes / (sus["us_per_launch_mean"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
        if band is None and not args.no_pipelined:
            # the same frames round-robin on two streams (fill of the next under the drain of the previous): what
            # a pipeline of independent frames gets (runtime.FramePipeline) -- beside the headline, not in it
            us2 = pipelined(lib, sets, dims, dev, nstreams=2)
            if us2:
                result["pipelined"] = {"streams": 2, "us_per_frame": round(us2, 3),
                                       "frac": round(abytes / (us2 * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                                       "note": "independent frames round-robin on 2 HIP streams, wall clock, best of 3 x "
                                               "2000 launches; NOT `value` (one stream, the op's plain semantics)"}
    if rank == 0 and world == 1:
        extra = {}
        if args.extra:
            # cache-resident rate (one buffer set, stays in the 256 MiB Infinity Cache): labelled, not `value`
            one = sets[:1]
            run_steps(lib, one, dims, stream, 10)
            _, g1 = timed(lambda n, k0: run_steps(lib, one, dims, stream, n, start=k0), args.steps, False, dev)
            extra["cache_resident_MPps"] = round(args.steps * mp_per_step / g1, 1)
            extra["cache_resident_GBps"] = round(abytes * args.steps / g1 / 1e9, 1)
            # same workload with a smooth (image-like) guide instead of U[0,1) noise
            s3 = make_sets(dev, nsets, B, H, W, GH, GW, GD, seed=77, smooth_guide=True)
            run_steps(lib, s3, dims, stream, min(args.warmup, 200))
            _, g3 = timed(lambda n, k0: run_steps(lib, s3, dims, stream, n, start=k0), args.steps, False, dev)
            extra["smooth_guide_avg_kernel_us"] = round(g3 / args.steps * 1e6, 3)
            extra["smooth_guide_hbm_frac"] = round(abytes / (g3 / args.steps) / 1e9 / HBM_PEAK_GBPS, 4)
            del s3
        if not args.no_extra_ops:
            t_x = time.perf_counter()
            extra["ops"] = extra_ops(lib, dev)
            extra["ops_seconds"] = round(time.perf_counter() - t_x, 2)
        result["extra"] = extra
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(H, W, GH, GW, GD)
    if rank == 0:
        emit(result)
    if dist_on:
        import torch.distributed as dist
        hd.barrier()
        dist.destroy_process_group()



def _init_ranks(world, local_rank, cpu=False, single=False):
    """One process per GPU (main()'s rules): returns (device, dist_on, backend).  single: a one-rank group at N = 1."""
    from hdrnet_amd import dist as hd
    if cpu:
        dev, backend = torch.device("cpu"), "gloo"
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
        dev = torch.device("cuda", device_index(local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(dev)
        backend = os.environ.get("HDRNET_BENCH_BACKEND", "nccl")
    if world > 1 or single:
        hd.init(backend=backend, device=dev, single=single)
    return dev, world > 1 or single, backend


def _finish(result, rank, dist_on):
    from hdrnet_amd import dist as hd
    if rank == 0:
        emit(result)
    if dist_on:
        import torch.distributed as dist
        hd.barrier()
        dist.destroy_process_group()


def main_train(args, rank, world, local_rank, stub=False):
    """BASELINE config #4 as stated: `Training step fwd+bwd 1920x1080 batch=32, grad all-reduce over 8 x MI355X`
    = 4 images per GPU.  Every rank runs runtime.GraphedTrainStep on ITS 4 images -- forward, loss and backward
    through the HIP kernels as one hipGraph whose gradients land in ONE persistent flat fp32 bucket -- then the
    step's only collective, one in-place all-reduce of that bucket (RCCL over xGMI; nothing at N = 1), then the
    (fused, capturable) Adam update.  The reference's step: hdrnet/bin/train.py:113-157 (single device).
    value = whole-job megapixels/s (weak scaling: per-GPU batch fixed).  `allreduce` = the collective alone,
    timed in a loop of its own after the timed steps.
    stub (tests, CPU + gloo): the same harness around runtime.TrainStep on a stand-in torch-only model -- the HIP
    kernels have no CPU path by design."""
    from hdrnet_amd import dist as hd
    from hdrnet_amd import metrics
    force = bool(getattr(args, "force_collective", False)) and world == 1
    dev, dist_on, backend = _init_ranks(world, local_rank, cpu=stub, single=force)
    B, H, W = 4, 1080, 1920
    torch.manual_seed(0)  # identical initial weights on every rank
    if stub:
        from hdrnet_amd.runtime import TrainStep
        model = torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.ReLU(), torch.nn.Linear(32, 3))
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
        torch.manual_seed(1234 + rank)
        inputs, targets = [torch.rand(64, 12)], [torch.rand(64, 3)]
        step = TrainStep(model, lambda out, tgt: metrics.l2_loss(tgt, out), opt)
        run = lambda: step(inputs, targets)  # noqa: E731
        sync = lambda: None  # noqa: E731
        kernel = "stub (torch-only stand-in model on CPU)"
    else:
        from hdrnet_amd import _lib, models
        from hdrnet_amd.runtime import GraphedTrainStep
        _lib.load()  # raises loudly if the HIP library is missing
        model = models.HDRNetPointwiseNNGuide(dict(batch_norm=bool(args.batch_norm))).to(dev).train()
        # Adam over ONE flat parameter buffer laid out like the flat gradient bucket: one 2-us kernel instead of the
        # ~40-us multi-tensor launch of torch.optim.Adam(fused=True) over 35 tensors (hdrnet_amd/optim.py)
        from hdrnet_amd import optim
        opt = optim.FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-4,
                             epsilon_hat=True)  # tf.train.AdamOptimizer's update (hdrnet/bin/train.py:113)
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)  # every rank its own images
        low = torch.rand((B, 256, 256, 3), device=dev, generator=gen)
        full = torch.rand((B, H, W, 3), device=dev, generator=gen)
        target = torch.rand((B, H, W, 3), device=dev, generator=gen)
        step = GraphedTrainStep(model, lambda out, tgt: metrics.l2_loss(tgt, out), opt, [low, full], [target],
                                flat_bucket=True)  # the multi-rank structure at every N, N = 1 included
        step.force_collective = force
        run = lambda: step([low, full], [target])  # noqa: E731
        sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
        kernel = "hipGraph(fwd + loss + bwd) + flat-bucket all-reduce + flat Adam"
    for _ in range(args.warmup):
        run()
    if dist_on:
        hd.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    sync()
    wall = time.perf_counter() - t0
    if dist_on:
        hd.barrier()
    # the collective alone
    n_ar = max(10, min(200, args.steps))
    sync()
    t1 = time.perf_counter()
    for _ in range(n_ar):
        step.bucket.allreduce(force=force)
    sync()
    ar = (time.perf_counter() - t1) / n_ar
    # the same step fed from the graph's own input buffers (a loader that writes the batch where the graph reads it):
    # without the three staging copies of the timed loop above (reported beside `value`, not instead of it)
    static_ms = None
    if not stub:
        feed_in, feed_tgt = step.static_inputs, step.static_targets
        for _ in range(min(args.warmup, 20)):
            step(feed_in, feed_tgt)
        sync()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            step(feed_in, feed_tgt)
        sync()
        static_ms = (time.perf_counter() - t2) / args.steps * 1e3
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    wall_max, ar_max = hd.max_over_ranks([wall, ar], device=red_dev)
    ms = wall_max / args.steps * 1e3
    mp = B * H * W / 1e6
    result = {
        "metric": "megapixels/sec training step fwd+bwd @1080p, 4 images/GPU (BASELINE config #4)",
        "value": round(world * args.steps * mp / wall_max, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "clock": "wall",
        "per_gpu_MPps": round(args.steps * mp / wall_max, 1),
        "ms_per_step_static_feed": None if static_ms is None else round(static_ms, 4),
        "allreduce": {"ms": round(ar_max * 1e3, 4), "share_of_step": round(ar_max * 1e3 / ms, 4),
                      "bucket_elements": int(step.bucket.flat.numel()), "collectives_per_step": 1 if (world > 1 or force) else 0,
                      "forced_at_world_1": force,
                      "backend": backend if dist_on else None},
        "config": {"workload": EXTRA_WORKLOADS["train_1080p_b4"].format(
                       bn="with batch norm" if args.batch_norm else "without batch norm, as the reference's training scripts"),
                   "images_per_gpu_per_step": B,
                   "global_batch": B * world, "parallelism": f"dp{world} (image shards)", "kernel": kernel,
                   "batch_norm": bool(args.batch_norm)},
    }
    _finish(result, rank, dist_on)


def main_hdrp_u16(args, rank, world, local_rank):
    """BASELINE config #5 as stated: `HDR+ 16-bit linear 4000x3000, grid 32x32x8x12, batch=8 sharded 1 image/GPU`:
    every rank slices its own uint16 image -- tf.to_float(im) / 32767 (hdrnet/data_pipeline.py:267-274) fused into the
    kernel (apply_fwd_io.hip), fp32 guide map in, fp32 out -- no data-path collective.  Algorithmic bytes per pixel:
    6 (uint16 RGB) + 4 (guide) + 12 (out) = 22."""
    import ctypes
    from hdrnet_amd import dist as hd
    dev, dist_on, backend = _init_ranks(world, local_rank)
    from hdrnet_amd import _lib
    lib = _lib.load()
    B, H, W, GH, GW, GD = 1, 3000, 4000, 32, 32, 8
    abytes = B * (H * W * 22 + 4 * GH * GW * GD * 12)
    nsets = max(3, -(-int(CACHE_BYTES * 1.5) // abytes))
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    sets = []
    for _ in range(nsets):
        sets.append((torch.rand((B, GH, GW, GD, 12), device=dev, generator=gen),
                     torch.rand((B, H, W), device=dev, generator=gen),
                     torch.randint(0, 32768, (B, H, W, 3), device=dev, generator=gen, dtype=torch.int32).to(torch.uint16),
                     torch.empty((B, H, W, 3), device=dev)))
    stream = torch.cuda.current_stream(dev).cuda_stream
    c_int, c_vp, c_f = ctypes.c_int, ctypes.c_void_p, ctypes.c_float
    fn = lib.hdrnet_bilateral_slice_apply_io
    calls = [(c_vp(g.data_ptr()), c_vp(gu.data_ptr()), c_vp(i.data_ptr()), c_vp(o.data_ptr())) +
             tuple(c_int(v) for v in (B, H, W, GH, GW, GD, 3, 3, 1, 2)) + (c_f(32767.0), c_int(0), None, None, c_int(0), None,
                                                                            c_vp(stream)) for (g, gu, i, o) in sets]

    def step(n, k0):
        for k in range(k0, k0 + n):
            if fn(*calls[k % nsets]):
                raise RuntimeError(lib.hdrnet_last_error().decode())

    lib.hdrnet_enable_kernel_names(1)
    step(1, 0)
    kernel = lib.hdrnet_last_kernel().decode()
    lib.hdrnet_enable_kernel_names(0)
    n_pre, pre_s = preroll(step, lambda: torch.cuda.synchronize(dev))
    step(args.warmup, 0)
    wall, gpu_s = timed(step, args.steps, dist_on, dev)
    wall_max, gpu_max = hd.max_over_ranks([wall, gpu_s], device=dev if backend == "nccl" else torch.device("cpu"))
    mp = B * H * W / 1e6
    avg = gpu_max / args.steps
    wall_l = wall_max / args.steps  # the clock of `value` (main(): one clock for the line)
    result = {
        "metric": "megapixels/sec BilateralSliceApply fwd, HDR+ uint16 wire format @4000x3000 (BASELINE config #5)",
        "value": round(world * args.steps * mp / wall_max, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(wall_max / args.steps * 1e3, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32 (uint16 in)", "data": "synthetic", "clock": "wall",
        "preroll_launches": n_pre, "preroll_ms": round(pre_s * 1e3, 1), "per_gpu_MPps": round(args.steps * mp / wall_max, 1),
        "config": {"workload": EXTRA_WORKLOADS["hdrp_u16"], "images_per_gpu_per_step": B, "global_batch": B * world,
                   "rotating_buffer_sets": nsets, "parallelism": f"image-shard x{world}", "kernel": kernel},
        "roofline": {"bound": "hbm", "achieved": round(abytes / wall_l / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(abytes / wall_l / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                     "algorithmic_bytes_per_launch": abytes, "wall_us_per_launch": round(wall_l * 1e6, 3),
                     "frac_events": round(abytes / avg / 1e9 / HBM_PEAK_GBPS, 4),
                     "event_us_per_launch": round(avg * 1e6, 3),
                     "timing": "achieved / frac: wall clock of the K timed launches / K (the clock of `value`); "
                               "frac_events: HIP events on the launch stream around the same launches"},
    }
    _finish(result, rank, dist_on)


def stub_main(args, rank, world):
    """CPU stand-in for the distributed skeleton of main() (tests/test_bench_launch.py): same
    launch path, barrier, max-over-ranks and JSON line, gloo backend, a sleep as the timed body."""
    from hdrnet_amd import dist as hd
    if world > 1:
        hd.init(backend="gloo", device=torch.device("cpu"))
        hd.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rank))
    wall = time.perf_counter() - t0
    if world > 1:
        hd.barrier()
    (wall_max,) = hd.max_over_ranks([wall], device=torch.device("cpu"))
    # what every rank read from the launcher's environment and the device it would bind (main()'s rule); the device
    # count of the box is faked in tests (HDRNET_BENCH_FAKE_DEVICE_COUNT: this container has no GPU)
    ndev = int(os.environ.get("HDRNET_BENCH_FAKE_DEVICE_COUNT", torch.cuda.device_count()))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    mine = {"rank": rank, "local_rank": local_rank, "world": world, "device_index": device_index(local_rank, ndev),
            "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")),
            "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    ranks = [mine]
    if world > 1:
        import torch.distributed as dist
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    if rank == 0:
        emit({"metric": "stub", "workload": args.workload, "value": round(world * args.steps / wall_max, 1), "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(wall_max / max(args.steps, 1) * 1e3, 5),
                          "higher_is_better": True, "scaling": "weak", "ranks": ranks})
    if world > 1:
        import torch.distributed as dist
        hd.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
