#!/usr/bin/env python3
"""Diagnostic: socket power and shader clock beside the per-window launch time of forward-kernel variants.

    python tools/power_probe.py [--variants 0,106] [--seconds 2.0] [--workload 4k] [--out gpurun_out/power.json]

For every variant (tools build; 0 = product, 106 = the no-compute memory skeleton, see include/hdrnet_amd_tools.h)
it launches windows of 100 back-to-back launches for `seconds`, keeps two windows queued, and after each
window's end event reads the current shader clock and socket power through librocm_smi64 (one sysfs read each,
same thread -- no second process, nothing between the launches).  Printed per variant: us / launch (mean, window
min / median / max, fraction of windows > 1.05 x the fastest), shader clock and power (mean, min, max) over the
loaded windows, and their means over the slow and the fast windows separately.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import CACHE_BYTES, WORKLOADS, algorithmic_bytes, make_sets  # noqa: E402
from hdrnet_amd import _lib  # noqa: E402


class _Freq(ctypes.Structure):
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                ("frequency", ctypes.c_uint64 * 33)]


class Smi:
    def __init__(self, index=0):
        self.lib = ctypes.CDLL("/opt/rocm/lib/librocm_smi64.so")
        rc = self.lib.rsmi_init(ctypes.c_uint64(0))
        if rc:
            raise RuntimeError(f"rsmi_init -> {rc}")
        self.i = ctypes.c_uint32(index)
        self.f = _Freq()
        self.p = ctypes.c_uint64(0)
        self.t = ctypes.c_int64(0)
        self.ptype = ctypes.c_int(0)

    def sclk_mhz(self):
        rc = self.lib.rsmi_dev_gpu_clk_freq_get(self.i, 0, ctypes.byref(self.f))  # RSMI_CLK_TYPE_SYS
        if rc or self.f.current >= 33:
            return float("nan")
        return self.f.frequency[self.f.current] / 1e6

    def power_w(self):
        rc = self.lib.rsmi_dev_power_get(self.i, ctypes.byref(self.p), ctypes.byref(self.ptype))
        if rc:
            rc = self.lib.rsmi_dev_current_socket_power_get(self.i, ctypes.byref(self.p))
        return float("nan") if rc else self.p.value / 1e6

    def hotspot_c(self):
        rc = self.lib.rsmi_dev_temp_metric_get(self.i, 1, 0, ctypes.byref(self.t))  # junction, current
        return float("nan") if rc else self.t.value / 1e3


class EffClk:
    """One sleeping wave per window on a second stream, started by the window's opening event: shader-clock ticks
    (clock64) per 100-MHz wall-clock tick = the clock the CU really ran at, whatever the firmware reports."""

    def __init__(self, dev, max_windows=4096, iters=700):
        path = os.path.join(ROOT, "tools", "debug", "ubench", "bin", "libclkprobe.so")
        if not os.path.exists(path):  # built artefacts are not in the history: make it (hipcc is in the image)
            import subprocess
            subprocess.run(["make", "-C", os.path.dirname(os.path.dirname(path)), "bin/libclkprobe.so"], check=True,
                           stdout=subprocess.DEVNULL)
        self.lib = ctypes.CDLL(path)
        self.lib.clk_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self.out = torch.zeros((max_windows, 2), dtype=torch.int64, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.iters = iters
        self.n = 0

    def start(self, opening_event):
        if self.n >= self.out.shape[0]:
            return
        self.stream.wait_event(opening_event)
        rc = self.lib.clk_probe_launch(self.out[self.n].data_ptr(), self.iters, self.stream.cuda_stream)
        if rc:
            raise RuntimeError(f"clk_probe_launch -> {rc}")
        self.n += 1

    def mhz(self):
        torch.cuda.synchronize()
        o = self.out[:self.n].cpu().tolist()
        self.n = 0
        return [100.0 * c / w if w else float("nan") for c, w in o]


class Metrics:
    """gpu_metrics through the amdsmi python package that ships with ROCm (per-XCD clocks, throttler residencies)."""

    def __init__(self):
        sys.path.insert(0, "/opt/rocm/share/amd_smi")
        import amdsmi  # noqa: PLC0415
        self.a = amdsmi
        amdsmi.amdsmi_init()
        self.h = amdsmi.amdsmi_get_processor_handles()[0]

    def read(self):
        return self.a.amdsmi_get_gpu_metrics_info(self.h)


def _num(x):
    return x if isinstance(x, (int, float)) and not isinstance(x, bool) else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="4k")
    ap.add_argument("--variants", default="0,106")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--window", type=int, default=100)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--metrics", action="store_true", help="also read gpu_metrics (amdsmi) once per window")
    ap.add_argument("--dump-keys", action="store_true")
    ap.add_argument("--effclk", action="store_true",
                    help="count the EFFECTIVE shader clock beside every window (tools/debug/ubench/clk_probe.hip on a second stream)")
    ap.add_argument("--grad", action="store_true", help="the gradient entry point instead of the forward variants")
    ap.add_argument("--pattern", default="", help="e.g. 0,0,0,106: one continuous run, windows cycling through these variants")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load_tools()
    lib.hdrnet_enable_kernel_names(1)
    smi = Smi(0)
    met = None
    if args.metrics:
        met = Metrics()
        m0 = met.read()
        if args.dump_keys:
            print("gpu_metrics keys:", ", ".join(f"{k}={m0[k]}" for k in sorted(m0)))
    B, H, W, GH, GW, GD, desc = WORKLOADS[args.workload]
    abytes = algorithmic_bytes(B, H, W, GH, GW, GD)
    nsets = max(3, -(-int(CACHE_BYTES * 1.5) // abytes))
    sets = make_sets(dev, nsets, B, H, W, GH, GW, GD, 1234)
    stream = torch.cuda.current_stream(dev).cuda_stream
    print(f"{desc}; idle: sclk {smi.sclk_mhz():.0f} MHz, power {smi.power_w():.0f} W, hotspot {smi.hotspot_c():.0f} C")

    def launcher(flags):
        def fn(k):
            grid, guide, inp, out = sets[k % nsets]
            rc = lib.hdrnet_bilateral_slice_apply_f32_ex(
                grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), out.data_ptr(),
                B, H, W, GH, GW, GD, 3, 3, 1, flags, stream)
            if rc:
                raise RuntimeError(lib.hdrnet_last_error().decode())
        return fn

    if args.pattern:
        # one continuous run whose windows cycle through the pattern's variants: does a kernel that never triggers
        # the slow state on its own (the skeleton) run slow INSIDE the slow episodes another kernel triggers?
        pat = [int(x) for x in args.pattern.split(",")]
        fns = {v: launcher(_lib.KERNEL_FAST | (v << 8)) for v in set(pat)}
        evs = [torch.cuda.Event(enable_timing=True)]
        evs[0].record()
        rows = []
        k = 0
        done = 0
        t_start = time.perf_counter()
        eff = EffClk(dev) if args.effclk else None
        while time.perf_counter() - t_start < args.seconds:
            fn = fns[pat[(len(evs) - 1) % len(pat)]]
            if eff:
                eff.start(evs[-1])
            for _ in range(args.window):
                fn(k)
                k += 1
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
            if len(evs) - 1 - done > 2:
                evs[done + 1].synchronize()
                rows.append((pat[done % len(pat)], evs[done].elapsed_time(evs[done + 1]) * 1e3 / args.window,
                             smi.sclk_mhz(), smi.power_w()))
                done += 1
        torch.cuda.synchronize()
        if eff:
            em = eff.mhz()
            rows = [r + (em[i] if i < len(em) else float("nan"),) for i, r in enumerate(rows)]
        rows = rows[len(rows) // 10:]
        mean = lambda a: (sum(a) / len(a)) if a else float("nan")
        lead = pat[0]
        fast = min(r[1] for r in rows if r[0] == lead)
        # state of a window = state of the nearest lead-variant window at or before it
        state = []
        cur = False
        for r in rows:
            if r[0] == lead:
                cur = r[1] > 1.05 * fast
            state.append(cur)
        print(f"pattern {pat}: {len(rows)} windows; lead variant {lead} fastest window {fast:.2f} us, slow fraction "
              f"{mean([1.0 if s_ else 0.0 for r, s_ in zip(rows, state) if r[0] == lead]):.2f}")
        for v in sorted(set(pat)):
            for st_name, st_val in (("fast episodes", False), ("slow episodes", True)):
                sel = [r for r, s_ in zip(rows, state) if r[0] == v and s_ == st_val]
                if sel:
                    print(f"  variant {v:3d} in {st_name}: {len(sel):4d} windows  {mean([r[1] for r in sel]):6.2f} us "
                          f"(min {min(r[1] for r in sel):.2f}, max {max(r[1] for r in sel):.2f})   sclk {mean([r[2] for r in sel]):.0f} MHz   "
                          f"power {mean([r[3] for r in sel]):.0f} W"
                          + (f"   EFFECTIVE clock {mean([r[4] for r in sel]):.0f} MHz ({min(r[4] for r in sel):.0f}-{max(r[4] for r in sel):.0f})"
                             if len(sel[0]) > 4 else ""))
        print("  timeline (variant:us): " + " ".join(f"{r[0]}:{r[1]:.1f}" for r in rows[:160]))
        return
    if args.grad:
        # the gradient entry point (product library): all three gradients, dgrid + dguide, dgrid alone
        plib = _lib.load()
        Cin, Cout, C = 3, 3, 12
        gen = torch.Generator(device=dev).manual_seed(1)
        G = [dict(dout=torch.randn((B, H, W, Cout), device=dev, generator=gen),
                  dgrid=torch.empty((B, GH, GW, GD, C), device=dev), dguide=torch.empty((B, H, W), device=dev),
                  dinput=torch.empty((B, H, W, Cin), device=dev)) for _ in range(nsets)]
        n = plib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(B, H, W, GH, GW, GD, Cin, Cout, 1)
        ws = torch.empty((max(n, 16),), dtype=torch.uint8, device=dev)
        for case, (dg, dgu, di) in (("all three", (1, 1, 1)), ("dgrid + dguide", (1, 1, 0)), ("dgrid", (1, 0, 0))):
            def fn(k):
                grid, guide, inp, _ = sets[k % nsets]
                g = G[k % nsets]
                rc = plib.hdrnet_bilateral_slice_apply_grad_f32(
                    grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), g["dout"].data_ptr(),
                    g["dgrid"].data_ptr() if dg else None, g["dguide"].data_ptr() if dgu else None,
                    g["dinput"].data_ptr() if di else None, B, H, W, GH, GW, GD, Cin, Cout, 1,
                    ws.data_ptr(), ws.numel(), stream)
                if rc:
                    raise RuntimeError(plib.hdrnet_last_error().decode())
            evs = [torch.cuda.Event(enable_timing=True)]
            evs[0].record()
            rows, k, done = [], 0, 0
            t_start = time.perf_counter()
            while time.perf_counter() - t_start < args.seconds:
                for _ in range(args.window // 2):
                    fn(k)
                    k += 1
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
                if len(evs) - 1 - done > 2:
                    evs[done + 1].synchronize()
                    rows.append((evs[done].elapsed_time(evs[done + 1]) * 1e3 / (args.window // 2), smi.sclk_mhz(), smi.power_w()))
                    done += 1
            torch.cuda.synchronize()
            rows = rows[len(rows) // 10:]
            mean = lambda a: sum(a) / len(a)
            print(f"apply grad, {case:15s} {mean([r[0] for r in rows]):7.2f} us (windows {min(r[0] for r in rows):.2f}-{max(r[0] for r in rows):.2f})"
                  f"   sclk {mean([r[1] for r in rows]):.0f} MHz ({min(r[1] for r in rows):.0f}-{max(r[1] for r in rows):.0f})"
                  f"   power {mean([r[2] for r in rows]):.0f} W ({min(r[2] for r in rows):.0f}-{max(r[2] for r in rows):.0f})")
            time.sleep(0.3)
        return
    record = {}
    eff = EffClk(dev) if args.effclk else None
    for rep in range(args.repeat):
        for v in [int(x) for x in args.variants.split(",")]:
            fn = launcher(_lib.KERNEL_FAST | (v << 8))
            fn(0)
            torch.cuda.synchronize()
            name = lib.hdrnet_last_kernel().decode()
            evs = [torch.cuda.Event(enable_timing=True)]
            evs[0].record()
            rows = []  # (t_host, us_per_launch, sclk, power)
            mrows = []  # gpu_metrics dict per window
            m_begin = met.read() if met else None
            k = 0
            t_start = time.perf_counter()
            done = 0
            while True:
                if eff:
                    eff.start(evs[-1])
                for _ in range(args.window):
                    fn(k)
                    k += 1
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
                if len(evs) - 1 - done > 2:  # keep two windows queued behind the one being read
                    evs[done + 1].synchronize()
                    us = evs[done].elapsed_time(evs[done + 1]) * 1e3 / args.window
                    rows.append((time.perf_counter() - t_start, us, smi.sclk_mhz(), smi.power_w()))
                    if met:
                        mrows.append(met.read())
                    done += 1
                if time.perf_counter() - t_start > args.seconds:
                    break
            torch.cuda.synchronize()
            temp = smi.hotspot_c()
            if eff:
                em = eff.mhz()[:len(rows)]
                cut = len(rows) // 10
                usw = [r[1] for r in rows[cut:]]
                lim = 1.05 * min(usw)
                ef = [m for m, u in zip(em[cut:], usw) if u <= lim]
                es = [m for m, u in zip(em[cut:], usw) if u > lim]
                mean_ = lambda a: (sum(a) / len(a)) if a else float("nan")
                print(f"            EFFECTIVE clock (counted): fast windows {mean_(ef):.0f} MHz ({min(ef):.0f}-{max(ef):.0f})"
                      + (f", slow windows {mean_(es):.0f} MHz ({min(es):.0f}-{max(es):.0f})" if es else ""))
            if met:
                m_end = met.read()
                cut = len(rows) // 10
                mr = mrows[cut:]
                usw = [r[1] for r in rows[cut:]]
                lim = 1.05 * min(usw)
                skip = ("firmware_timestamp", "system_clock_counter", "energy_accumulator", "pcie_", "xgmi_", "current_dclk",
                        "current_vclk", "jpeg", "vcn", "accumulation_counter", "temperature_vrsoc")
                for key in sorted(m_end):
                    if any(key.startswith(p) or p in key for p in skip):
                        continue
                    b, e = m_begin.get(key), m_end.get(key)
                    if key.startswith("xcp_stats.") and isinstance(e, list) and e and isinstance(e[0], list):
                        d = [y - x for x, y in zip(b[0], e[0]) if _num(x) is not None and _num(y) is not None]
                        if d and any(d):
                            print(f"            {key}[partition 0] per XCD: +{d}")
                        continue
                    if key.endswith("_acc") or "residency" in key or "accumul" in key:
                        if _num(b) is not None and _num(e) is not None and e != b:
                            print(f"            {key}: +{e - b}")
                        continue
                    vals = [m.get(key) for m in mr]
                    if vals and isinstance(vals[0], list):
                        nums = [[x for x in v if _num(x) is not None and x not in (65535, 4294967295)] for v in vals]
                        if not nums or not nums[0]:
                            continue
                        lo = [min(v) for v in nums if v]
                        hi = [max(v) for v in nums if v]
                        fastv = [min(v) for v, u in zip(nums, usw) if v and u <= lim]
                        slowv = [min(v) for v, u in zip(nums, usw) if v and u > lim]
                        mean = lambda a: (sum(a) / len(a)) if a else float("nan")
                        print(f"            {key}: per-sample min {min(lo)}..{max(lo)}, max {min(hi)}..{max(hi)};  mean of the per-sample MIN over"
                              f" fast windows {mean(fastv):.0f}, slow windows {mean(slowv):.0f}")
                    elif vals and _num(vals[0]) is not None:
                        nv = [x for x in vals if _num(x) is not None]
                        if nv and (max(nv) != min(nv) or key in ("throttle_status", "indep_throttle_status")):
                            fastv = [x for x, u in zip(vals, usw) if _num(x) is not None and u <= lim]
                            slowv = [x for x, u in zip(vals, usw) if _num(x) is not None and u > lim]
                            mean = lambda a: (sum(a) / len(a)) if a else float("nan")
                            print(f"            {key}: {min(nv)}..{max(nv)}  (fast windows {mean(fastv):.1f}, slow windows {mean(slowv):.1f})")
            rows = rows[len(rows) // 10:]  # the ramp-up of clock and power is not the load state
            us = [r[1] for r in rows]
            fast = min(us)
            slow_rows = [r for r in rows if r[1] > 1.05 * fast]
            fast_rows = [r for r in rows if r[1] <= 1.05 * fast]
            mean = lambda a: (sum(a) / len(a)) if a else float("nan")
            print(f"variant {v:3d} {name:44s} {mean(us):6.2f} us  windows {fast:.2f} / {statistics.median(us):.2f} / {max(us):.2f}"
                  f"  slow {len(slow_rows) / len(rows):.2f}   sclk {mean([r[2] for r in rows]):.0f} MHz"
                  f" ({min(r[2] for r in rows):.0f}-{max(r[2] for r in rows):.0f})   power {mean([r[3] for r in rows]):.0f} W"
                  f" ({min(r[3] for r in rows):.0f}-{max(r[3] for r in rows):.0f})   hotspot {temp:.0f} C")
            if slow_rows:
                print(f"            fast windows: sclk {mean([r[2] for r in fast_rows]):.0f} MHz, power {mean([r[3] for r in fast_rows]):.0f} W;"
                      f"  slow windows ({mean([r[1] for r in slow_rows]):.2f} us): sclk {mean([r[2] for r in slow_rows]):.0f} MHz,"
                      f" power {mean([r[3] for r in slow_rows]):.0f} W", flush=True)
            record.setdefault(str(v), []).append({"name": name, "rows": rows})
            time.sleep(0.3)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(record, fh)


if __name__ == "__main__":
    main()
