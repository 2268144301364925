#!/usr/bin/env python3
"""In-process interleaved A/B timing of BilateralSliceApply forward kernel variants.

    python tools/ab_bench.py [--workload 4k] [--variants 0,19,20] [--rounds 5] [--steps 100]
                             [--trace 36,39] [--settle 80] [--out gpurun_out/ab.json]

A variant may carry experiment knobs (include/hdrnet_amd_tools.h: hdrnet_tools_set_knob), written
`variant@knob=value@knob=value`; the knobs are set before every launch batch of that entry and cleared after it.
(The product kernel's own flavours -- variants 20-72: loads / stores / pixel phase / ticketed tail / timeline trace,
knobs 0-2 and 7 -- were removed in round 5; their records are profiles/r02 .. r04.)

Uses the TOOLS build of the library (libhdrnet_amd_tools.so, include/hdrnet_amd_tools.h): variant 0
is the product kernel, the others are documented in that header.  Every round times each variant
once (HIP events around `steps` back-to-back launches over rotating buffer sets larger than the
Infinity Cache), variants interleaved, and the table reports median / min of the per-launch time.
Each variant is first checked against the generic (bit-exact-to-reference) kernel; a variant that
fails the check or the launch is reported and dropped, the others still run.

--trace: for the listed apply_fwd_seg variants, one launch of the variant's traced twin writes every
workgroup's {start, end} wall-clock ticks (100 MHz) and XCC id; the summary printed is the
launch's timeline: dispatch ramp, workgroup lifetimes, resident-workgroup and retirement profiles
in 1-us buckets, tail.
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import CACHE_BYTES, WORKLOADS, algorithmic_bytes, make_sets  # noqa: E402
from hdrnet_amd import _lib  # noqa: E402


def time_launches(fn, steps):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for k in range(steps):
        fn(k)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps  # us per launch


def trace_summary(tr, name):
    """tr: [nwg, 3] int64: start, end (ticks of 10 ns), XCC id.  The 100 MHz counters of the XCDs
    are not guaranteed to be aligned; the per-XCC lines show each XCD's own first start / last
    end relative to the global first start, and the summary below them is computed AFTER shifting
    every XCD's timeline to a common first start (assumes the XCDs begin within ~0.3 us of each
    other, which is what the dispatcher does)."""
    import numpy as np
    xcc = tr[:, 2]
    t0 = tr[:, 0].min()
    print(f"  trace [{name}]: {len(tr)} workgroups")
    st = np.zeros(len(tr))
    en = np.zeros(len(tr))
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        s0 = tr[m, 0].min()
        print(f"    xcc {x}: {int(m.sum())} wgs, first start {(s0 - t0) * 0.01:+.2f} us, span "
              f"{(tr[m, 1].max() - s0) * 0.01:.2f} us")
        st[m] = (tr[m, 0] - s0) * 0.01
        en[m] = (tr[m, 1] - s0) * 0.01
    life = en - st
    total = en.max()
    nb = int(total) + 1
    started = np.bincount(st.astype(int), minlength=nb)
    retired = np.bincount(en.astype(int), minlength=nb)
    resident = np.cumsum(started) - np.cumsum(retired) + retired  # resident at some point in the bucket
    q = lambda a, p: float(np.percentile(a, p))
    print(f"    span {total:.2f} us (first start -> last end, per-XCD clocks aligned)")
    print(f"    starts: 50% by {q(st, 50):.2f} us, 90% by {q(st, 90):.2f}, last {st.max():.2f}")
    print(f"    lifetime us: p10 {q(life, 10):.2f}  p50 {q(life, 50):.2f}  p90 {q(life, 90):.2f}  max {life.max():.2f}")
    print(f"    ends: first {en.min():.2f} us, 10% by {q(en, 10):.2f}, 90% by {q(en, 90):.2f}, 99% by {q(en, 99):.2f}, last {total:.2f}")
    print("    per-us buckets  resident: " + " ".join(f"{int(v)}" for v in resident))
    print("    per-us buckets  retired : " + " ".join(f"{int(v)}" for v in retired))
    return {"name": name, "nwg": int(len(tr)), "span_us": float(total), "life_p50": q(life, 50),
            "life_p90": q(life, 90), "end_p90": q(en, 90), "end_p99": q(en, 99),
            "first_end": float(en.min()), "resident": [int(v) for v in resident],
            "retired": [int(v) for v in retired]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="4k")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--yardstick", action="store_true")
    ap.add_argument("--trace", default="")
    ap.add_argument("--settle", type=int, default=80, help="untimed launches of a variant before its timed window")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load_tools()
    lib.hdrnet_enable_kernel_names(1)
    B, H, W, GH, GW, GD, desc = WORKLOADS[args.workload]
    abytes = algorithmic_bytes(B, H, W, GH, GW, GD)
    nsets = max(3, -(-int(CACHE_BYTES * 1.5) // abytes))
    sets = make_sets(dev, nsets, B, H, W, GH, GW, GD, 1234)
    stream = torch.cuda.current_stream(dev).cuda_stream
    class Spec:
        """One table entry: a variant number and the knobs to set while it runs."""

        def __init__(self, text):
            parts = text.split("@")
            self.v = int(parts[0])
            self.knobs = [tuple(int(x) for x in kv.split("=")) for kv in parts[1:]]
            self.key = text

        def __enter__(self):
            for k, val in self.knobs:
                lib.hdrnet_tools_set_knob(k, val)

        def __exit__(self, *exc):
            for k, _ in self.knobs:
                lib.hdrnet_tools_set_knob(k, 0)

    variants = [Spec(v) for v in args.variants.split(",") if v != ""]

    def launcher(flags):
        def fn(k):
            grid, guide, inp, out = sets[k % nsets]
            rc = lib.hdrnet_bilateral_slice_apply_f32_ex(
                grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), out.data_ptr(),
                B, H, W, GH, GW, GD, 3, 3, 1, flags, stream)
            if rc:
                raise RuntimeError(lib.hdrnet_last_error().decode())
        return fn

    # correctness of every variant against the generic kernel (same inputs)
    grid, guide, inp, out = sets[0]
    launcher(_lib.KERNEL_GENERIC)(0)
    ref = out.clone()
    names = {}
    errs = {}
    good = []
    for sp in variants:
        v = sp.key
        try:
            out.fill_(float("nan"))
            with sp:
                launcher(_lib.KERNEL_FAST | (sp.v << 8))(0)
                launcher(_lib.KERNEL_FAST | (sp.v << 8))(0)  # twice: a variant with launch-to-launch state must leave it clean
                torch.cuda.synchronize()
            names[v] = lib.hdrnet_last_kernel().decode()
            err = (out - ref).abs().max().item()
            err = float("inf") if err != err else err  # NaN left in the output: pixels nobody wrote
            errs[v] = err
            ok = (err < 1e-5) or names[v].startswith("ABLATION")
            print(f"variant {v} [{names[v]}]: max|fast - generic| = {err:.3e}{'' if ok else '   <-- WRONG, dropped'}",
                  flush=True)
            if ok:
                good.append(sp)
        except Exception as e:  # noqa: BLE001
            print(f"variant {v}: FAILED ({e}), dropped", flush=True)
    variants = good

    results = {sp.key: [] for sp in variants}
    yard = {"copy(out<-in, 2x100MB)": [], "elementwise(out=in*a+b, in 133MB out 100MB)": []}
    for r in range(args.rounds):
        for sp in variants:
            fn = launcher(_lib.KERNEL_FAST | (sp.v << 8))
            # settle: cache-policy stores run ~15 % slower for the first ~50 launches after a kernel that left
            # plain-store dirty lines behind (profiles/r02/exp25); a timed window must not start inside that
            with sp:
                time_launches(fn, args.settle)
                results[sp.key].append(time_launches(fn, args.steps))
        if args.yardstick:
            def cp(k):
                sets[k % nsets][3].copy_(sets[k % nsets][2])
            time_launches(cp, 5)
            yard["copy(out<-in, 2x100MB)"].append(time_launches(cp, args.steps))

            def ew(k):
                g, gu, i, o = sets[k % nsets]
                torch.addcmul(i, i, gu.unsqueeze(-1), out=o)
            time_launches(ew, 5)
            yard["elementwise(out=in*a+b, in 133MB out 100MB)"].append(time_launches(ew, args.steps))

    print(f"\n{desc}; {nsets} rotating sets; algorithmic {abytes / 1e6:.1f} MB/launch")
    table = []
    for sp in variants:
        v = sp.key
        t = results[v]
        med = statistics.median(t)
        print(f"variant {v:>18s} {names[v]:34s} median {med:7.2f} us  min {min(t):7.2f} us  "
              f"-> {abytes / med / 1e3:7.1f} GB/s ({abytes / med / 1e3 / 8000 * 100:4.1f}% of 8 TB/s)  "
              f"{B * H * W / med:9.0f} MP/s   all: {[round(x, 1) for x in t]}")
        table.append({"variant": v, "name": names[v], "median_us": med, "min_us": min(t), "all_us": t,
                      "max_abs_err_vs_generic": errs[v]})
    if args.yardstick:
        vol = {"copy(out<-in, 2x100MB)": 2 * 4 * B * H * W * 3,
               "elementwise(out=in*a+b, in 133MB out 100MB)": 4 * B * H * W * 7}
        for k, t in yard.items():
            med = statistics.median(t)
            print(f"yardstick {k}: median {med:7.2f} us -> {vol[k] / med / 1e3:7.1f} GB/s")

    traces = []
    for sp in [Spec(x) for x in args.trace.split(",") if x != ""]:
        v = sp.v
        # (traced, plain) twins (hdrnet_amd_tools.h): 71 / 70 ticketed tail, 72 / 0 the product
        vt, vp = (v, 70) if v == 71 else (v, 0) if v == 72 else (v + 20, v) if 20 <= v < 40 else (v + 4, v)
        try:
            buf = torch.zeros((1 << 16, 3), dtype=torch.int64, device=dev)
            lib.hdrnet_tools_set_trace(buf.data_ptr())
            fn = launcher(_lib.KERNEL_FAST | (vt << 8))
            plain = launcher(_lib.KERNEL_FAST | (vp << 8))
            with sp:
                time_launches(plain, 50)  # warm clocks, queue is busy right up to the traced launch
                fn(1)
                plain(2)
                torch.cuda.synchronize()
            tr = buf.cpu().numpy()
            tr = tr[tr[:, 0] != 0]
            nm = lib.hdrnet_last_kernel().decode()
            traces.append(dict(trace_summary(tr, f"variant {sp.key} {nm}"), variant=sp.key))
        except Exception as e:  # noqa: BLE001
            print(f"trace variant {v}: FAILED ({e})", flush=True)
        finally:
            lib.hdrnet_tools_set_trace(None)

    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump({"workload": desc, "algorithmic_bytes": abytes, "table": table, "traces": traces}, f)


if __name__ == "__main__":
    main()
