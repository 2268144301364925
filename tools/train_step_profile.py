#!/usr/bin/env python3
"""Anatomy of one graph-captured training step (BASELINE config #4, `bench.py --workload train_1080p_b4`) from a
rocprofv3 kernel trace of that command:

    cd /tmp && rocprofv3 --kernel-trace --stats -d out -o tr --output-format csv -- python bench.py --workload train_1080p_b4 --steps 50 --warmup 10
    python tools/train_step_profile.py out/.../tr_kernel_trace.csv [--list]

One step = the kernels between two consecutive launches of the slice-apply forward (or of --anchor NAME).  Prints the launch count, the span,
the launches grouped by kernel; --list prints them in order.  (Under the profiler a launch costs ~4.5 us even when it
does nothing: the un-profiled step is ~25 % shorter than the span shown.)
"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::|hdrnet_amd::", "", n)
    return n[:96]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    anchor = "apply_fwd_seg"  # a kernel launched once per step; --anchor NAME for graphs where that is another one
    if "--anchor" in sys.argv:
        anchor = sys.argv[sys.argv.index("--anchor") + 1]
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    a, b = idx[-3], idx[-2]
    step = rows[a:b]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3  # noqa: E731
    span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
    print(f"{len(step)} launches per step, span {span:.1f} us under the profiler, sum of kernel durations {sum(map(dur, step)):.1f} us")
    agg = collections.OrderedDict()
    for r in step:
        c = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0])
        c[0] += 1
        c[1] += dur(r)
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{d:8.1f} us {c:3d}x  {k}")
    if "--list" in sys.argv:
        t0 = int(step[0]["Start_Timestamp"])
        for i, r in enumerate(step):
            print(f"{i:3d} {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {dur(r):6.1f}  {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main()
