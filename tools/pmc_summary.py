#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel mean of each counter.

    python tools/pmc_summary.py <dir-or-csv> [<dir-or-csv> ...] [--match apply_fwd]

FETCH_SIZE / WRITE_SIZE are reported raw (KiB) and as bytes with the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE under-reports a wide coalesced stream by 2x).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def files(args):
    for a in args:
        if os.path.isdir(a):
            yield from glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
        else:
            yield a


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    acc = defaultdict(lambda: defaultdict(list))
    for f in files(args):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if match and match not in k:
                    continue
                short = k.replace("(anonymous namespace)::", "").replace("hdrnet_amd::", "")
                short = short[:short.rfind("(")] if "(" in short else short
                short = short.replace("void ", "")[-80:]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
                acc[short]["_vgpr"] = [float(row.get("VGPR_Count", 0) or 0)]
                acc[short]["_lds"] = [float(row.get("LDS_Block_Size", 0) or 0)]
    for k, ctrs in acc.items():
        print(k)
        for c, v in sorted(ctrs.items()):
            m = sum(v) / len(v)
            extra = ""
            if c == "FETCH_SIZE":
                extra = f"  -> {m * 1024 / 1e6:.2f} MB raw, x2 gfx950 correction = {2 * m * 1024 / 1e6:.2f} MB"
            if c == "WRITE_SIZE":
                extra = f"  -> {m * 1024 / 1e6:.2f} MB"
            print(f"  {c:28s} n={len(v):4d} mean={m:16.1f}{extra}")


if __name__ == "__main__":
    main()
