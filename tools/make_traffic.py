#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/rNN/traffic.json, the record
bench.py's `roofline.traffic` reads.

    python tools/make_traffic.py --fetch <dir> --write <dir> --calib-fetch <dir> --calib-write <dir> \
        --workload 4k --out profiles/r02/traffic.json

* `--fetch` / `--write`: counter_collection CSVs of `python bench.py --steps 20 --warmup 5` (the
  product kernel, named in the record as bench.py names it).
* `--calib-*`: the same passes over `tools/ab_bench.py --variants 106` -- the memory skeleton with
  the product kernel's access widths (16 B per lane, lane-contiguous, nontemporal loads) and an
  exactly known byte count (no grid, no staging: reads 4*(1+3) B/px, writes 12 B/px).  The ratio
  known bytes / raw counter is the correction applied to the product kernel's counters
  (MI355X_MICROARCH.md section HBM: gfx950 FETCH_SIZE reports 1/2 of a wide coalesced stream; other
  widths and WRITE_SIZE are to be calibrated on the pattern at hand).
* the record is stamped with bench.source_digest(): bench.py refuses it once the kernel sources change.
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def counter_mean(path, counter, match):
    vals = []
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter and match in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for kernel ~ {match!r} under {path}")
    return sum(vals) / len(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--calib-fetch", required=True)
    ap.add_argument("--calib-write", required=True)
    ap.add_argument("--workload", default="4k")
    ap.add_argument("--kernel-match", default="apply_fwd_seg")
    ap.add_argument("--kernel-name", default="apply_fwd_seg/vec4")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import bench
    B, H, W, GH, GW, GD, _ = bench.WORKLOADS[args.workload]
    npx = H * W
    known_rd, known_wr = 4 * npx * 4, 4 * npx * 3
    cf, ncf = counter_mean(args.calib_fetch, "FETCH_SIZE", "apply_fwd_skeleton")
    cw, ncw = counter_mean(args.calib_write, "WRITE_SIZE", "apply_fwd_skeleton")
    kf = known_rd / (cf * 1024.0)
    kw = known_wr / (cw * 1024.0)
    f, nf = counter_mean(args.fetch, "FETCH_SIZE", args.kernel_match)
    w, nw = counter_mean(args.write, "WRITE_SIZE", args.kernel_match)
    rd = f * 1024.0 * kf
    wr = w * 1024.0 * kw
    rec = {
        "workload": args.workload, "kernel": args.kernel_name,
        "FETCH_SIZE_KiB_per_launch": round(f, 1), "WRITE_SIZE_KiB_per_launch": round(w, 1),
        "calibration": {"kernel": "ABLATION/skeleton nt-ld-contig st-contig (tools variant 106)",
                        "known_read_bytes": known_rd, "known_write_bytes": known_wr,
                        "FETCH_SIZE_KiB": round(cf, 1), "WRITE_SIZE_KiB": round(cw, 1),
                        "fetch_factor": round(kf, 4), "write_factor": round(kw, 4), "launches": [ncf, ncw]},
        "read_bytes": int(round(rd)), "write_bytes": int(round(wr)),
        "bytes_per_launch": int(round(rd + wr)),
        "algorithmic_bytes_per_launch": bench.algorithmic_bytes(B, H, W, GH, GW, GD),
        "launches": [nf, nw], "source_digest": bench.source_digest(),
        "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py --steps 20 "
               "--warmup 5`; per-launch means; counters (KiB) x 1024 x the factor calibrated on the skeleton's "
               "known byte count in the same passes' twin (tools/collect_profiles.sh)",
    }
    recs = []
    if os.path.exists(args.out):
        recs = [r for r in json.load(open(args.out)) if not (r.get("workload") == args.workload and r.get("kernel") == args.kernel_name)]
    recs.append(rec)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(recs, open(args.out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
