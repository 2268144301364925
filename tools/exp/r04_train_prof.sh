#!/bin/bash
R=$(pwd); O=$R/gpurun_out/train; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o tr --output-format csv -- python $R/bench.py --workload train_1080p_b4 --steps 50 --warmup 10 > $O/bench.log 2>&1
tail -2 $O/bench.log | cut -c1-400
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/stats -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/stats
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 50 steps: find the repeating period by the fused Adam kernel
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "multi_tensor_apply" in n or "fused_adam" in n.lower()]
print("kernels total", len(rows), "adam-like launches", len(idx))
if len(idx) > 4:
    a, b = idx[-3], idx[-2]
    step = rows[a + 1:b + 1]
    t0 = int(step[0]["Start_Timestamp"]); t1 = int(step[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    print(f"one step: {len(step)} kernels, span {(t1 - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
    agg = {}
    for r in step:
        n = r["Kernel_Name"][:70]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c = agg.setdefault(n, [0, 0]); c[0] += 1; c[1] += d
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"{d / 1e3:9.1f} us {c:4d}x  {n}")
PY
