#!/bin/bash
# coefficient network on HIP kernels: parity tests, end-to-end timings, per-kernel durations
R=$(pwd); O=$R/gpurun_out/coeff; mkdir -p $O
python -m pytest tests/test_coeff_net.py -x -q -m gpu -s > $O/tests.txt 2>&1; tail -25 $O/tests.txt
timeout 600 python tools/e2e_bench.py > $O/e2e.txt 2>&1; head -8 $O/e2e.txt
cd /tmp && export TMPDIR=/tmp
cat > /tmp/coef_loop.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from hdrnet_amd import models
m = models.HDRNetPointwiseNNGuide().to("cuda:0").eval()
low = torch.rand(1, 256, 256, 3, device="cuda:0")
with torch.no_grad():
    for _ in range(200):
        m.coefficients(low)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d $O/stats -o coef --output-format csv -- python /tmp/coef_loop.py $R > /dev/null 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/coef_kernel_stats.csv \;
find $O/stats -name "*kernel_trace.csv" -exec cp {} $O/coef_kernel_trace.csv \;
rm -rf $O/stats
head -20 $O/coef_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/coef_kernel_trace.csv")))
rows = [r for r in rows if "coeff_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete pass of 10 launches: start-to-start gaps and durations
last = rows[-10:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:8.2f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:6.2f} us  grid {r.get("Grid_Size_X", "?")}x{r.get("Grid_Size_Y", "?")}  {r["Kernel_Name"][:90]}')
print("pass:", (int(last[-1]["End_Timestamp"]) - t0) / 1e3, "us")
PY
