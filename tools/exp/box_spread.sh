mkdir -p gpurun_out/box
T=$(date +%s)
python bench.py --no-cpu-baseline > gpurun_out/box/bench_$T.json 2>/dev/null
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null; done > gpurun_out/box/steps20_$T.txt
python tools/guide_prepared_ab.py --workload 4k --rounds 5 2>&1 | grep -v amdgpu.ids > gpurun_out/box/guides_$T.txt
python - <<PY
import json,glob
d=json.load(open("gpurun_out/box/bench_$T.json")); r=d["roofline"]
print("box $T default wall us", r["wall_us_per_launch"], "frac", r["frac"], "events", r["frac_events"], "sustained", r["frac_sustained"], "two streams", d["pipelined"]["us_per_frame"])
for l in open("gpurun_out/box/steps20_$T.txt"):
    if l.startswith("{"):
        r=json.loads(l)["roofline"]; print("  steps20 wall", r["wall_us_per_launch"], r["frac"], "events", r["frac_events"])
PY
cat gpurun_out/box/guides_$T.txt
