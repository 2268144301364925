#!/bin/bash
# the final build on a fresh box: default bench + the training step
mkdir -p gpurun_out/boxes_final
i=$1
python bench.py --no-cpu-baseline > gpurun_out/boxes_final/bench_box$i.json 2>/dev/null
python bench.py --workload train_1080p_b4 --steps 100 --warmup 20 > gpurun_out/boxes_final/train_box$i.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/boxes_final/bench_box$i.json").read().strip().splitlines()[-1])
t=json.loads(open("gpurun_out/boxes_final/train_box$i.json").read().strip().splitlines()[-1])
r=d["roofline"]; s=d["sustained"]
print($i, d["ms_per_step"], r["frac"], r["frac_sustained"], s["slow_window_fraction"], s["window_us_max"], s["power"]["socket_w_mean"], s["power"]["sclk_mhz_mean"], d["pipelined"]["us_per_frame"], t["ms_per_step"], t["ms_per_step_static_feed"])
PY
