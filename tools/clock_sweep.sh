#!/bin/bash
# Diagnostic (never part of bench.py): how the 4K forward and its no-compute memory skeleton respond to the
# shader clock.  Runs the default bench + an interleaved product / skeleton A/B under the box's own power
# management, then under `rocm-smi --setperfdeterminism <MHz>` caps, sampling rocm-smi's clocks and power beside
# each run; resets the setting at the end (boxes are per-call and discarded anyway).
#   gpurun --timeout 600 -- 'bash tools/clock_sweep.sh'   ->  gpurun_out/clock/
set -u
R=$(pwd); O=$R/gpurun_out/clock; mkdir -p $O
SMI=/opt/rocm/bin/rocm-smi
$SMI --showclocks --showpower --showperflevel --showmaxpower > $O/idle.txt 2>&1
sample() { while true; do $SMI -c -P --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done; }
run() {
  tag=$1
  sample > $O/smi_$tag.jsonl & SP=$!
  python $R/bench.py --no-cpu-baseline > $O/bench_$tag.json 2>> $O/err.txt
  python $R/tools/ab_bench.py --variants 0,106 --rounds 5 --steps 200 > $O/ab_$tag.txt 2>&1
  kill $SP; wait $SP 2>/dev/null
}
run default
for mhz in ${CLOCKS:-2100 1900 1700 1500 1300}; do
  $SMI --autorespond y --setperfdeterminism $mhz > $O/set_$mhz.txt 2>&1
  run det$mhz
done
$SMI --autorespond y --resetperfdeterminism > $O/reset.txt 2>&1
$SMI --autorespond y -r >> $O/reset.txt 2>&1
run after_reset
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    s = d["sustained"]
    print(f"{sys.argv[1].split('bench_')[-1][:-5]:12s} timed {d['roofline']['avg_kernel_us']:6.2f} us  sustained {s['us_per_launch_mean']:6.2f} "
          f"(windows {s['window_us_min']:.1f}-{s['window_us_max']:.1f}, slow {s['slow_window_fraction']:.2f})")
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
grep -h "variant" $O/ab_*.txt | head -40
