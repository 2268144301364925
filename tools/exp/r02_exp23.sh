#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -q -x -m gpu 2>&1 | tail -2
{
python tools/ab_bench.py --variants 0,28,31,39,106 --rounds 7 --steps 100 2>&1 | grep "^variant" | grep median | cut -c1-190
python tools/ab_bench.py --workload 1080p --variants 0,28,31,39,106 --rounds 7 --steps 400 2>&1 | grep "^variant" | grep median | cut -c1-190
python tools/ab_bench.py --workload hdrp --variants 0,28,31,39 --rounds 5 --steps 100 2>&1 | grep "^variant" | grep median | cut -c1-190
for i in 1 2 3; do python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py', d['value'], d['roofline']['avg_kernel_us'], d['roofline']['frac'])"; done
python bench.py --no-cpu-baseline --workload 1080p | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py 1080p', d['value'], d['roofline']['avg_kernel_us'], d['roofline']['frac'])"
} | tee $O/exp23_store_load_policy.txt
