#!/bin/bash
set -u
R=$(pwd)
for i in 1 2 3; do
  (cd $R/tmp_old && python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OLD 1e9030f', d['roofline']['avg_kernel_us'], d['roofline']['frac'])")
  python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NEW HEAD   ', d['roofline']['avg_kernel_us'], d['roofline']['frac'])"
done
