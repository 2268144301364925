#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/boxdiag; mkdir -p $O
T=$(date +%H%M%S)
{
rocminfo 2>/dev/null | grep -i "Uuid" | grep GPU | head -2
python tools/ab_bench.py --variants 39,23,31,21,29,37,20,19,8,106 --rounds 3 --steps 1000 2>&1 | grep "^variant" | grep -v "max|" | cut -c1-150
python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py', d['roofline']['avg_kernel_us'], d['roofline']['frac'])"
} > $O/diag2_$T.txt 2>&1
cat $O/diag2_$T.txt
