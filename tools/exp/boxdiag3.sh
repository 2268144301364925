#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/boxdiag; mkdir -p $O
T=$(date +%H%M%S)
{
rocminfo 2>/dev/null | grep -i "Uuid" | grep GPU | head -2
python tools/ab_bench.py --variants 39,31,37,29,33,21,19,106 --rounds 3 --steps 1000 2>&1 | grep "^variant" | grep -v "max|" | cut -c1-150
python tools/ab_bench.py --workload 1080p --variants 39,31,37,29,106 --rounds 3 --steps 2000 2>&1 | grep "^variant" | grep -v "max|" | cut -c1-150
} > $O/diag3_$T.txt 2>&1
cat $O/diag3_$T.txt
