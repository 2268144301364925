#!/bin/bash
# Which box is this, and how do the forward flavours run on it?
set -u
R=$(pwd); O=$R/gpurun_out/boxdiag; mkdir -p $O
T=$(date +%H%M%S)
{
hostname; uname -r
rocm-smi --showproductname --showclocks --showpower --showtemp --showperflevel --showmemuse --showmemvendor --showcomputepartition --showmemorypartition 2>&1 | grep -v "^=\|^$" | head -60
rocminfo 2>/dev/null | grep -i "Marketing Name\|Compute Unit\|Max Clock\|L2:\|L3:\|Uuid" | head -12
python tools/ab_bench.py --variants 39,21,19,106 --rounds 3 --steps 200 2>&1 | grep "^variant" | grep -v "max|" | cut -c1-130
rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk\|power\|Temp" | head -12
} > $O/diag_$T.txt 2>&1
cat $O/diag_$T.txt
