#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/exp31_stats -o s --output-format csv -- python $R/tools/bwd_ab.py --rounds 2 --steps 50 --cases g,gg,all --variants 0 > /dev/null 2>&1
find $O/exp31_stats -name "*kernel_stats.csv" -exec cp {} $O/exp31_bwd_kernel_stats.csv \;
rm -rf $O/exp31_stats
cut -d, -f1-4 $O/exp31_bwd_kernel_stats.csv | cut -c1-200 | head -8
