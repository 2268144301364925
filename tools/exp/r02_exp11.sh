#!/bin/bash
# backward ablations: what bounds the fused dgrid pass?
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python tools/bwd_ab.py --rounds 4 --steps 50 --cases all,gg,g --variants 0,4,5,6,7 2>&1 | grep -v amdgpu.ids | tee $O/exp11_bwd_ablate.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/exp11_stats -o s --output-format csv -- python $R/tools/bwd_ab.py --rounds 2 --steps 20 --cases g --variants 0 > /dev/null 2>&1
find $O/exp11_stats -name "*kernel_stats.csv" -exec cp {} $O/exp11_kernel_stats.csv \;
rm -rf $O/exp11_stats
cut -c1-200 $O/exp11_kernel_stats.csv | head -8
