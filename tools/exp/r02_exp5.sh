#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r02_exp5; mkdir -p $O
for st in 100 400 2000; do
timeout 300 python tools/ab_bench.py --workload 1080p --variants 23,31,35,39,106 --rounds 3 --steps $st > $O/ab_1080p_$st.txt 2>&1; echo "== 1080p steps $st"; grep "^variant" $O/ab_1080p_$st.txt | grep -v "max|" | cut -c1-150
done
for st in 100 2000; do
timeout 300 python tools/ab_bench.py --workload 4k --variants 23,31,35,39,106 --rounds 3 --steps $st > $O/ab_4k_$st.txt 2>&1; echo "== 4k steps $st"; grep "^variant" $O/ab_4k_$st.txt | grep -v "max|" | cut -c1-150
done
