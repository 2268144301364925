#!/bin/bash
# Round-2 experiment 2: store cache policy (end-of-kernel L2 write-back?), traces with XCC ids.
set -u
R=$(pwd)
O=$R/gpurun_out/r02_exp2
mkdir -p $O
export HDRNET_AMD_KERNEL_NAMES=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "benchmark_variants or no_benchmark" > $O/pytest_variants.log 2>&1; echo "pytest rc=$?" >> $O/pytest_variants.log
tail -4 $O/pytest_variants.log
for wl in 4k 1080p; do
  timeout 300 python tools/ab_bench.py --workload $wl --variants 0,21,25,29,33,37 --rounds 5 --steps 100 --trace 21,33 --out $O/ab_${wl}_a.json > $O/ab_${wl}_a.txt 2>&1
  timeout 300 python tools/ab_bench.py --workload $wl --variants 0,23,27,31,35,39 --rounds 5 --steps 100 --trace 23,35 --out $O/ab_${wl}_b.json > $O/ab_${wl}_b.txt 2>&1
  timeout 300 python tools/ab_bench.py --workload $wl --variants 20,32,22,34,106 --rounds 5 --steps 100 --out $O/ab_${wl}_c.json > $O/ab_${wl}_c.txt 2>&1
done
grep -h "^variant\|trace\|xcc\|span\|starts\|lifetime\|ends:" $O/ab_*.txt | grep -v "max|fast"
