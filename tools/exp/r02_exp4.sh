#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r02_exp4; mkdir -p $O
python tools/exp/hostloop.py 2>&1 | tail -3
timeout 600 python bench.py --workload 1080p --no-cpu-baseline > $O/bench_1080p.json 2>> $O/bench.err; cat $O/bench_1080p.json
timeout 600 python bench.py --workload 4k --no-cpu-baseline > $O/bench_4k.json 2>> $O/bench.err; cat $O/bench_4k.json
timeout 300 python tools/ab_bench.py --workload 1080p --variants 0,39 --rounds 5 --steps 2000 > $O/ab_1080p.txt 2>&1; grep "^variant" $O/ab_1080p.txt | grep -v "max|"
