#!/bin/bash
# the default bench command (minus the CPU baseline and the two-stream block) under rocprofv3 --kernel-trace --stats
set -u
R=$(pwd); O=$R/gpurun_out/r04stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o fwd --output-format csv -- python $R/bench.py --no-cpu-baseline --no-pipelined > $O/bench_under_rocprof.log 2>&1
grep '^{"metric' $O/bench_under_rocprof.log > $O/bench_under_rocprof.json
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/fwd_kernel_stats.csv \;
rm -rf $O/stats
cat $O/fwd_kernel_stats.csv | head -5 | cut -c1-250; cut -c1-700 $O/bench_under_rocprof.json
