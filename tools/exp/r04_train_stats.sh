#!/bin/bash
# rocprofv3 --kernel-trace --stats summary of the training-step bench at the round's final state
R=$PWD; O=$R/gpurun_out/train_stats; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o tr --output-format csv -- python $R/bench.py --workload train_1080p_b4 --steps 200 --warmup 20 > $O/bench_under_rocprof.json 2>/dev/null
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp $f $O/train_kernel_stats.csv
t=$(find $O/stats -name '*kernel_trace.csv' | head -1)
python $R/tools/train_step_profile.py $t --list > $O/step.txt 2>&1
rm -rf $O/stats
head -12 $O/train_kernel_stats.csv | cut -c1-150; head -3 $O/step.txt
