#!/bin/bash
R=$(pwd); O=$R/gpurun_out/train; mkdir -p $O
for i in 1 2 3; do python bench.py --workload train_1080p_b4 --steps 100 --warmup 20 2>/dev/null; done > $O/bench_bn.txt
for i in 1 2 3; do python bench.py --workload train_1080p_b4 --no-batch-norm --steps 100 --warmup 20 2>/dev/null; done > $O/bench_nobn.txt
cut -c1-230 $O/bench_bn.txt $O/bench_nobn.txt
cd /tmp && export TMPDIR=/tmp
for v in bn nobn; do
  flag=""; [ $v = nobn ] && flag="--no-batch-norm"
  rocprofv3 --kernel-trace --stats -d $O/stats_$v -o tr --output-format csv -- python $R/bench.py --workload train_1080p_b4 $flag --steps 50 --warmup 10 > /dev/null 2>&1
  f=$(find $O/stats_$v -name "*kernel_trace.csv" | head -1)
  python $R/tools/train_step_profile.py $f --list > $O/step_$v.txt 2>&1
  rm -rf $O/stats_$v
  head -30 $O/step_$v.txt
done
