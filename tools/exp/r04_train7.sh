#!/bin/bash
# gradients written into the bucket; A/B of the step against the two-pass loss as before (same process)
R=$PWD; O=$R/gpurun_out/train7; mkdir -p $O
python -m pytest tests/test_coeff_net.py tests/test_models.py tests/test_bench_launch.py -q -m gpu -x 2>&1 | tail -4 > $O/tests.txt
for i in 1 2 3; do python bench.py --workload train_1080p_b4 --steps 100 --warmup 20 2>/dev/null; done > $O/bench_train_repeat.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_n -o tr --output-format csv -- python $R/bench.py --workload train_1080p_b4 --steps 50 --warmup 10 > /dev/null 2>&1
f=$(find $O/stats_n -name '*kernel_trace.csv' | head -1)
python $R/tools/train_step_profile.py $f --list > $O/step.txt 2>&1
rm -rf $O/stats_n
cat $O/tests.txt; cut -c1-220 $O/bench_train_repeat.txt; head -30 $O/step.txt | cut -c1-110
