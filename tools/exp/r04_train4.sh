#!/bin/bash
R=$(pwd); O=$R/gpurun_out/train; mkdir -p $O
python -m pytest tests/test_models.py tests/test_coeff_net.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do python bench.py --workload train_1080p_b4 --no-batch-norm --steps 100 --warmup 20 2>/dev/null; done > $O/bench_nobn_native.txt
cut -c1-230 $O/bench_nobn_native.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_n -o tr --output-format csv -- python $R/bench.py --workload train_1080p_b4 --no-batch-norm --steps 50 --warmup 10 > /dev/null 2>&1
f=$(find $O/stats_n -name "*kernel_trace.csv" | head -1)
python $R/tools/train_step_profile.py $f --list > $O/step_nobn_native.txt 2>&1
rm -rf $O/stats_n
head -40 $O/step_nobn_native.txt | cut -c1-150
