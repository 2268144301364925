#!/bin/bash
# after the fused l2 loss, the batched fc backward, 4-wide partial-sum reduction, prefetching first-layer backward-weights, runs of backward-data tiles
R=$PWD; O=$R/gpurun_out/train6; mkdir -p $O
python -m pytest tests/test_coeff_net.py tests/test_models.py -q -m gpu -x 2>&1 | tail -4 > $O/tests.txt
python bench.py --workload train_1080p_b4 --steps 100 --warmup 20 > $O/bench_train_1080p_b4.json 2>/dev/null
for i in 1 2; do python bench.py --workload train_1080p_b4 --steps 100 --warmup 20 2>/dev/null; done > $O/bench_train_repeat.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_n -o tr --output-format csv -- python $R/bench.py --workload train_1080p_b4 --steps 50 --warmup 10 > /dev/null 2>&1
f=$(find $O/stats_n -name '*kernel_trace.csv' | head -1)
python $R/tools/train_step_profile.py $f --list > $O/step.txt 2>&1
rm -rf $O/stats_n
cat $O/tests.txt; cut -c1-300 $O/bench_train_1080p_b4.json; cut -c1-200 $O/bench_train_repeat.txt; head -45 $O/step.txt
