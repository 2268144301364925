#!/bin/bash
# fc3 sums the per-tile column sums itself (no slab-sum launch), recompute in 32 slabs + x1 workgroups; e2e lines for the other model classes
R=$PWD; O=$R/gpurun_out/train8; mkdir -p $O
python -m pytest tests/test_coeff_net.py tests/test_models.py -q -m gpu -x 2>&1 | tail -3 > $O/tests.txt
for i in 1 2; do python bench.py --workload train_1080p_b4 --steps 100 --warmup 20 2>/dev/null; done > $O/bench_train_repeat.txt
python tools/e2e_bench.py > $O/e2e.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_n -o tr --output-format csv -- python $R/bench.py --workload train_1080p_b4 --steps 50 --warmup 10 > /dev/null 2>&1
f=$(find $O/stats_n -name '*kernel_trace.csv' | head -1)
python $R/tools/train_step_profile.py $f --list > $O/step.txt 2>&1
rm -rf $O/stats_n
cat $O/tests.txt; cut -c1-220 $O/bench_train_repeat.txt; tail -4 $O/e2e.txt | cut -c1-260; head -30 $O/step.txt | cut -c1-110
