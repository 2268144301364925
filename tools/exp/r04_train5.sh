#!/bin/bash
R=$(pwd); O=$R/gpurun_out/train; mkdir -p $O
python -m pytest tests/test_models.py tests/test_coeff_net.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
python bench.py --workload train_1080p_b4 --steps 100 --warmup 20 > $O/bench_train_1080p_b4.json 2>/dev/null
python bench.py --workload train_1080p_b4 --batch-norm --steps 100 --warmup 20 > $O/bench_train_1080p_b4_batch_norm.json 2>/dev/null
for i in 1 2; do python bench.py --workload train_1080p_b4 --steps 100 --warmup 20 2>/dev/null; done > $O/bench_train_repeat.txt
cut -c1-420 $O/bench_train_1080p_b4.json $O/bench_train_1080p_b4_batch_norm.json
python tools/e2e_bench.py > $O/e2e.txt 2>&1; tail -4 $O/e2e.txt | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_n -o tr --output-format csv -- python $R/bench.py --workload train_1080p_b4 --steps 50 --warmup 10 > /dev/null 2>&1
f=$(find $O/stats_n -name "*kernel_trace.csv" | head -1)
python $R/tools/train_step_profile.py $f --list > $O/step_final.txt 2>&1
rm -rf $O/stats_n
head -32 $O/step_final.txt | cut -c1-130
