#!/bin/bash
# full GPU suite + smoke + the two training lines + the driver-style headline after the caller-side work (forward digest unchanged)
R=$PWD; O=$R/gpurun_out/final2; mkdir -p $O
python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py --workload train_1080p_b4 --steps 100 --warmup 20 > $O/bench_train_1080p_b4.json 2>/dev/null
python bench.py --workload train_1080p_b4 --batch-norm --steps 100 --warmup 20 > $O/bench_train_1080p_b4_batch_norm.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2>/dev/null
python tools/e2e_bench.py > $O/e2e.txt 2>&1
cat $O/gpu_suite.txt; tail -2 $O/smoke.txt; cut -c1-330 $O/bench_train_1080p_b4.json $O/bench_train_1080p_b4_batch_norm.json; cut -c1-200 $O/bench_steps20.json; tail -12 $O/e2e.txt | cut -c1-250
