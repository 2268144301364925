#!/bin/bash
# Round-2 experiment 3: the seg kernel as the product forward -- full GPU test suite, smoke, bench.
set -u
R=$(pwd)
O=$R/gpurun_out/r02_exp3
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_short.json 2>> $O/bench.err; cat $O/bench_short.json
timeout 600 python bench.py --workload 1080p --no-cpu-baseline > $O/bench_1080p.json 2>> $O/bench.err; cat $O/bench_1080p.json
timeout 600 python bench.py --workload hdrp --no-cpu-baseline > $O/bench_hdrp.json 2>> $O/bench.err; cat $O/bench_hdrp.json
