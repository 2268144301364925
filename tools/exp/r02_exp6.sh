#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r02_exp6; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "backward or grad or determin or optimi or slice" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log
timeout 600 python tools/op_bench.py --tools --workload 4k > $O/ops_4k.txt 2>&1; grep -i "bwd\|error\|Trace" $O/ops_4k.txt
timeout 600 python tools/op_bench.py --tools --workload 1080p > $O/ops_1080p.txt 2>&1; grep -i "bwd\|error\|Trace" $O/ops_1080p.txt
