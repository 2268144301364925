#!/bin/bash
R=$(pwd); O=$R/gpurun_out/coeff; mkdir -p $O
python -m pytest tests/test_coeff_net.py -x -q -m gpu -s > $O/tests.txt 2>&1; tail -22 $O/tests.txt
python tools/coeff_trace.py 2>&1 | tee $O/trace.txt
python tools/coeff_trace.py --model pyramid 2>&1 | tail -3
timeout 600 python tools/e2e_bench.py > $O/e2e.txt 2>&1; head -8 $O/e2e.txt
