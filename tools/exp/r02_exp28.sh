#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -q -x -m gpu 2>&1 | tail -4
{
python tools/bwd_ab.py --rounds 5 --steps 50 --cases all,gg,g,sl --variants 0,2,3 2>&1 | grep "^case"
python tools/bwd_ab.py --workload 1080p --rounds 5 --steps 100 --cases all,gg,g,sl --variants 0 2>&1 | grep "^case"
python tools/bwd_ab.py --workload hdrp --rounds 4 --steps 50 --cases all,gg,g,sl --variants 0 2>&1 | grep "^case"
} | tee $O/exp28_bwd_fit.txt
