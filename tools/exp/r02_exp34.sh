#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -q -x -m gpu 2>&1 | tail -3
{
python tools/bwd_ab.py --rounds 6 --steps 100 --cases v --variants 0,11 2>&1 | grep "^case"
python tools/bwd_ab.py --workload 1080p --rounds 6 --steps 200 --cases v --variants 0,11 2>&1 | grep "^case"
python tools/bwd_ab.py --workload hdrp --rounds 5 --steps 100 --cases v --variants 0,11 2>&1 | grep "^case"
} | tee $O/exp34_vjp_seg.txt
