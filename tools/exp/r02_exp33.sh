#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -m gpu -k "grad or bwd or backward or determin" 2>&1 | tail -2
{
python tools/bwd_ab.py --rounds 6 --steps 50 --cases all,gg,g --variants 0,10 2>&1 | grep "^case"
python tools/bwd_ab.py --workload 1080p --rounds 6 --steps 100 --cases all,gg,g --variants 0,10 2>&1 | grep "^case"
python tools/bwd_ab.py --rounds 4 --steps 50 --cases sl --variants 0 2>&1 | grep "^case"
} | tee $O/exp33_bwd_deferred_fold.txt
