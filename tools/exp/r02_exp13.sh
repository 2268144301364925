#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -m gpu -k "grad or bwd or backward or determin" 2>&1 | tail -3
{
echo "== rg 8"; python tools/bwd_ab.py --rounds 4 --steps 50 --cases all,gg,g,sl,v --variants 0 2>&1 | grep "^case"
echo "== rg 16"; HDRNET_GG_RG=16 python tools/bwd_ab.py --rounds 4 --steps 50 --cases all,gg,g,sl --variants 0 2>&1 | grep "^case"
echo "== rg 12"; HDRNET_GG_RG=12 python tools/bwd_ab.py --rounds 4 --steps 50 --cases all,gg,g,sl --variants 0 2>&1 | grep "^case"
} | tee $O/exp13_bwd_rg.txt
