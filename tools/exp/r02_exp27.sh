#!/bin/bash
# one-round grids: rows per task chosen so that the workgroups just fill the resident slots
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
{
for rg in 16 27 29 32; do echo "== g rg $rg"; HDRNET_GG_RG=$rg python tools/bwd_ab.py --rounds 4 --steps 50 --cases g --variants 0 2>&1 | grep "^case"; done
for rg in 16 34 36 40; do echo "== gg rg $rg"; HDRNET_GG_RG=$rg python tools/bwd_ab.py --rounds 4 --steps 50 --cases gg --variants 0 2>&1 | grep "^case"; done
for rg in 16 24 45 48 54; do echo "== all rg $rg"; HDRNET_GG_RG=$rg python tools/bwd_ab.py --rounds 4 --steps 50 --cases all,sl --variants 0 2>&1 | grep "^case"; done
} | tee $O/exp27_bwd_one_round.txt
