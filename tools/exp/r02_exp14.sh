#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
{
for wl in 1080p hdrp; do
for rg in 8 16; do
echo "== $wl rg $rg"; HDRNET_GG_RG=$rg python tools/bwd_ab.py --workload $wl --rounds 4 --steps 100 --cases all,gg,g,sl --variants 0 2>&1 | grep "^case"
done; done
} | tee $O/exp14_bwd_rg_sizes.txt
