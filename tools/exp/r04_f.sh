#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04f
mkdir -p $O
export HDRNET_AMD_ALLOW_STALE_LIB=1
for w in 1080p 4k hdrp 1080p_b4; do
timeout 300 python tools/ab_bench.py --workload $w --rounds 9 --variants "0,106,108" > $O/ab_$w.txt 2>&1
done
grep -hE "^variant +[0-9]|^Bilateral" $O/ab_*.txt | cut -c1-200
