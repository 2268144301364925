#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04stagger; mkdir -p $O
export HDRNET_AMD_ALLOW_STALE_LIB=1
T=${1:-x}
timeout 500 python tools/ab_bench.py --workload 4k --rounds 9 --steps 300 --settle 100 --variants "0,0@7=1,0@7=3,0@7=6,66@0=3000,66@0=6000,66@0=9000,65" > $O/ab_$T.txt 2>&1
grep -hE "^variant +[0-9]" $O/ab_$T.txt | cut -c1-260
