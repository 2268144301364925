#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04occ2; mkdir -p $O
export HDRNET_AMD_ALLOW_STALE_LIB=1
T=${1:-x}
timeout 300 python tools/ab_bench.py --workload 4k --rounds 7 --steps 300 --settle 100 --variants "0,66@0=6000" > $O/ab_4k_$T.txt 2>&1
timeout 300 python tools/ab_bench.py --workload 1080p_b4 --rounds 9 --steps 300 --settle 100 --variants "0,66@0=3000,66@0=7000" > $O/ab_b4_$T.txt 2>&1
timeout 300 python tools/ab_bench.py --workload hdrp --rounds 9 --steps 200 --settle 100 --variants "0,66@0=3000,66@0=7000" > $O/ab_hdrp_$T.txt 2>&1
timeout 300 python tools/ab_bench.py --workload 1080p --rounds 9 --steps 1000 --settle 300 --variants "0,66@0=3000,66@0=6000,66@0=10000" > $O/ab_1080p_$T.txt 2>&1
for w in 4k b4 hdrp 1080p; do echo "-- $w"; grep -hE "^variant +[0-9]" $O/ab_${w}_$T.txt | grep -v "max|" | cut -c1-260; done
