#!/bin/bash
# Round 4, experiment A: launch shape of the forward (tools build).  Occupancy caps (knob 0), ticketed tail
# (variants 70 / 71), timeline traces.  Output: gpurun_out/r04a/
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04a
mkdir -p $O
export HDRNET_AMD_ALLOW_STALE_LIB=1
V4K="0,66@0=3000,66@0=6000,66@0=9000,66@0=13000,66@0=19000,70@1=1000@2=600,70@1=1500@2=800,70@1=2500@2=1200,70@1=4000@2=1600,70@1=2500@2=2304@3=1,70@1=5000@2=2304@3=1,70@1=10800@2=2304@3=1,70@1=10800@2=2560@3=1,106"
timeout 600 python tools/ab_bench.py --workload 4k --rounds 5 --variants "$V4K" \
    --trace "72,72@0=9000,72@0=13000,71@1=2500@2=1200,71@1=2500@2=2304@3=1,71@1=10800@2=2304@3=1" \
    --out $O/ab_4k.json > $O/ab_4k.txt 2>&1
echo "4k rc=$?" >> $O/status.txt
timeout 300 python tools/ab_bench.py --workload 1080p_b4 --rounds 5 \
    --variants "0,66@0=6000,66@0=13000,70@1=1500@2=800,70@1=2500@2=1200,70@1=2500@2=2048@3=1,70@1=8640@2=2048@3=1" \
    --trace "72,71@1=2500@2=1200" --out $O/ab_1080p_b4.json > $O/ab_1080p_b4.txt 2>&1
echo "1080p_b4 rc=$?" >> $O/status.txt
timeout 300 python tools/ab_bench.py --workload hdrp --rounds 5 \
    --variants "0,66@0=6000,70@1=2500@2=1200,70@1=2500@2=2304@3=1" --out $O/ab_hdrp.json > $O/ab_hdrp.txt 2>&1
echo "hdrp rc=$?" >> $O/status.txt
timeout 300 python tools/ab_bench.py --workload 1080p --rounds 5 \
    --variants "0,66,66@0=3000,66@0=6000,28,28@0=3000" --trace "72,48" --out $O/ab_1080p.json > $O/ab_1080p.txt 2>&1
echo "1080p rc=$?" >> $O/status.txt
tail -30 $O/ab_4k.txt
