#!/bin/bash
# Round 4, experiment A2: ticketed tail, second form (32 pools, one draw per workgroup, self-zeroing).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04a2
mkdir -p $O
export HDRNET_AMD_ALLOW_STALE_LIB=1
V4K="0,70@1=0,70@1=512@2=512,70@1=1024@2=512,70@1=1024@2=1024,70@1=1536@2=1024,70@1=2048@2=1024,70@1=2048@2=2048,70@1=3072@2=1536,106"
timeout 600 python tools/ab_bench.py --workload 4k --rounds 7 --variants "$V4K" \
    --trace "72,71@1=0,71@1=1024@2=1024,71@1=2048@2=1024,71@1=2048@2=2048" \
    --out $O/ab_4k.json > $O/ab_4k.txt 2>&1
echo "4k rc=$?" >> $O/status.txt
timeout 300 python tools/ab_bench.py --workload hdrp --rounds 5 \
    --variants "0,70@1=0,70@1=1024@2=1024,70@1=2048@2=1024,70@1=3072@2=1536" --trace "72,71@1=2048@2=1024" --out $O/ab_hdrp.json > $O/ab_hdrp.txt 2>&1
echo "hdrp rc=$?" >> $O/status.txt
timeout 300 python tools/ab_bench.py --workload 1080p_b4 --rounds 5 \
    --variants "0,70@1=0,70@1=1024@2=1024,70@1=2048@2=1024" --trace "72,71@1=1024@2=1024" --out $O/ab_1080p_b4.json > $O/ab_1080p_b4.txt 2>&1
echo "1080p_b4 rc=$?" >> $O/status.txt
grep -E "^variant +[0-9]" $O/ab_4k.txt | cut -c1-200
