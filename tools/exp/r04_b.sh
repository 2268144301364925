#!/bin/bash
# Round 4, experiment B: 32-px segment rounding at 4000x3000 (knob 4 = the round-3 cut), the empty-launch floor at
# every size, baselines of every entry point on this box.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04b
mkdir -p $O
export HDRNET_AMD_ALLOW_STALE_LIB=1
timeout 300 python tools/ab_bench.py --workload hdrp --rounds 7 --variants "0,0@4=1,70@1=2048@2=1024,70@1=2048@2=1024@4=1,106,107" \
   --trace "72,72@4=1" > $O/ab_hdrp.txt 2>&1
timeout 300 python tools/ab_bench.py --workload 4k --rounds 5 --variants "0,106,107" > $O/ab_4k.txt 2>&1
timeout 300 python tools/ab_bench.py --workload 1080p --rounds 5 --variants "0,106,107" --trace 72 > $O/ab_1080p.txt 2>&1
timeout 300 python tools/ab_bench.py --workload 1080p_b4 --rounds 5 --variants "0,106,107" > $O/ab_1080p_b4.txt 2>&1
timeout 600 python tools/op_bench.py --workload 4k --json $O/ops_4k.json > $O/ops_4k.txt 2>&1
timeout 600 python tools/op_bench.py --workload hdrp --json $O/ops_hdrp.json > $O/ops_hdrp.txt 2>&1
grep -hE "^variant +[0-9]" $O/ab_*.txt | cut -c1-200
