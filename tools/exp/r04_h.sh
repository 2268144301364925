#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04h
mkdir -p $O
export HDRNET_AMD_ALLOW_STALE_LIB=1
timeout 600 python -m pytest tests -m gpu -x -q -s -k "inference_sigmoid" > $O/test_sigmoid.txt 2>&1
timeout 400 python tools/ab_bench.py --workload hdrp --rounds 9 --variants "0,0@4=1,0@4=1@6=1,70@1=2048@2=1024,70@1=2048@2=1024@4=1,106,108" > $O/ab_hdrp.txt 2>&1
tail -4 $O/test_sigmoid.txt; grep -hE "^variant +[0-9]" $O/ab_hdrp.txt | cut -c1-200
