#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04g
mkdir -p $O
tools/debug/ubench/bin/cvt_pk_u8 > $O/cvt_pk_u8.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "wire or uint16 or curves or guide_network" > $O/tests.txt 2>&1
echo "tests rc=$?" > $O/status.txt
timeout 600 python tools/prev_vs_new.py --prev tools/exp/prev/libhdrnet_amd_r03.so --workload 4k --cases u8,u8nn,curves,u8curves,nn,fwd > $O/prev_vs_new_4k.txt 2>&1
timeout 600 python tools/op_bench.py --workload hdrp --json $O/ops_hdrp.json > $O/ops_hdrp.txt 2>&1
cat $O/cvt_pk_u8.txt; tail -3 $O/tests.txt; cat $O/prev_vs_new_4k.txt | cut -c1-200; grep -i "u16\|u8" $O/ops_hdrp.txt
