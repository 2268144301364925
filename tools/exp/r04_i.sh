#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1
echo "tests rc=$?" > $O/status.txt
timeout 600 python tools/prev_vs_new.py --prev tools/exp/prev/libhdrnet_amd_r03.so --workload 4k --rounds 9 --cases fwd,nn,upadd > $O/prev_vs_new_4k.txt 2>&1
timeout 600 python tools/prev_vs_new.py --prev tools/exp/prev/libhdrnet_amd_r03.so --workload 1080p_b4 --rounds 9 --cases fwd,nn > $O/prev_vs_new_b4.txt 2>&1
timeout 600 python tools/prev_vs_new.py --prev tools/exp/prev/libhdrnet_amd_r03.so --workload hdrp --rounds 9 --cases fwd,nn > $O/prev_vs_new_hdrp.txt 2>&1
tail -2 $O/tests.txt; cat $O/prev_vs_new_*.txt | grep -v amdgpu | cut -c1-170
