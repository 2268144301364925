#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -q -x -m gpu -k "nnguide or upadd or pyr or guide or model or wire or curves" 2>&1 | tail -2
python tools/op_bench.py --workload 4k 2>&1 | grep -v amdgpu.ids | sed -n 2,10p | tee $O/exp30_packed_guides.txt
python tools/op_bench.py --workload 1080p 2>&1 | grep -v amdgpu.ids | sed -n 2,8p | tee -a $O/exp30_packed_guides.txt
