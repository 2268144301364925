#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -q -x -m gpu -k "nnguide or upadd or pyr or guide or model" 2>&1 | tail -3
python tools/op_bench.py --workload 4k 2>&1 | grep -v amdgpu.ids | head -10 | tee $O/exp18_ops_fwd.txt
python tools/op_bench.py --workload 1080p 2>&1 | grep -v amdgpu.ids | head -10 | tee -a $O/exp18_ops_fwd.txt
