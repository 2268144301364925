#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -q -x -m gpu -k "nnguide or upadd or pyr or guide or model or wire" 2>&1 | tail -2
python tools/bwd_ab.py --rounds 4 --steps 50 --cases all,g --variants 0,6,8 2>&1 | grep "^case" | tee $O/exp20_bwd_fixed.txt
python tools/op_bench.py --workload 4k 2>&1 | grep -v amdgpu.ids | sed -n 2,4p
