#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -q -x -m gpu 2>&1 | tail -4
python tools/op_bench.py --workload 4k 2>&1 | grep -v amdgpu.ids | head -10 | tee $O/exp19_ops_fwd.txt
python tools/op_bench.py --workload 1080p 2>&1 | grep -v amdgpu.ids | head -10 | tee -a $O/exp19_ops_fwd.txt
python tools/op_bench.py --workload hdrp 2>&1 | grep -v amdgpu.ids | head -11 | tee -a $O/exp19_ops_fwd.txt
