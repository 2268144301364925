#!/bin/bash
R=$(pwd); O=$R/gpurun_out/train; mkdir -p $O
python -m pytest tests/test_models.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
python bench.py --workload train_1080p_b4 --steps 100 --warmup 20 2>/dev/null | cut -c1-330
bash tools/exp/r04_train_prof.sh 2>&1 | tail -34
