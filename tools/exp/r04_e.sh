#!/bin/bash
# Round 4, E: harness for configs #4 / #5 on one GPU (+ 2 ranks on one GPU over gloo), GPU tests of the touched code.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04e
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s -k "guide_network or train_step or frame_pipeline or wire or curves or dguide_noise" > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/status.txt
timeout 600 python bench.py --workload train_1080p_b4 --steps 50 --warmup 10 > $O/train_1gpu.json 2> $O/train_1gpu.err
echo "train1 rc=$?" >> $O/status.txt
HDRNET_BENCH_BACKEND=gloo timeout 900 python bench.py --workload train_1080p_b4 --gpus 2 --steps 20 --warmup 5 > $O/train_2ranks_one_gpu_gloo.json 2> $O/train_2ranks.err
echo "train2 rc=$?" >> $O/status.txt
timeout 600 python bench.py --workload hdrp_u16 > $O/hdrp_u16_1gpu.json 2> $O/hdrp_u16.err
echo "hdrp_u16 rc=$?" >> $O/status.txt
HDRNET_BENCH_BACKEND=gloo timeout 600 python bench.py --workload hdrp_u16 --gpus 2 --steps 200 --warmup 50 > $O/hdrp_u16_2ranks_one_gpu_gloo.json 2> $O/hdrp2.err
echo "hdrp_u16_2 rc=$?" >> $O/status.txt
tail -5 $O/tests.txt; cat $O/status.txt; cat $O/train_1gpu.json $O/train_2ranks_one_gpu_gloo.json $O/hdrp_u16_1gpu.json | cut -c1-900; for f in $O/*.err; do tail -n 3 $f; done
