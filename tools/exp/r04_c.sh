#!/bin/bash
# Round 4, experiment C: curves guide as a table lookup -- parity tests + timings.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04c
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "curves or wire or frame_pipeline" > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/status.txt
timeout 600 python tools/op_bench.py --workload 4k --json $O/ops_4k.json > $O/ops_4k.txt 2>&1
timeout 600 python tools/op_bench.py --workload 1080p --json $O/ops_1080p.json > $O/ops_1080p.txt 2>&1
tail -5 $O/tests.txt; grep -i "curves\|u8" $O/ops_4k.txt
