#!/bin/bash
# Round 4, experiment D: u8 guide network on the bf16 matrix cores -- parity + timings.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04d
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "matrix_cores or wire or nnguide" > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/status.txt
timeout 600 python tools/op_bench.py --workload 4k --json $O/ops_4k.json > $O/ops_4k.txt 2>&1
tail -30 $O/tests.txt; grep -i "nn\|u8" $O/ops_4k.txt
