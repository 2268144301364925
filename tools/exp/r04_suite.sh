#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04suite
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -s > $O/gpu_suite.txt 2>&1
echo "suite rc=$?" > $O/status.txt
grep -E "passed|failed" $O/gpu_suite.txt | tail -3; grep -E "HIP - f64|sigmoid\)" $O/gpu_suite.txt | cut -c1-300
