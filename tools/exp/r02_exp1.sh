#!/bin/bash
# Round-2 experiment 1: parity of the apply_fwd_seg knobs, A/B timing vs the round-1 kernel, WG timelines, PMC.
set -u
R=$(pwd)
O=$R/gpurun_out/r02_exp1
mkdir -p $O
export HDRNET_AMD_KERNEL_NAMES=1
timeout 600 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_apply_forward_benchmark_variants > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "benchmark_variants or no_benchmark" > $O/pytest_variants.log 2>&1; echo "pytest rc=$?" >> $O/pytest_variants.log
tail -15 $O/pytest_variants.log
for wl in 4k 1080p; do
  timeout 300 python tools/ab_bench.py --workload $wl --variants 0,8,105,106 --rounds 5 --steps 100 --out $O/ab_${wl}_a.json > $O/ab_${wl}_a.txt 2>&1
  timeout 300 python tools/ab_bench.py --workload $wl --variants 0,20,21 --rounds 5 --steps 100 --trace 36,37 --out $O/ab_${wl}_b.json > $O/ab_${wl}_b.txt 2>&1
  timeout 300 python tools/ab_bench.py --workload $wl --variants 0,22,23 --rounds 5 --steps 100 --trace 39 --out $O/ab_${wl}_c.json > $O/ab_${wl}_c.txt 2>&1
  timeout 300 python tools/ab_bench.py --workload $wl --variants 0,24,25,28,29 --rounds 5 --steps 100 --trace 40,41,45 --out $O/ab_${wl}_d.json > $O/ab_${wl}_d.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc_sq -o p --output-format csv -- python $R/tools/ab_bench.py --variants 0,20,21,23,24,25 --rounds 1 --steps 3 > $O/pmc_run.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq --match apply_fwd > $O/pmc_sq.txt 2>&1
rm -rf $O/pmc_sq
cd $R
grep -h "^variant\|trace\|starts\|lifetime\|ends:" $O/ab_4k_*.txt | grep -v "max|fast" | head -60
