#!/bin/bash
# Collects the round's evidence on an MI355X box into gpurun_out/profiles/ (copy what you want judged
# into profiles/rNN/).  Run from the repo root:  gpurun --timeout 900 -- 'bash tools/collect_profiles.sh'
set -u
R=$(pwd)
O=$R/gpurun_out/profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the default bench command, un-profiled and under rocprofv3 --kernel-trace --stats
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats -o fwd --output-format csv -- python $R/bench.py > $O/bench_under_rocprof.log 2>&1
grep '^{"metric' $O/bench_under_rocprof.log > $O/bench_under_rocprof.json
cp $O/stats/*kernel_stats.csv $O/fwd_kernel_stats.csv 2>/dev/null || find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/fwd_kernel_stats.csv \;
# 2. HBM traffic of the forward kernel: separate --pmc passes (never combined with trace domains)
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $O/pmc_sq -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq --match apply_fwd > $O/fwd_pmc.txt 2>&1
# 3. every entry point, both sizes; A/B of the forward variants; end-to-end configs
cd $R
python tools/op_bench.py --workload 4k --json $O/ops_4k.json > $O/ops_4k.txt 2>&1
python tools/op_bench.py --workload 1080p --json $O/ops_1080p.json > $O/ops_1080p.txt 2>&1
python tools/ab_bench.py --variants 0,2,3,7,8,9,11,101,103,104,105,106 --rounds 5 --steps 100 > $O/ab_variants_4k.txt 2>&1
python tools/e2e_bench.py > $O/e2e.txt 2>&1
# 4. device micro-benchmarks
for b in stream_patterns valu_rates mfma_valu_overlap; do
  ./tools/debug/ubench/bin/$b > $O/ubench_$b.txt 2>&1
done
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
ls -la $O
tail -3 $O/fwd_pmc.txt; cat $O/bench.json
