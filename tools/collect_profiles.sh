#!/bin/bash
# Collects a round's evidence on an MI355X box into gpurun_out/profiles/ (copy what you want judged into
# profiles/rNN/).  Two calls, each on a fresh box, from the repo root:
#   gpurun --timeout 600  -- 'bash tools/collect_profiles.sh pmc'     counters -> traffic.json; copy it to
#                                                                     profiles/rNN/ BEFORE the second call
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh timing'  every timing, on a box no counter
#                                                                     session has touched (timings taken
#                                                                     after rocprofv3 --pmc passes on the
#                                                                     same box were intermittently 5 us
#                                                                     slower: profiles/r02/README.md)
# Optional: a previous build of the library (e.g. round 2's, built from that commit in a scratch worktree and
# copied to tools/exp/prev/libhdrnet_amd_r02.so -- *.so files are git-ignored but travel to the GPU box) is
# timed interleaved with the tree's build (tools/prev_vs_new.py).
set -u
R=$(pwd)
O=$R/gpurun_out/profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MODE=${1:-timing}
PREV=$R/tools/exp/prev/libhdrnet_amd_r05.so
if [ "$MODE" = pmc ]; then
# 1. HBM traffic of the forward kernel: separate --pmc passes (never combined with trace domains), each
#    with a calibration twin on the memory skeleton (known byte count, same access widths)
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-ops > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-ops > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/cal_fetch -o p --output-format csv -- python $R/tools/ab_bench.py --variants 106 --rounds 1 --steps 20 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/cal_write -o p --output-format csv -- python $R/tools/ab_bench.py --variants 106 --rounds 1 --steps 20 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/pmc_sq -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-ops > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD -d $O/pmc_sq2 -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-ops > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 --match apply_fwd > $O/fwd_pmc.txt 2>&1
python $R/tools/pmc_summary.py $O/cal_fetch $O/cal_write --match skeleton > $O/fwd_pmc_calibration.txt 2>&1
python $R/tools/make_traffic.py --fetch $O/pmc_fetch --write $O/pmc_write --calib-fetch $O/cal_fetch --calib-write $O/cal_write --workload 4k --out $O/traffic.json > $O/traffic.log 2>&1
# the fused gradient pass: where its waves spend their cycles
CMD="python $R/tools/bwd_ab.py --rounds 1 --steps 5 --cases g,gg,all --variants 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/b1 -o p --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/b2 -o p --output-format csv -- $CMD > /dev/null 2>&1
# ... its HBM traffic (the backward's roofline fraction rested on algorithmic bytes alone): FETCH_SIZE / WRITE_SIZE
# of stage 1 and stage 2, separate passes; and the LDS counters again with a smooth guide (the A-operand scatter's bank
# conflicts are a property of a random guide: bank = 4 z + lane, z random)
rocprofv3 --pmc FETCH_SIZE -d $O/b3 -o p --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/b4 -o p --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/b5 -o p --output-format csv -- $CMD --smooth > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/b1 $O/b2 $O/b3 $O/b4 --match grid_grad_stage > $O/bwd_pmc.txt 2>&1
python $R/tools/pmc_summary.py $O/b5 --match grid_grad_stage1 > $O/bwd_pmc_smooth_guide.txt 2>&1
# the fused-guide / wire-format forwards: every apply_fwd_io instantiation and the guide-network
# instantiation of apply_fwd_seg that tools/op_bench.py launches at 4K
CMD="python $R/tools/op_bench.py --workload 4k --steps 5"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/g1 -o p --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD -d $O/g2 -o p --output-format csv -- $CMD > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/g1 $O/g2 --match apply_fwd_io_rows > $O/guide_wire_pmc.txt 2>&1
python $R/tools/pmc_summary.py $O/g1 $O/g2 --match "false, true, false, 1" >> $O/guide_wire_pmc.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 $O/cal_fetch $O/cal_write $O/b1 $O/b2 $O/b3 $O/b4 $O/b5 $O/g1 $O/g2
tail -40 $O/fwd_pmc.txt; tail -25 $O/traffic.log
exit 0
fi
# 2. the default bench command, un-profiled and under rocprofv3 --kernel-trace --stats
python $R/bench.py > $O/bench.json 2> $O/bench.err
# ... and, on the same box, the whole -m gpu suite with its printed error figures (-s) and the smoke entry point
(cd $R && python -m pytest tests -q -m gpu -s 2>&1 | grep -v amdgpu.ids > $O/gpu_suite.txt; python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> $O/gpu_suite.txt 2>&1)
# the same command under rocprofv3 --kernel-trace --stats (its per-kernel average must agree with the line above)
rocprofv3 --kernel-trace --stats -d $O/stats -o fwd --output-format csv -- python $R/bench.py --no-cpu-baseline --no-pipelined > $O/bench_under_rocprof.log 2>&1
grep '^{"metric' $O/bench_under_rocprof.log > $O/bench_under_rocprof.json
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/fwd_kernel_stats.csv \;
rm -rf $O/stats
for i in 1 2 3; do python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-ops; done > $O/bench_steps20.txt 2>> $O/bench.err
python $R/bench.py --workload 1080p --no-cpu-baseline > $O/bench_1080p.json 2>> $O/bench.err
python $R/bench.py --workload 1080p_b4 --no-cpu-baseline > $O/bench_1080p_b4.json 2>> $O/bench.err
python $R/bench.py --workload hdrp --no-cpu-baseline > $O/bench_hdrp.json 2>> $O/bench.err
python $R/bench.py --workload hdrp_u16 > $O/bench_hdrp_u16.json 2>> $O/bench.err
python $R/bench.py --workload train_1080p_b4 --steps 100 --warmup 20 > $O/bench_train_1080p_b4.json 2>> $O/bench.err
python $R/bench.py --workload train_1080p_b4 --batch-norm --steps 100 --warmup 20 > $O/bench_train_1080p_b4_batch_norm.json 2>> $O/bench.err
for i in 1 2 3; do python $R/bench.py --no-cpu-baseline; done > $O/bench_repeat.txt 2>> $O/bench.err
# the smooth-guide pair SURVEY.md section 8d asks for beside U[0,1) (+ the cache-resident rate), same protocol
python $R/bench.py --extra --no-cpu-baseline --no-pipelined > $O/bench_extra_smooth_guide.json 2>> $O/bench.err
# 3. every entry point, all sizes; A/B of the forward variants; this build vs the previous round's; end to end
cd $R
python tools/op_bench.py --tools --workload 4k --json $O/ops_4k.json > $O/ops_4k.txt 2>&1
python tools/op_bench.py --tools --workload 1080p --json $O/ops_1080p.json > $O/ops_1080p.txt 2>&1
# luma_bins = 4 / 16 (hdrnet/bin/train.py:235): the grid depths beside BASELINE.json's 8, random and smooth guide
for lb in 4 16; do
  python tools/op_bench.py --workload 4k --luma-bins $lb > $O/ops_4k_lb$lb.txt 2>&1
  python tools/op_bench.py --workload 4k --luma-bins $lb --smooth-guide --only "apply bwd,apply fwd" > $O/ops_4k_lb${lb}_smooth.txt 2>&1
done
python tools/op_bench.py --workload 1080p_b4 --json $O/ops_1080p_b4.json > $O/ops_1080p_b4.txt 2>&1
python tools/op_bench.py --workload hdrp --json $O/ops_hdrp.json > $O/ops_hdrp.txt 2>&1
python tools/op_bench.py --workload refbench --json $O/ops_refbench.json > $O/ops_refbench.txt 2>&1
python tools/ab_bench.py --variants 0,19,106,108 --rounds 9 --steps 200 > $O/ab_variants_4k.txt 2>&1
python tools/ab_bench.py --workload 1080p --variants 0,19,106,108 --rounds 7 --steps 400 > $O/ab_variants_1080p.txt 2>&1
python tools/ab_bench.py --workload hdrp --variants 0,19,106,108 --rounds 5 --steps 100 > $O/ab_variants_hdrp.txt 2>&1
if [ -f $PREV ]; then
  python tools/prev_vs_new.py --prev $PREV --workload 4k --cases fwd,nn,u8,u8nn,curves,u8curves,upadd,nnupadd,all,gg,g,v,slice_fwd > $O/prev_vs_new_4k.txt 2>&1
  python tools/prev_vs_new.py --prev $PREV --workload 1080p --steps 150 --cases fwd,nn,u8,u8nn,curves,u8curves,all > $O/prev_vs_new_1080p.txt 2>&1
fi
# the guide forwards with the exported arrays and with the parameters prepared once per parameter set (round 5), interleaved
for w in 4k 1080p; do python tools/guide_prepared_ab.py --workload $w 2>&1 | grep -v amdgpu.ids; done > $O/guide_prepared_ab.txt
python tools/bwd_ab.py --rounds 5 --steps 50 --cases all,gg,g,sl,v --variants 0,2,10,3,4,5,6,7,8 > $O/bwd_ab_4k.txt 2>&1
python tools/bwd_ab.py --workload 1080p --rounds 5 --steps 100 --cases all,gg,g,sl,v --variants 0,3 > $O/bwd_ab_1080p.txt 2>&1
python tools/bwd_ab.py --smooth --rounds 5 --steps 50 --cases all,gg,g --variants 0 > $O/bwd_ab_4k_smooth_guide.txt 2>&1
python tools/pyramid_onepass_bench.py --workload 4k --segs 512,768,1024 > $O/pyramid_onepass_4k.txt 2>&1
python tools/e2e_bench.py > $O/e2e.txt 2>&1
# the coefficient network's launches, per-workgroup timeline (tools build)
python tools/coeff_trace.py > $O/coeff_trace.txt 2>&1
# socket power / shader clock beside the forward and its no-compute skeleton; skeleton windows inside the product's run
python tools/power_probe.py --metrics --variants 0,106,0 --seconds 2 2>&1 | grep -v amdgpu.ids > $O/power_probe.txt
python tools/power_probe.py --pattern 0,0,0,106 --seconds 3 2>&1 | grep -v amdgpu.ids | cut -c1-1600 > $O/power_pattern.txt
ls -la $O
cat $O/bench.json
