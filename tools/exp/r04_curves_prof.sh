#!/bin/bash
R=$PWD; O=$R/gpurun_out/curves; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for m in HDRNetCurves HDRNetGaussianPyrNN; do
rocprofv3 --kernel-trace --stats -d $O/stats_$m -o tr --output-format csv -- python $R/tools/debug/model_train_probe.py $m 40 > $O/run_$m.txt 2>&1
f=$(find $O/stats_$m -name '*kernel_trace.csv' | head -1)
python $R/tools/train_step_profile.py $f --list --anchor l2_loss_partial > $O/step_$m.txt 2>&1
rm -rf $O/stats_$m
done
tail -1 $O/run_HDRNetCurves.txt; head -24 $O/step_HDRNetCurves.txt | cut -c1-120; tail -1 $O/run_HDRNetGaussianPyrNN.txt; head -30 $O/step_HDRNetGaussianPyrNN.txt | cut -c1-120
