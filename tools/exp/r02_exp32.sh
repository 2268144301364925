#!/bin/bash
# FP32-pipe occupancy of the backward: MFMA busy cycles and VALU instruction counts per launch (final sources)
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -d $O/exp32_pmc -o p --output-format csv -- python $R/tools/bwd_ab.py --rounds 1 --steps 5 --cases g,gg,all --variants 0 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/exp32_pmc --match grid_grad_stage1 > $O/exp32_bwd_fp32_pipe.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/exp32_pmc2 -o p --output-format csv -- python $R/tools/bwd_ab.py --rounds 1 --steps 5 --cases g,gg,all --variants 0 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/exp32_pmc2 --match grid_grad_stage1 >> $O/exp32_bwd_fp32_pipe.txt 2>&1
rm -rf $O/exp32_pmc $O/exp32_pmc2
cat $O/exp32_bwd_fp32_pipe.txt
