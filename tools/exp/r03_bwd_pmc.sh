#!/bin/bash
# Where does the fused gradient pass wait?  Counter passes (own runs, no trace domains) over dgrid / dgrid+dguide / all three.
set -u
R=$(pwd); O=$R/gpurun_out/r03f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
CMD="python $R/tools/bwd_ab.py --rounds 1 --steps 5 --cases g,gg,all --variants 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/p1 -o p --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA -d $O/p2 -o p --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU -d $O/p3 -o p --output-format csv -- $CMD > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/p1 $O/p2 $O/p3 --match grid_grad_stage1 > $O/bwd_pmc.txt 2>&1
rm -rf $O/p1 $O/p2 $O/p3
cat $O/bwd_pmc.txt
