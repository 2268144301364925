cd /tmp && export TMPDIR=/tmp
R=/root/repo
CMD="python $R/tools/debug/dgrid_ab.py --variants ${V:-5} --rounds 1 --steps 5"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d $R/gpurun_out/pmc_gg1 -o p1 --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc_gg2 -o p2 --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d $R/gpurun_out/pmc_gg3 -o p3 --output-format csv -- $CMD > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_gg1 $R/gpurun_out/pmc_gg2 $R/gpurun_out/pmc_gg3 --match stage1
