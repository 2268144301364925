#!/bin/bash
# PMC passes over the forward kernel / a skeleton (V = ab_bench variant number).  SQ counters only: a
# pass with TA_* / SPI_* counters hung for the full gpurun limit on this pool.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
CMD="python $R/tools/ab_bench.py --variants ${V:-0} --rounds 1 --steps 20"
O=$R/gpurun_out/pmc_fwd_${V:-0}
timeout 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES -d $O/a -o p --output-format csv -- $CMD > /dev/null 2>&1
timeout 90 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM -d $O/b -o p --output-format csv -- $CMD > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/a $O/b --match "${M:-apply_fwd}"
