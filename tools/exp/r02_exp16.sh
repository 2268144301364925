#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -m gpu -k "grad or bwd or backward or determin" 2>&1 | tail -3
{
python tools/bwd_ab.py --rounds 4 --steps 50 --cases all,gg,g,sl --variants 0,2,4,5,7 2>&1 | grep "^case"
python tools/bwd_ab.py --workload 1080p --rounds 4 --steps 100 --cases all,gg,g,sl --variants 0 2>&1 | grep "^case"
} | tee $O/exp16_bwd.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/exp16_pmc -o p --output-format csv -- python $R/tools/bwd_ab.py --rounds 1 --steps 5 --cases all,g --variants 0 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/exp16_pmc --match grid_grad_stage1 > $O/exp16_pmc.txt 2>&1
rm -rf $O/exp16_pmc
cat $O/exp16_pmc.txt
