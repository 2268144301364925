#!/bin/bash
# backward after the ds_read_b64 operand reads: timing, parity subset, LDS counters
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -m gpu -k "grad or bwd or backward or determin" 2>&1 | tail -3
python tools/bwd_ab.py --rounds 4 --steps 50 --cases all,gg,g,sl --variants 0,4,5,7 2>&1 | grep -v amdgpu.ids | tee $O/exp12_bwd_ablate.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/exp12_pmc -o p --output-format csv -- python $R/tools/bwd_ab.py --rounds 1 --steps 5 --cases all,g --variants 0 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/exp12_pmc --match grid_grad > $O/exp12_pmc.txt 2>&1
rm -rf $O/exp12_pmc
cat $O/exp12_pmc.txt
