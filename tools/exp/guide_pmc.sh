#!/bin/bash
# Counter passes over the guide forwards, ONE parameter form per pass (tools/guide_prepared_ab.py --only): VALU / SALU / LDS
# instructions per launch of the guide-network kernels with the exported and with the prescaled parameters, of the curves
# kernels with the sorted tables and with the prepared cells.  Output: gpurun_out/guide_pmc/{exported,prepared}.txt
R=$(pwd); O=$R/gpurun_out/guide_pmc; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for form in exported prepared; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $O/$form -o p --output-format csv -- python $R/tools/guide_prepared_ab.py --workload 4k --only $form > /dev/null 2>&1
  python $R/tools/pmc_summary.py $O/$form --match apply_fwd > $O/$form.txt 2>&1
  rm -rf $O/$form
done
grep -h "^apply\|SQ_INSTS_VALU\|SQ_WAVES\|SQ_INSTS_LDS" $O/exported.txt $O/prepared.txt
