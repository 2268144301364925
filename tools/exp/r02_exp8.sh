#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r02_exp8; mkdir -p $O
cat > /tmp/bwd_only.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["R"])
import torch
from hdrnet_amd import _lib
lib = _lib.load_tools()
dev = torch.device("cuda:0")
H, W, GH, GW, GD = 2160, 3840, 16, 16, 8
g = torch.Generator(device=dev).manual_seed(1)
grid = torch.rand((1, GH, GW, GD, 12), device=dev, generator=g)
guide = torch.rand((1, H, W), device=dev, generator=g)
inp = torch.rand((1, H, W, 3), device=dev, generator=g)
dout = torch.randn((1, H, W, 3), device=dev, generator=g)
dgrid = torch.empty_like(grid); dguide = torch.empty_like(guide); dinput = torch.empty_like(inp)
wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(1, H, W, GH, GW, GD, 3, 3, 1)
ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for variant in (0, 3):
    for _ in range(3):
        rc = lib.hdrnet_bilateral_slice_apply_grad_f32_ex(grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), dout.data_ptr(),
            dgrid.data_ptr(), dguide.data_ptr(), dinput.data_ptr(), 1, H, W, GH, GW, GD, 3, 3, 1, ws.data_ptr(), wsb, variant << 8, st)
        assert rc == 0
torch.cuda.synchronize()
PY
export R
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc1 -o p --output-format csv -- python /tmp/bwd_only.py > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY -d $O/pmc2 -o p --output-format csv -- python /tmp/bwd_only.py > $O/pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python /tmp/bwd_only.py > $O/kt.log 2>&1
python $R/tools/pmc_summary.py $O/pmc1 $O/pmc2 > $O/pmc.txt 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-200 | head -12
rm -rf $O/pmc1 $O/pmc2 $O/kt
grep -v "stage2" $O/pmc.txt | head -70
