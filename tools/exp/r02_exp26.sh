#!/bin/bash
# Box state while the forward runs: clocks / power sampled DURING a 6-s launch loop, + the speed of three flavours
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
T=$(date +%H%M%S)
{
rocminfo 2>/dev/null | grep -i "Uuid" | grep GPU | head -1
rocm-smi --showperflevel --showmaxpower 2>&1 | grep -i "level\|max" | tr -s " " | cut -c1-100
python - <<'PY' &
import time, torch, sys
sys.path.insert(0, '.')
from bench import WORKLOADS, make_sets
from hdrnet_amd import _lib
lib = _lib.load()
H, W, GH, GW, GD, desc = WORKLOADS['4k']
dev = torch.device('cuda:0')
S = make_sets(dev, 3, H, W, GH, GW, GD, 1)
st = torch.cuda.current_stream(dev).cuda_stream
t0 = time.time()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
while time.time() - t0 < 7:
    e0.record()
    for k in range(3000):
        g, gu, i, o = S[k % 3]
        lib.hdrnet_bilateral_slice_apply_f32(g.data_ptr(), gu.data_ptr(), i.data_ptr(), o.data_ptr(), 1, H, W, GH, GW, GD, 3, 3, 1, st)
    e1.record(); torch.cuda.synchronize()
    print(f"  loop t={time.time()-t0:4.1f}s  {e0.elapsed_time(e1)/3000*1e3:.2f} us/launch", flush=True)
PY
sleep 3.5
for i in 1 2 3; do rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|mclk\|fclk\|Power (W)\|junction" | tr -s ' ' | cut -c1-80 | tr '\n' ';'; echo; sleep 0.7; done
wait
python tools/ab_bench.py --variants 39,8,23,106 --rounds 3 --steps 200 2>&1 | grep "^variant" | grep median | cut -c1-150
} 2>&1 | grep -v amdgpu.ids | tee $O/exp26_boxstate_$T.txt
