#!/bin/bash
# Does the forward's speed depend on how long the box has been under load?  (cache-policy stores vs plain)
set -u
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
smi() { rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk\|Power (W)\|Temperature (Sensor junction)\|Temperature (Sensor memory)" | tr -s ' ' | cut -c1-90; }
{
echo "== t=0 (fresh box)"; smi
python tools/ab_bench.py --variants 23,31,39,106 --rounds 3 --steps 100 2>&1 | grep "^variant" | grep median | cut -c1-150
echo "== sustained load: 90 s of back-to-back forward launches"
python - <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
from bench import WORKLOADS, make_sets
from hdrnet_amd import _lib
lib = _lib.load()
H, W, GH, GW, GD, desc = WORKLOADS['4k']
dev = torch.device('cuda:0')
S = make_sets(dev, 3, H, W, GH, GW, GD, 1)
st = torch.cuda.current_stream(dev).cuda_stream
t0 = time.time(); n = 0
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
while time.time() - t0 < 90:
    e0.record()
    for k in range(2000):
        g, gu, i, o = S[k % 3]
        lib.hdrnet_bilateral_slice_apply_f32(g.data_ptr(), gu.data_ptr(), i.data_ptr(), o.data_ptr(), 1, H, W, GH, GW, GD, 3, 3, 1, st)
    e1.record(); torch.cuda.synchronize()
    n += 1
    if n % 25 == 1: print(f"  t={time.time()-t0:5.1f}s  {e0.elapsed_time(e1)/2000*1e3:.2f} us/launch", flush=True)
PY
echo "== after load"; smi
python tools/ab_bench.py --variants 23,31,39,106 --rounds 3 --steps 100 2>&1 | grep "^variant" | grep median | cut -c1-150
echo "== idle 20 s"; sleep 20; smi
python tools/ab_bench.py --variants 23,31,39,106 --rounds 3 --steps 100 2>&1 | grep "^variant" | grep median | cut -c1-150
} 2>&1 | grep -v amdgpu.ids | tee $O/exp24_sustained_load.txt
