"""How long does the HOST need per launch?  Times bench.run_steps with the GPU kept trivially busy
(a 64x64 image: the kernel is ~2 us), i.e. the pure Python + ctypes + HIP launch cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from hdrnet_amd import _lib
dev = torch.device("cuda:0")
lib = _lib.load()
dims = (64, 64, 16, 16, 8)
sets = bench.make_sets(dev, 3, *dims, seed=1)
stream = torch.cuda.current_stream(dev).cuda_stream
bench.run_steps(lib, sets, dims, stream, 200)
torch.cuda.synchronize()
for rep in range(3):
    t = time.perf_counter()
    bench.run_steps(lib, sets, dims, stream, 5000)
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(f"host loop: {t_issue / 5000 * 1e6:.2f} us per launch issued, {t_all / 5000 * 1e6:.2f} us incl. drain")
