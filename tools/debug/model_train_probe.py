"""A graph-captured training step of one model class at 4 x 1080p, replayed N times (run it under rocprofv3 --kernel-trace and
feed the trace to tools/train_step_profile.py):  python tools/debug/model_train_probe.py HDRNetCurves [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdrnet_amd import models, metrics, optim
from hdrnet_amd.runtime import GraphedTrainStep

cls = getattr(models, sys.argv[1] if len(sys.argv) > 1 else "HDRNetCurves")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
B, H, W = 4, 1080, 1920
torch.manual_seed(0)
low = torch.rand(B, 256, 256, 3, device=dev); full = torch.rand(B, H, W, 3, device=dev); tgt = torch.rand(B, H, W, 3, device=dev)
m = cls(dict(batch_norm=False)).to(dev).train()
opt = optim.FlatAdam([p for p in m.parameters() if p.requires_grad], lr=1e-4, epsilon_hat=True)
step = GraphedTrainStep(m, lambda o, t: metrics.l2_loss(t, o), opt, [low, full], [tgt], flat_bucket=True)
for _ in range(10): step([low, full], [tgt])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step([low, full], [tgt])
torch.cuda.synchronize()
print("%s: %.4f ms/step" % (cls.__name__, (time.perf_counter() - t0) / steps * 1e3))
