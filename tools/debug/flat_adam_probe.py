import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdrnet_amd import models, metrics, optim
from hdrnet_amd.runtime import GraphedTrainStep
dev = torch.device("cuda:0")
B, H, W = 4, 1080, 1920
low = torch.rand(B, 256, 256, 3, device=dev); full = torch.rand(B, H, W, 3, device=dev); tgt = torch.rand(B, H, W, 3, device=dev)
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for bn, kind in ((False, "torch"), (False, "flat"), (True, "torch"), (True, "flat"), (False, "torch"), (False, "flat")):
    torch.manual_seed(0)
    m = models.HDRNetPointwiseNNGuide(dict(batch_norm=bn)).to(dev).train()
    ps = [p for p in m.parameters() if p.requires_grad]
    opt = optim.FlatAdam(ps, lr=1e-4) if kind == "flat" else torch.optim.Adam(ps, lr=1e-4, capturable=True, fused=True)
    step = GraphedTrainStep(m, lambda o, t: metrics.l2_loss(t, o), opt, [low, full], [tgt], flat_bucket=True)
    si, st = step.static_inputs, step.static_targets
    print("bn" if bn else "no-bn", kind, "copy feed %.4f ms" % timeit(lambda: step([low, full], [tgt])), "| static feed %.4f" % timeit(lambda: step(si, st)),
          "| replay only %.4f" % timeit(lambda: step.graph.replay()),
          "| optimizer only %.4f" % timeit(lambda: opt.step()),
          "| host time of optimizer.step %.1f us" % (sum((lambda t0: (opt.step(), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(50)) / 50 * 1e6))
