import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hdrnet_amd import models
from hdrnet_amd.runtime import GraphedTrainStep
dev = torch.device("cuda:0")
torch.manual_seed(4)
low = torch.rand(2, 256, 256, 3, device=dev)
full = torch.rand(2, 136, 240, 3, device=dev)
target = torch.rand(2, 136, 240, 3, device=dev)
loss_fn = lambda out, tgt: (out - tgt).square().mean()
m0 = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev).train()
state = {k: v.clone() for k, v in m0.state_dict().items()}
def make(cap):
    m = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev).train()
    m.load_state_dict(state)
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=1e-3, momentum=0.9)
    return m, opt
for cap in (False, True):
    me, oe = make(cap)
    ls = []
    for _ in range(6):
        oe.zero_grad(set_to_none=True)
        le = loss_fn(me(low, full), target)
        le.backward()
        oe.step()
        ls.append(round(le.item(), 5))
    print("eager capturable=%s" % cap, ls)
mg, og = make(True)
class Spy(GraphedTrainStep):
    def _eager_step(self):
        l = super()._eager_step()
        print("  warmup loss", round(l.item(), 5))
        return l
g = Spy(mg, loss_fn, og, [low, full], [target], warmup=3)
print("graphed", [round(g([low, full], [target]).item(), 5) for _ in range(3)])
