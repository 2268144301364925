"""The training step's batch feed: the three staging copies in front of the graph on its stream (serial), staged on a copy
stream into the OTHER buffer set while the previous batch's graph replays (GraphedTrainStep(feeds=2).prefetch / .step), and
no copies at all (the graph's own buffers).  Also checks that the pipelined run computes what the serial one does."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdrnet_amd import models, metrics, optim
from hdrnet_amd.runtime import GraphedTrainStep

dev = torch.device("cuda:0")
B, H, W = 4, 1080, 1920
gen = torch.Generator(device=dev).manual_seed(1)
batches = [(torch.rand(B, 256, 256, 3, device=dev, generator=gen), torch.rand(B, H, W, 3, device=dev, generator=gen),
            torch.rand(B, H, W, 3, device=dev, generator=gen)) for _ in range(3)]


def make(feeds):
    torch.manual_seed(0)
    m = models.HDRNetPointwiseNNGuide(dict(batch_norm=False)).to(dev).train()
    opt = optim.FlatAdam([p for p in m.parameters() if p.requires_grad], lr=1e-4, epsilon_hat=True)
    low, full, tgt = batches[0]
    return m, GraphedTrainStep(m, lambda o, t: metrics.l2_loss(t, o), opt, [low, full], [tgt], flat_bucket=True, feeds=feeds)


def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


# equivalence: 6 steps over rotating batches, serial vs pipelined
m1, s1 = make(1)
m2, s2 = make(2)
l1, l2 = [], []
for k in range(6):
    low, full, tgt = batches[k % 3]
    l1.append(float(s1([low, full], [tgt]).detach()))
low, full, tgt = batches[0]
s2.prefetch([low, full], [tgt])
for k in range(6):
    low, full, tgt = batches[(k + 1) % 3]
    s2.prefetch([low, full], [tgt])
    l2.append(float(s2.step().detach()))
s2.step()  # drain
print("losses serial   ", ["%.6f" % v for v in l1])
print("losses pipelined", ["%.6f" % v for v in l2])
# (the drain above made a 7th update in the pipelined model only: compare the losses, which cover 6 identical updates)
assert l1 == l2, "pipelined feed computes something else"

k = [0]
def serial():
    low, full, tgt = batches[k[0] % 3]; k[0] += 1
    s1([low, full], [tgt])
def piped():
    low, full, tgt = batches[k[0] % 3]; k[0] += 1
    s2.prefetch([low, full], [tgt]); s2.step()
low, full, tgt = batches[0]
s2.prefetch([low, full], [tgt])
for r in range(3):
    print("round %d  serial feed %.4f ms | double-buffered feed %.4f ms | static feed %.4f ms" % (
        r, timeit(serial), timeit(piped), timeit(lambda: s1(s1.static_inputs, s1.static_targets))), flush=True)
