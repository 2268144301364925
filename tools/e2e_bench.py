#!/usr/bin/env python3
"""End-to-end timings of the reference's model graphs with the HIP hot path inside
(BASELINE.json configs #3 and #4 on ONE MI355X; the 8-GPU runs are the driver's).

    python tools/e2e_bench.py [--steps 20]

config #3: HDRNetPointwiseNNGuide inference, 3840x2160, batch 1 (coefficient net + guide in
           PyTorch-ROCm ops, slice-apply in the HIP kernel).
config #4: training step (fwd + bwd + Adam) at 1920x1080, 4 images per GPU (= 32 / 8).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from hdrnet_amd import dist as hd  # noqa: E402
from hdrnet_amd import metrics, models, optim  # noqa: E402


def timeit(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)

    m = models.HDRNetPointwiseNNGuide().to(dev).eval()
    low = torch.rand(1, 256, 256, 3, device=dev)
    full = torch.rand(1, 2160, 3840, 3, device=dev)
    with torch.no_grad():
        t_all = timeit(lambda: m(low, full), args.steps)
        t_coef = timeit(lambda: m.coefficients(low), args.steps)
        t_guide = timeit(lambda: m.guide(full), args.steps)
        coeffs, guide = m.coefficients(low), m.guide(full)
        from hdrnet_amd import layers
        t_slice = timeit(lambda: layers.bilateral_slice_apply(coeffs, guide, full, has_offset=True), args.steps)
        # the whole inference captured in one hipGraph: the coefficient network is ~25 tiny
        # launches on a 256 x 256 tensor -- launch-bound in eager mode
        from hdrnet_amd.runtime import GraphedInference
        graphed = GraphedInference(m, [low, full])
        t_graph = timeit(lambda: graphed(graphed.static_inputs[0], graphed.static_inputs[1]), args.steps)
        assert torch.allclose(graphed(low, full), m(low, full), rtol=1e-5, atol=1e-5)
        # independent frames round-robin over 2 / 3 streams (runtime.FramePipeline), each lane its own captured graph
        from hdrnet_amd.runtime import FramePipeline
        t_pipe = {}
        for depth in (2, 3):
            pipe = FramePipeline(lambda: GraphedInference(m, [low, full]), depth=depth)
            lanes = pipe.lanes

            def frame():
                lane = pipe.next
                pipe.submit(lanes[lane].static_inputs[0], lanes[lane].static_inputs[1])

            t_pipe[depth] = timeit(frame, args.steps)
        # the coefficient network alone: HIP kernels (csrc/coeff_net.hip) vs the stock PyTorch-ROCm ops, eager and
        # as a hipGraph; then the whole inference with the stock-op network for comparison
        net = m.coefficients
        gn = GraphedInference(net, [low])
        t_coef_graph = timeit(lambda: gn(gn.static_inputs[0]), args.steps * 5)
        net.native = False
        t_coef_stock = timeit(lambda: net(low), args.steps)
        gs = GraphedInference(net, [low])
        t_coef_stock_graph = timeit(lambda: gs(gs.static_inputs[0]), args.steps * 5)
        graphed_stock = GraphedInference(m, [low, full])
        t_graph_stock = timeit(lambda: graphed_stock(graphed_stock.static_inputs[0], graphed_stock.static_inputs[1]), args.steps)
        net.native = True
    mp = 2160 * 3840 / 1e6
    print(f"coefficient network 256x256 b=1: HIP kernels {t_coef * 1e6:.1f} us eager / {t_coef_graph * 1e6:.1f} us as a hipGraph; "
          f"stock ops {t_coef_stock * 1e6:.1f} us eager / {t_coef_stock_graph * 1e6:.1f} us as a hipGraph")
    print(f"config #3  (hipGraph replay of the whole inference): {t_graph * 1e3:.3f} ms/frame = {mp / t_graph:.0f} MP/s"
          f"   [with the stock-op coefficient network: {t_graph_stock * 1e3:.3f} ms/frame]")
    print(f"config #3  the same, frames round-robin over 2 / 3 streams: {t_pipe[2] * 1e3:.3f} / {t_pipe[3] * 1e3:.3f} ms/frame"
          f" = {mp / t_pipe[2]:.0f} / {mp / t_pipe[3]:.0f} MP/s")
    print(f"config #3  HDRNetPointwiseNNGuide 3840x2160 b=1: {t_all * 1e3:.3f} ms/frame = {mp / t_all:.0f} MP/s "
          f"(coefficients {t_coef * 1e3:.3f} ms, guide net {t_guide * 1e3:.3f} ms, slice-apply {t_slice * 1e3:.3f} ms)")

    # HDRNetCurves (the reference's default model) inference at 4K: curves guide fused vs composed
    mc = models.HDRNetCurves().to(dev).eval()
    with torch.no_grad():
        t_cf = timeit(lambda: mc(low, full), args.steps)
        gc = GraphedInference(mc, [low, full])
        t_cg = timeit(lambda: gc(gc.static_inputs[0], gc.static_inputs[1]), args.steps)
        mc.fuse_guide = False
        t_cu = timeit(lambda: mc(low, full), 3)
        mc.fuse_guide = True
    print(f"curves     HDRNetCurves 3840x2160 b=1: composed {t_cu * 1e3:.2f} ms/frame, fused {t_cf * 1e3:.3f} ms/frame, "
          f"fused + hipGraph {t_cg * 1e3:.3f} ms/frame = {mp / t_cg:.0f} MP/s")

    # HDRNetGaussianPyrNN inference at 4K: 3 levels, fused (resize kernel + one launch per level)
    # vs composed from the un-fused ops
    mp_ = models.HDRNetGaussianPyrNN().to(dev).eval()
    with torch.no_grad():
        t_pf = timeit(lambda: mp_(low, full), args.steps)
        gp = GraphedInference(mp_, [low, full])
        t_pg = timeit(lambda: gp(gp.static_inputs[0], gp.static_inputs[1]), args.steps)
        mp_.fuse_guide = False
        t_pu = timeit(lambda: mp_(low, full), 3)
        mp_.fuse_guide = True
    print(f"pyramid    HDRNetGaussianPyrNN 3840x2160 b=1: composed {t_pu * 1e3:.2f} ms/frame, fused "
          f"{t_pf * 1e3:.3f} ms/frame, fused + hipGraph {t_pg * 1e3:.3f} ms/frame = {mp / t_pg:.0f} MP/s")

    mt = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev).train()
    opt = torch.optim.Adam([p for p in mt.parameters() if p.requires_grad], lr=1e-4)
    B = 4
    low = torch.rand(B, 256, 256, 3, device=dev)
    full = torch.rand(B, 1080, 1920, 3, device=dev)
    target = torch.rand(B, 1080, 1920, 3, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = metrics.l2_loss(target, mt(low, full))
        loss.backward()
        hd.allreduce_gradients_flat(mt.parameters())
        opt.step()

    t_step = timeit(step, max(5, args.steps // 2))
    print(f"config #4  training step 1920x1080, {B} images/GPU: {t_step * 1e3:.2f} ms/step = "
          f"{B * 1080 * 1920 / 1e6 / t_step:.0f} MP/s per GPU")

    # the same step with the guide network composed from torch ops (what round 1 first measured)
    mt.fuse_guide = False
    t_unfused = timeit(step, 3)
    mt.fuse_guide = True
    print(f"config #4  same step, guide network un-fused (torch ops): {t_unfused * 1e3:.2f} ms/step")

    # the reference's default model (HDRNetCurves), same step: fused curves guide vs composed
    mcv = models.HDRNetCurves(dict(batch_norm=True)).to(dev).train()
    optc = torch.optim.Adam([p for p in mcv.parameters() if p.requires_grad], lr=1e-4)

    def step_curves():
        optc.zero_grad(set_to_none=True)
        metrics.l2_loss(target, mcv(low, full)).backward()
        optc.step()

    t_cs = timeit(step_curves, max(5, args.steps // 2))
    mcv.fuse_guide = False
    t_csu = timeit(step_curves, 3)
    mcv.fuse_guide = True
    print(f"curves     HDRNetCurves training step 1920x1080, {B} images/GPU: composed {t_csu * 1e3:.2f} ms/step, "
          f"fused {t_cs * 1e3:.2f} ms/step")

    # the whole step (fwd + loss + bwd + Adam) as one hipGraph
    from hdrnet_amd.runtime import GraphedTrainStep
    t_graph = {}
    for bn, native in ((True, True), (False, False), (False, True)):
        mg = models.HDRNetPointwiseNNGuide(dict(batch_norm=bn)).to(dev).train()
        mg.coefficients.native_training = native
        optg = optim.FlatAdam([p for p in mg.parameters() if p.requires_grad], lr=1e-4, epsilon_hat=True)
        gstep = GraphedTrainStep(mg, lambda out, tgt: metrics.l2_loss(tgt, out), optg, [low, full], [target])
        t_graph[(bn, native)] = timeit(lambda: gstep([low, full], [target]), max(5, args.steps // 2))
    t = t_graph[(False, True)]
    print(f"config #4  the whole step as one hipGraph, no batch norm (the reference's scripts), coefficient network's "
          f"forward + backward on the HIP kernels: {t * 1e3:.3f} ms/step = {B * 1080 * 1920 / 1e6 / t:.0f} MP/s per GPU;  "
          f"on stock ops: {t_graph[(False, False)] * 1e3:.3f};  with batch norm (stock ops): {t_graph[(True, True)] * 1e3:.3f}")

    # the reference's DEFAULT model class (hdrnet/bin/train.py:225: models.__all__[0] = HDRNetCurves) and the pyramid model,
    # the same graph-captured step without batch norm
    for cls in (models.HDRNetCurves, models.HDRNetGaussianPyrNN):
        mg = cls(dict(batch_norm=False)).to(dev).train()
        optg = optim.FlatAdam([p for p in mg.parameters() if p.requires_grad], lr=1e-4, epsilon_hat=True)
        gstep = GraphedTrainStep(mg, lambda out, tgt: metrics.l2_loss(tgt, out), optg, [low, full], [target])
        t = timeit(lambda: gstep([low, full], [target]), max(5, args.steps // 2))
        print(f"{cls.__name__:<22s} the whole training step as one hipGraph, no batch norm, {B} x 1080p: {t * 1e3:.3f} ms/step = "
              f"{B * 1080 * 1920 / 1e6 / t:.0f} MP/s per GPU  (coefficient network native: "
              f"{bool(mg.coefficients._use_native_training(low))})")


if __name__ == "__main__":
    main()
