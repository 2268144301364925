"""hipGraph replay of a whole inference (MI355X-first runtime piece: HIP graphs instead of a
tracing compiler).

The coefficient network of HDRNet is ~25 tiny launches on a 256 x 256 tensor -- launch-bound in
eager mode (0.5 ms of host time against 0.05 ms of slice-apply at 4K).  ``GraphedInference``
captures one forward of a module -- stock PyTorch-ROCm ops and the hand-written HIP kernels
alike, all of which launch on the capturing stream -- into a hipGraph and replays it per frame
into static buffers.  (The reference rebuilds a TF session graph for the same purpose,
``hdrnet/bin/run.py:70-95``; its desktop renderer keeps the grid upload + one draw call,
``benchmark/src/renderer.cc:119-171``.)
"""
from __future__ import annotations

from typing import Sequence

import torch


class GraphedInference:
    """Capture ``module(*example_inputs)`` once; ``__call__`` copies new inputs into the static
    buffers, replays the graph and returns the static output (valid until the next call).

    Shapes are fixed at capture time; a call with different shapes raises ``ValueError``.
    """

    def __init__(self, module: torch.nn.Module, example_inputs: Sequence[torch.Tensor], warmup: int = 3):
        if not all(t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedInference needs device tensors (MI355X)")
        self.module = module.eval()
        self.static_inputs = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream(device=self.static_inputs[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):  # library loading, allocator warm-up, kernel selection
                self.module(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_output = self.module(*self.static_inputs)

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        if len(inputs) != len(self.static_inputs):
            raise ValueError(f"expected {len(self.static_inputs)} inputs, got {len(inputs)}")
        for dst, src in zip(self.static_inputs, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_output


class GraphedTrainStep:
    """One training step -- forward, loss, backward through the HIP VJPs, (at world size 1) the
    optimizer update -- captured into a hipGraph and replayed per batch.

    With the guide network fused, a config #4 step (4 x 1080p per GPU) is ~1.7 ms of GPU work
    behind ~3.7 ms of host-side launches of the coefficient network's small ops; the graph removes
    the host from the loop.  The reference's equivalent is the TF session running its static
    training graph (``hdrnet/bin/train.py:156-226``).

    ``loss_fn(output, *targets) -> scalar``.  ``optimizer`` must be capturable (e.g.
    ``torch.optim.Adam(..., capturable=True)``).  In a multi-process job (``torch.distributed``
    initialised, world size > 1) only forward + backward are captured; ``__call__`` then runs the
    flat-bucket gradient all-reduce (``dist.allreduce_gradients_flat``) and the optimizer step
    eagerly, so the collective stays outside the graph.
    """

    def __init__(self, module: torch.nn.Module, loss_fn, optimizer: torch.optim.Optimizer,
                 example_inputs: Sequence[torch.Tensor], example_targets: Sequence[torch.Tensor],
                 warmup: int = 3):
        from . import dist as hd
        if not all(t.is_cuda for t in list(example_inputs) + list(example_targets)):
            raise RuntimeError("GraphedTrainStep needs device tensors (MI355X)")
        self.module, self.loss_fn, self.optimizer = module, loss_fn, optimizer
        self.static_inputs = [t.clone() for t in example_inputs]
        self.static_targets = [t.clone() for t in example_targets]
        self._hd = hd
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1
        side = torch.cuda.Stream(device=self.static_inputs[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # also materialises the optimizer state before capture
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.static_loss = self.loss_fn(self.module(*self.static_inputs), *self.static_targets)
            self.static_loss.backward()
            if not self.distributed:
                self.optimizer.step()

    def _eager_step(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.module(*self.static_inputs), *self.static_targets)
        loss.backward()
        if self.distributed:
            self._hd.allreduce_gradients_flat(self.module.parameters())
        self.optimizer.step()
        return loss

    def __call__(self, inputs: Sequence[torch.Tensor], targets: Sequence[torch.Tensor]) -> torch.Tensor:
        for dst, src in zip(self.static_inputs + self.static_targets, list(inputs) + list(targets)):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()  # gradients land in the captured .grad buffers
        if self.distributed:
            self._hd.allreduce_gradients_flat(self.module.parameters())
            self.optimizer.step()
        return self.static_loss
