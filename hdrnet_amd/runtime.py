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

from . import _state


class GraphedInference:
    """Capture ``module(*example_inputs)`` once; ``__call__`` copies new inputs into the static
    buffers, replays the graph and returns the static output (valid until the next call).

    Shapes are fixed at capture time; a call with different shapes raises ``ValueError``.

    The PARAMETERS are those of capture time: the models keep the derived arrays the kernels read (batch norm folded
    into the guide and the coefficient network, ``models._Coefficients.exported``) in a cache that the warm-up fills,
    so the graph holds pointers to them rather than the ~40 launches that derive them.  After changing the module's
    parameters (an optimizer step, ``load_state_dict``) call ``recapture()``: a replay with parameters other than the
    captured ones RAISES (``check_parameters``; the check reads every parameter's and buffer's identity and version
    counter plus the package's parameter-state generation, ~10 us of host time per call) instead of serving the stale
    weights.
    """

    def __init__(self, module: torch.nn.Module, example_inputs: Sequence[torch.Tensor], warmup: int = 3,
                 check_parameters: bool = True):
        if not all(t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedInference needs device tensors (MI355X)")
        self.module = module.eval()
        self.static_inputs = [t.clone() for t in example_inputs]
        self.warmup = warmup
        self.check_parameters = bool(check_parameters)
        self.recapture()

    def _parameter_state(self) -> tuple:
        m = self.module
        return tuple((t.data_ptr(), t._version, _state.generation_of(t)) for t in list(m.parameters()) + list(m.buffers()))

    def recapture(self) -> None:
        """(Re-)capture the graph with the module's current parameters; the static input buffers are kept."""
        side = torch.cuda.Stream(device=self.static_inputs[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(self.warmup):  # library loading, allocator warm-up, kernel selection, derived-array caches
                self.module(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_output = self.module(*self.static_inputs)
        self._captured_state = self._parameter_state()

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        if len(inputs) != len(self.static_inputs):
            raise ValueError(f"expected {len(self.static_inputs)} inputs, got {len(inputs)}")
        if self.check_parameters and self._parameter_state() != self._captured_state:
            raise RuntimeError("GraphedInference: the module's parameters or buffers changed since the graph was captured "
                               "(optimizer step, load_state_dict, .to()); the graph holds the OLD folded weights -- call "
                               "recapture()")
        for dst, src in zip(self.static_inputs, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_output


class FramePipeline:
    """Independent frames round-robin over ``depth`` HIP streams.

    One launch of the 4K slice-apply spends ~5 us filling the memory pipeline and ~6 us draining it; on ONE stream
    consecutive launches are serialised, so every frame pays both.  The frames of a stream of images are
    independent: with two streams the next frame's fill runs under the previous frame's drain -- 39.9 -> 36.9 us per
    4K frame, 11.6 -> 9.5 (two streams) / 8.6 us (three) per 1080p frame for the bare op
    (``tools/two_stream_probe.py``, profiles/r03/two_streams.txt).  ``make_lane()`` builds one lane's callable (for
    a whole model: a ``GraphedInference`` with its own static buffers); ``submit(*inputs)`` runs the next lane on
    its stream and returns a ticket; ``result(ticket)`` makes the CALLER's current stream wait for that frame and
    returns its output (valid until the lane is submitted to again, ``depth`` frames later).
    """

    def __init__(self, make_lane, depth: int = 2, device=None):
        if depth < 1:
            raise ValueError("depth >= 1")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.lanes = []
        for st in self.streams:  # build (and, for graphs, capture) every lane under its own stream
            st.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(st):
                self.lanes.append(make_lane())
        with torch.cuda.device(dev):  # events belong to the device of the streams, whichever device is current
            self.done = [torch.cuda.Event() for _ in range(depth)]
        self.outputs = [None] * depth
        self.next = 0

    def submit(self, *inputs: torch.Tensor) -> int:
        lane = self.next
        self.next = (lane + 1) % len(self.lanes)
        st = self.streams[lane]
        st.wait_stream(torch.cuda.current_stream(st.device))  # the inputs were produced on the caller's stream
        for t in inputs:
            # ... and are READ on the lane's stream: tell the caching allocator, or a caller that drops (or lets go
            # of) an input right after submit() -- the normal streaming pattern -- gets its block handed out again
            # on the caller's stream while the lane is still reading it
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(st)
        with torch.cuda.stream(st):
            self.outputs[lane] = self.lanes[lane](*inputs)
            self.done[lane].record(st)
        return lane

    def result(self, ticket: int):
        cur = torch.cuda.current_stream(self.streams[ticket].device)
        cur.wait_event(self.done[ticket])
        out = self.outputs[ticket]
        for t in (out if isinstance(out, (tuple, list)) else (out,)):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)  # allocated on the lane's stream, read on the caller's: keep the allocator honest
        return out


class TrainStep:
    """One training step -- forward, loss, backward, gradient all-reduce across the ranks, optimizer update -- with
    the gradients kept in ONE persistent flat bucket (``dist.GradBucket``: ``.grad`` of every parameter is a view),
    so that the step's only collective is one in-place all-reduce (RCCL over xGMI on the node; SURVEY.md section
    8e) with no per-step flatten / scatter.  Eager; ``GraphedTrainStep`` replays forward + backward from a hipGraph
    and shares everything after the backward with this class.  The reference's step is the single-device
    ``sess.run(train_op)`` of ``hdrnet/bin/train.py:113-157``.

    ``loss_fn(output, *targets) -> scalar``.
    """

    def __init__(self, module: torch.nn.Module, loss_fn, optimizer: torch.optim.Optimizer):
        from . import dist as hd
        self.module, self.loss_fn, self.optimizer = module, loss_fn, optimizer
        self._hd = hd
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1
        # True: issue the flat-bucket all-reduce even at world size 1 (a one-rank communicator; dist.init(single=True))
        self.force_collective = False
        # an optimizer that keeps its own flat gradient bucket (optim.FlatAdam) shares it with the step
        self.bucket = getattr(optimizer, "bucket", None) or hd.GradBucket(module.parameters())

    def _forward_backward(self, inputs, targets) -> torch.Tensor:
        self.bucket.release()  # .grad = None: backward assigns its gradients (no memset, no `+=` launch per parameter)
        loss = self.loss_fn(self.module(*inputs), *targets)
        one = getattr(self, "_one", None)   # the root gradient, kept: `loss.backward()` fills a fresh ones_like per step
        if one is None or one.shape != loss.shape or one.dtype != loss.dtype or one.device != loss.device:
            one = self._one = torch.ones_like(loss)
        loss.backward(one)
        self.bucket.gather()   # one multi-tensor copy into the flat buffer; every .grad is its view again
        return loss

    def _after_backward(self) -> None:
        if self.distributed or self.force_collective:
            self.bucket.allreduce(force=self.force_collective)
        self.optimizer.step()

    def __call__(self, inputs: Sequence[torch.Tensor], targets: Sequence[torch.Tensor]) -> torch.Tensor:
        loss = self._forward_backward(list(inputs), list(targets))
        self._after_backward()
        return loss


class GraphedTrainStep(TrainStep):
    """``TrainStep`` with forward, loss and backward through the HIP VJPs -- and, at world size 1, the optimizer
    update too -- captured into a hipGraph and replayed per batch.

    With the guide network fused, a config #4 step (4 x 1080p per GPU) is ~1.5 ms of GPU work behind ~3.7 ms of
    host-side launches of the coefficient network's small ops; the graph removes the host from the loop.  The
    reference's equivalent is the TF session running its static training graph (``hdrnet/bin/train.py:156-226``).

    ``optimizer`` must be capturable (e.g. ``torch.optim.Adam(..., capturable=True)``).  In a multi-process job
    (``torch.distributed`` initialised, world size > 1) the graph ends with the backward: ``__call__`` then runs
    the ONE in-place all-reduce of the flat gradient bucket and the optimizer step eagerly -- the collective stays
    outside the graph.  ``flat_bucket=True`` forces that structure at world size 1 as well (bench / tests).
    """

    def __init__(self, module: torch.nn.Module, loss_fn, optimizer: torch.optim.Optimizer,
                 example_inputs: Sequence[torch.Tensor], example_targets: Sequence[torch.Tensor],
                 warmup: int = 3, flat_bucket: bool = False, feeds: int = 1):
        """``feeds``: the number of input-buffer sets, each with a captured graph of its own.  With 2, ``prefetch()``
        stages batch k + 1 on a copy stream while the replay of batch k runs (the staging copies of a 4 x 1080p batch are
        200 MB each way, 75-90 us of a 0.58-ms step when they run in front of the graph on its stream)."""
        if not all(t.is_cuda for t in list(example_inputs) + list(example_targets)):
            raise RuntimeError("GraphedTrainStep needs device tensors (MI355X)")
        if feeds < 1:
            raise ValueError("feeds >= 1")
        super().__init__(module, loss_fn, optimizer)
        # this step's generation cell on everything a replay may write behind the version counters: the parameters
        # (a capturable optimizer inside the graph) and the buffers (batch norm's running statistics)
        self._writer = _state.Writer(list(module.parameters()) + list(module.buffers()))
        self.split = self.distributed or flat_bucket  # graph = forward + backward only; the rest eager
        self._inputs = [[t.clone() for t in example_inputs] for _ in range(feeds)]
        self._targets = [[t.clone() for t in example_targets] for _ in range(feeds)]
        self.static_inputs, self.static_targets = self._inputs[0], self._targets[0]
        dev = self.static_inputs[0].device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # also materialises the optimizer state before capture
                self._forward_backward(self.static_inputs, self.static_targets)
                self._after_backward()
        torch.cuda.current_stream().wait_stream(side)
        self.graphs, self._losses = [], []
        for ins, tgts in zip(self._inputs, self._targets):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss = self._forward_backward(ins, tgts)
                if not self.split:
                    self.optimizer.step()
            self.graphs.append(g)
            self._losses.append(loss)
        self.graph, self.static_loss = self.graphs[0], self._losses[0]
        if not self.bucket.attached():
            raise RuntimeError("a parameter's .grad was re-bound during capture: the flat bucket is detached")
        self._copy_stream = torch.cuda.Stream(device=dev) if feeds > 1 else None
        with torch.cuda.device(dev):  # events belong to the device current at their creation
            self._ready = [torch.cuda.Event() for _ in range(feeds)]
        self._staged: list = []   # slots filled by prefetch(), oldest first
        self._next_slot = 0

    @staticmethod
    def _check(dst: torch.Tensor, src: torch.Tensor) -> None:
        if dst.shape != src.shape or dst.dtype != src.dtype:
            raise ValueError(f"captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")

    def prefetch(self, inputs: Sequence[torch.Tensor], targets: Sequence[torch.Tensor]) -> int:
        """Stage a batch into the next buffer set on the copy stream and return at once; the next ``step()`` without
        arguments consumes the oldest staged batch.  The copy waits for the work already queued on the current stream
        (the last replay that read this buffer set is among it) and runs beside whatever is queued afterwards -- call it
        BEFORE the ``step()`` it should overlap with.  Needs ``feeds >= 2``."""
        if self._copy_stream is None:
            raise RuntimeError("prefetch() needs GraphedTrainStep(..., feeds=2)")
        if len(self._staged) >= len(self.graphs):
            raise RuntimeError("every buffer set holds a staged batch: call step() first")
        slot = self._next_slot
        self._next_slot = (slot + 1) % len(self.graphs)
        cur = torch.cuda.current_stream(self._copy_stream.device)
        self._copy_stream.wait_stream(cur)
        with torch.cuda.stream(self._copy_stream):
            for dst, src in zip(self._inputs[slot] + self._targets[slot], list(inputs) + list(targets)):
                self._check(dst, src)
                dst.copy_(src, non_blocking=True)
                src.record_stream(self._copy_stream)
            self._ready[slot].record(self._copy_stream)
        self._staged.append(slot)
        return slot

    def step(self) -> torch.Tensor:
        """Replay the graph of the oldest staged batch (``prefetch``); returns that graph's loss tensor."""
        if not self._staged:
            raise RuntimeError("no staged batch: call prefetch() first")
        slot = self._staged.pop(0)
        torch.cuda.current_stream(self._copy_stream.device).wait_event(self._ready[slot])
        self.graphs[slot].replay()  # gradients land in the flat bucket
        self._writer.bump()  # the replay wrote parameters / batch-norm statistics behind the version counters' back
        if self.split:
            self._after_backward()
        return self._losses[slot]

    def __call__(self, inputs: Sequence[torch.Tensor], targets: Sequence[torch.Tensor]) -> torch.Tensor:
        if self._staged:
            raise RuntimeError("a prefetched batch is pending: consume it with step()")
        for dst, src in zip(self.static_inputs + self.static_targets, list(inputs) + list(targets)):
            self._check(dst, src)
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()  # gradients land in the flat bucket
        self._writer.bump()  # the replay wrote parameters / batch-norm statistics behind the version counters' back
        if self.split:
            self._after_backward()
        return self.static_loss
