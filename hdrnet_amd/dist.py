"""Multi-GPU plumbing for the bilateral-grid path: one process per GPU, images sharded
across ranks, NO data-path collective.

The path shards by image: every output pixel depends only on its own
guide / input pixel and on its own image's grid, so rank r simply owns a contiguous block of
the batch.  ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the MI355X node, "gloo"
in the CPU tests) is used for the control plane only: the barrier around a timed region, the
max-over-ranks of the elapsed time, and -- in tests -- gathering results to check them.
The reference itself has no multi-device code at all (hdrnet/models.py:194 pins '/gpu:0').
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: Optional[str] = None, device: Optional[torch.device] = None, single: bool = False) -> Tuple[int, int]:
    """Join the process group described by the environment (no-op for world size 1).

    ``single=True`` creates the group at world size 1 as well -- a one-rank RCCL communicator on one GPU, so that the
    collective code path (communicator set-up, the flat-bucket all-reduce kernel) runs on a single-GPU box exactly
    as it will on the node (tests/test_gpu_rccl.py, ``bench.py --workload train_1080p_b4 --force-collective``)."""
    rank, world, _ = env_rank_world()
    if world == 1 and not single:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, stop) of `n_items` images for `rank` (first ranks get the
    remainder).  Config #5: 8 images / 8 GPUs -> one each; config #4: 32 / 8 -> four each."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def row_range(height: int, rank: int, world: int) -> Tuple[int, int]:
    """Row band [y0, y1) of a `height`-row frame for `rank` when ONE frame is split over `world` GPUs
    (the optional intra-image split; hdrnet_ops.bilateral_slice_apply_rows).  The
    bands are contiguous, balanced (first ranks get the remainder) and cover the frame exactly; the
    grid (96-384 KiB) is replicated, nothing else is exchanged."""
    return shard_range(height, rank, world)


def gather_rows(local: torch.Tensor, height: int) -> torch.Tensor:
    """All-gather row bands [B, rows_r, W, C] back into whole frames [B, height, W, C] (tests /
    validation only; a pipeline that consumes bands never needs it)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    parts: List[Optional[torch.Tensor]] = [None] * world
    dist.all_gather_object(parts, local.cpu())
    out = torch.cat([p for p in parts if p is not None and p.shape[1] > 0], dim=1)
    assert out.shape[1] == height
    return out


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(values: List[float], device: Optional[torch.device] = None) -> List[float]:
    """Element-wise max of a few floats over all ranks (timings)."""
    if not (dist.is_available() and dist.is_initialized()):
        return list(values)
    t = torch.tensor(values, dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def sharded_apply(fn: Callable[..., torch.Tensor], grid: torch.Tensor, guide: torch.Tensor,
                  inp: torch.Tensor, rank: int, world: int, **kw) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """Run `fn(grid, guide, input, **kw)` on this rank's image shard of a full batch."""
    lo, hi = shard_range(guide.shape[0], rank, world)
    return fn(grid[lo:hi], guide[lo:hi], inp[lo:hi], **kw), (lo, hi)


def gather_batch(local: torch.Tensor, n_items: int) -> torch.Tensor:
    """All-gather variable-sized batch shards back into the full batch (tests / validation only;
    the product path never needs it)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    parts: List[Optional[torch.Tensor]] = [None] * world
    dist.all_gather_object(parts, local.cpu())
    out = torch.cat([p for p in parts if p is not None and p.shape[0] > 0], dim=0)
    assert out.shape[0] == n_items
    return out


class GradBucket:
    """The training step's gradients as ONE persistent flat fp32 buffer: every parameter's ``.grad`` is a VIEW into
    it, so the step's only collective (~482 k parameters, 1.9 MB) is a single in-place
    all-reduce on the buffer -- no per-step ``cat`` and no per-parameter copy back (what
    ``allreduce_gradients_flat`` does, ~35 small launches per step).

    Two ways to fill it.  (a) Leave the views bound: autograd accumulates in place into an existing ``.grad``, so the
    views survive backward -- call ``zero_()`` (one memset) instead of ``optimizer.zero_grad(set_to_none=True)``, which
    would drop them; costs one ``+=`` launch per parameter inside backward (35 of a graph-captured step's 210 launches,
    profiles/r04/train_step.md).  (b) ``release()`` before backward (every ``.grad`` = None: autograd then ASSIGNS its
    gradients, no zeroing and no adds) and ``gather()`` after it: one multi-tensor copy into the flat buffer, the views
    re-bound.  ``runtime.TrainStep`` does (b).
    """

    def __init__(self, params, align: int = 1):
        """``align``: every parameter's segment starts at a multiple of ``align`` elements (``optim.FlatAdam`` lays the
        parameters themselves out the same way and needs 16-byte aligned segments: 4)."""
        self.params = [p for p in params if p.requires_grad]
        self.align = max(int(align), 1)
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("GradBucket: parameters must share one device and dtype")
        pad = lambda n: -(-n // self.align) * self.align  # noqa: E731
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += pad(p.numel())
        self.flat = torch.zeros(off, device=dev, dtype=dt)
        self.no_grad_last: list = []  # parameters the last gather() found without a gradient (none before the first)
        self.views = []
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            seg = self.flat[off:off + n]
            # the view takes the PARAMETER's strides where it is a dense permutation (channels_last conv weights):
            # autograd's layout contract, and what the fused optimizers insist on
            dense = torch.empty_like(p).stride() == p.stride()  # preserve_format keeps a dense tensor's strides
            self.views.append(seg.as_strided(p.shape, p.stride()) if dense else seg.view_as(p))
            p.grad = self.views[-1]
            # a backward that computes this parameter's gradient with its own kernels may write it HERE when .grad is
            # None (released): autograd then adopts the alias and gather() has nothing to copy (hdrnet_ops._grad_out)
            p._hdrnet_grad_view = self.views[-1]
            p._hdrnet_grad_claimed = True  # handed out at most once per release(): a second use of the parameter adds

    def release(self) -> None:
        """Unbind the views (``.grad = None``) so that the next backward assigns its gradients instead of adding them
        to a zeroed buffer; ``gather()`` brings them into the flat buffer."""
        for p in self.params:
            p.grad = None
            p._hdrnet_grad_claimed = False

    def gather(self) -> None:
        """After a backward that ran on released gradients: copy them into the flat buffer (one multi-tensor launch),
        zero the segments of parameters that received none, re-bind every ``.grad`` to its view.

        A parameter that received NO gradient in this backward (an unused branch) thus ends up with a zero gradient,
        not ``None``: an Adam-type optimizer then still decays its moments and moves it by what is left of its momentum,
        where ``torch.optim`` would skip a ``None`` gradient.  Every parameter of the three model graphs receives a
        gradient in every step; ``self.no_grad_last`` lists the parameters that did not, for a caller that freezes
        branches and wants to mask them."""
        self.no_grad_last = [p for p in self.params if p.grad is None]
        dst, src = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
            p.grad = v
        if dst:
            fused = getattr(torch, "_foreach_copy_", None)
            if fused is not None:
                fused(dst, src)  # one multi-tensor launch
            else:
                for d, g in zip(dst, src):
                    d.copy_(g)

    def attached(self) -> bool:
        """True while every parameter's .grad still is its view of the flat buffer."""
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                return False
        return True

    def zero_(self) -> None:
        self.flat.zero_()

    def allreduce(self, world: Optional[int] = None, force: bool = False) -> int:
        """Average over the ranks, in place, ONE collective (RCCL over xGMI on the GPU node; gloo in the CPU
        tests).  No-op without a process group, and at world size 1 unless ``force`` (then the one-rank collective
        is issued all the same: the communicator and its kernel run, the values do not change).  Returns the
        element count."""
        if dist.is_available() and dist.is_initialized():
            w = float(world or dist.get_world_size())
            if w > 1 or force:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                if w > 1:
                    self.flat.mul_(1.0 / w)
        return int(self.flat.numel())


def allreduce_gradients_flat(params, world: Optional[int] = None) -> int:
    """The training step's only collective: every gradient flattened into
    ONE contiguous fp32 bucket, one all-reduce (RCCL over xGMI on the GPU node), averaged, and
    scattered back.  The network is ~482 k parameters (1.9 MB): a single latency-bound ring
    all-reduce per step; nothing to overlap with.  Returns the bucket size in elements."""
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return 0
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= float(world or dist.get_world_size())
    off = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
    return int(flat.numel())
