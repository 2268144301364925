"""N > 1 path on CPU: world_size-2 gloo processes, images sharded per rank, no data-path
collective; the control plane (barrier, max-over-ranks timing, gather-for-checking) is what
bench.py uses with RCCL on the GPU node.  The per-rank compute here is the CPU oracle (this is
a test of the sharding logic; the GPU kernels are covered by test_gpu_parity.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import oracle
    from hdrnet_amd import dist as hd
    r, w = hd.init(backend="gloo")
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(1234)  # same full batch on every rank
    grid = torch.from_numpy(rng.random((B, 4, 5, 3, 12), dtype=np.float32))
    guide = torch.from_numpy(rng.random((B, 12, 20), dtype=np.float32))
    inp = torch.from_numpy(rng.random((B, 12, 20, 3), dtype=np.float32))
    P = oracle.port()
    P.set_threads(1)

    def fn(g, gu, i, has_offset):
        return torch.from_numpy(P.bilateral_slice_apply(g.numpy(), gu.numpy(), i.numpy(), has_offset))

    hd.barrier()
    local, (lo, hi) = hd.sharded_apply(fn, grid, guide, inp, rank, world, has_offset=True)
    hd.barrier()
    t = hd.max_over_ranks([0.5 + rank, 7.0 - rank])
    full = hd.gather_batch(local, B)
    want = fn(grid, guide, inp, True)
    ok = bool(torch.equal(full, want)) and t == [0.5 + world - 1, 7.0] and local.shape[0] == hi - lo
    if rank == 0:
        q.put((ok, (lo, hi), t))
    hd.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("B", [8, 5, 1])
def test_two_rank_image_sharding(B):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    ok, rng0, t = q.get()
    assert ok, (rng0, t)


def test_shard_range_properties():
    from hdrnet_amd.dist import shard_range
    for n in (0, 1, 7, 8, 32, 33):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(8, 3, 8) == (3, 4)      # config #5: one image per GPU
    assert shard_range(32, 7, 8) == (28, 32)   # config #4: four per GPU
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


# ---- one frame over N ranks: the row split (optional) ---------------------------
def _rows_worker(rank, world, port, H, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import oracle
    from hdrnet_amd import dist as hd
    hd.init(backend="gloo")
    rng = np.random.default_rng(77)  # the same frames on every rank
    B, W = 2, 24
    grid = rng.random((B, 4, 5, 3, 12), dtype=np.float32)  # replicated (a few hundred KiB at most)
    guide = rng.random((B, H, W), dtype=np.float32)
    inp = rng.random((B, H, W, 3), dtype=np.float32)
    P = oracle.port()
    P.set_threads(1)
    y0, y1 = hd.row_range(H, rank, world)
    band = P.bilateral_slice_apply_rows(grid, guide[:, y0:y1], inp[:, y0:y1], H, y0, True)
    full = hd.gather_rows(torch.from_numpy(band), H)
    want = torch.from_numpy(P.bilateral_slice_apply(grid, guide, inp, True))
    ok = bool(torch.equal(full, want)) and band.shape[1] == y1 - y0  # bit for bit: gyf is the frame's
    if rank == 0:
        q.put((ok, (y0, y1)))
    hd.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("H", [13, 12, 1])
def test_two_rank_row_split_equals_whole_frame(H):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, 2, port, H, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    ok, span = q.get()
    assert ok, span


def test_row_range_covers_the_frame():
    from hdrnet_amd.dist import row_range
    for H in (0, 1, 7, 1080, 2160, 3000):
        for world in (1, 2, 4, 8):
            spans = [row_range(H, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == H
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert row_range(2160, 7, 8) == (1890, 2160)  # 4K over 8 GPUs: 270 rows each
