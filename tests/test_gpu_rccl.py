"""RCCL at world size 1 on ONE GPU: `backend="nccl"` IS RCCL on ROCm, and until this test the
nccl branch of hdrnet_amd/dist.py had never been initialised by this code -- every earlier multi-rank run used gloo.
A one-rank RCCL communicator exercises communicator set-up and the all-reduce kernel on the flat gradient bucket
exactly where the 8-GPU job issues them: after a GraphedTrainStep(flat_bucket=True) replay (graph = forward + loss +
backward; the collective and the optimizer update eager).  No scaling number is expected from this."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    import torch.distributed as dist
    from hdrnet_amd import dist as hd
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    assert not dist.is_initialized()
    saved = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for k in ("RANK", "WORLD_SIZE", "MASTER_PORT"):
        os.environ.pop(k, None)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rank, world = hd.init(backend="nccl", device=dev, single=True)
    assert (rank, world) == (0, 1) and dist.get_backend() == "nccl"
    yield dev
    dist.destroy_process_group()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_rccl_allreduce_of_the_flat_bucket(nccl_group):
    """The 482 k-element bucket of HDRNetPointwiseNNGuide through dist.GradBucket.allreduce(force=True): a one-rank
    sum leaves the values as they are, and librccl is what ran it."""
    from hdrnet_amd import dist as hd, models
    dev = nccl_group
    model = models.HDRNetPointwiseNNGuide(dict(batch_norm=False)).to(dev)
    bucket = hd.GradBucket(model.parameters(), align=4)
    assert bucket.flat.numel() > 400_000
    gen = torch.Generator(device=dev).manual_seed(5)
    bucket.flat.copy_(torch.randn(bucket.flat.shape, device=dev, generator=gen))
    before = bucket.flat.clone()
    n = bucket.allreduce(force=True)
    torch.cuda.synchronize()
    assert n == bucket.flat.numel()
    assert torch.equal(bucket.flat, before)
    assert bucket.attached()
    maps = open("/proc/self/maps").read()
    assert "librccl" in maps, "the nccl backend of torch on ROCm is RCCL: librccl.so must be mapped after a collective"
    hd.barrier()
    assert hd.max_over_ranks([1.5, 2.5], device=dev) == [1.5, 2.5]


def test_graphed_train_step_with_the_collective_forced(nccl_group):
    """GraphedTrainStep(flat_bucket=True): the graph ends with the backward, then the all-reduce (forced here at
    world size 1, the real RCCL kernel) and FlatAdam run eagerly -- the structure every rank of the 8-GPU job runs.
    The step must equal the same step without the collective, bit for bit (a one-rank sum is the identity)."""
    from hdrnet_amd import metrics, models, optim
    from hdrnet_amd.runtime import GraphedTrainStep
    dev = nccl_group
    B, H, W = 2, 96, 128
    gen = torch.Generator(device=dev).manual_seed(11)
    low = torch.rand((B, 256, 256, 3), device=dev, generator=gen)
    full = torch.rand((B, H, W, 3), device=dev, generator=gen)
    target = torch.rand((B, H, W, 3), device=dev, generator=gen)
    results = []
    for forced in (False, True):
        torch.manual_seed(0)
        model = models.HDRNetPointwiseNNGuide(dict(batch_norm=False)).to(dev).train()
        opt = optim.FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, epsilon_hat=True)
        step = GraphedTrainStep(model, lambda out, tgt: metrics.l2_loss(tgt, out), opt, [low, full], [target],
                                flat_bucket=True)
        assert step.split
        step.force_collective = forced
        losses = [float(step([low, full], [target])) for _ in range(3)]
        torch.cuda.synchronize()
        assert step.bucket.attached()
        results.append((losses, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()))
    assert results[0][0] == results[1][0]
    assert torch.equal(results[0][1], results[1][1])
    assert results[0][0][2] < results[0][0][0]  # it trains
