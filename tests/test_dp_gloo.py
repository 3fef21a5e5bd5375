"""CPU-only, world_size 2 over gloo: the data-parallel gradient reduction averages every flat bucket
(and the loose prototype tensor) across ranks, with and without the per-bucket overlap hooks, and the
batch sharding of train.py gives each rank a disjoint, covering set of batches."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ser_amd  # noqa: F401


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeSystem(torch.nn.Module):
    """Same bucket interface as SERSystem, tiny modules, CPU tensors."""

    def __init__(self):
        super().__init__()
        from ser_amd.models import FusionLayer
        from ser_amd.models.pooling import AttentiveStatsPooling
        from ser_amd.models.prototypes import PrototypeMemory
        torch.manual_seed(0)
        self.fusion = FusionLayer(16, 16, 8)
        self.pool_a = AttentiveStatsPooling(16, 8)
        self.prototypes = PrototypeMemory(4, 8)

    def buckets(self):
        return [self.fusion._flat, self.pool_a._flat]


def _worker(rank, world, port, overlap, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ser_amd.system import GradReducer
    s = _FakeSystem()
    for b in s.buckets():
        b.ensure()
    red = GradReducer(s, overlap=overlap)
    for step in range(2):
        red.arm()
        for b in s.buckets():            # "backward": every rank writes rank-dependent gradients
            b.gflat.copy_(torch.arange(b.total, dtype=torch.float32) * (rank + 1) + step)
            b.publish()
        s.prototypes.prototypes.grad = torch.full((4, 8), float(rank + 1 + step))
        red.finish()
        mean_scale = sum(r + 1 for r in range(world)) / world
        for b in s.buckets():
            want = torch.arange(b.total, dtype=torch.float32) * mean_scale + step
            assert torch.allclose(b.gflat, want), f"rank {rank} step {step}: bucket not averaged"
            assert b.params[0].grad.data_ptr() == b.gflat.data_ptr()      # optimizer reads the reduced bucket
        assert torch.allclose(s.prototypes.prototypes.grad, torch.full((4, 8), mean_scale + step))
    out.put((rank, True))
    dist.destroy_process_group()


def _run(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]


def test_bucket_mean_with_overlap_hooks():
    _run(True)


def test_bucket_mean_without_hooks():
    _run(False)


def test_batch_sharding_is_disjoint_and_covering():
    world, nb = 8, 59          # BASELINE config 4: 7 442 utterances in batches of 16 x 8
    seen = [[bi for bi in range(nb) if bi % world == r] for r in range(world)]
    flat = sorted(b for s in seen for b in s)
    assert flat == list(range(nb))
    assert max(len(s) for s in seen) - min(len(s) for s in seen) <= 1
