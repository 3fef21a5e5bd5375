"""CPU-only, world_size 2 over gloo: the data-parallel gradient reduction averages every flat bucket
(and the loose prototype tensor) across ranks, with and without the per-bucket overlap hooks, and the
epoch loop of train.py (sharded, length-bucketed batches) terminates with equal steps per rank on an odd batch count."""
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


class _TwinSystem(torch.nn.Module):
    """Two buckets of EQUAL size (as pool_a / pool_t are): pairing them wrongly in an all-reduce goes unnoticed by gloo."""

    def __init__(self):
        super().__init__()
        from ser_amd.models.pooling import AttentiveStatsPooling
        from ser_amd.models.prototypes import PrototypeMemory
        torch.manual_seed(0)
        self.pool_a = AttentiveStatsPooling(16, 8)
        self.pool_t = AttentiveStatsPooling(16, 8)
        self.prototypes = PrototypeMemory(4, 8)

    def buckets(self):
        return [self.pool_a._flat, self.pool_t._flat]


def _worker_mixed_paths(rank, world, port, out):
    """Rank 0 takes the eager path with hooks that fire in the REVERSE of the bucket order (autograd may run pool_t's
    backward before pool_a's), rank 1 the graph path (everything in finish(), plus an early start() of one bucket): the
    collectives must still pair bucket with bucket - the reducer issues them in one fixed sequence."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ser_amd.system import GradReducer
    s = _TwinSystem()
    for b in s.buckets():
        b.ensure()
    red = GradReducer(s, overlap=True)
    fill = lambda b, bi, step: b.gflat.copy_(torch.arange(b.total, dtype=torch.float32) * (rank + 1) + 100.0 * bi + step)
    for step in range(3):
        bs = s.buckets()
        if rank == 0:
            red.arm()
            for bi in (1, 0):                     # reverse order
                fill(bs[bi], bi, step)
                bs[bi].publish()
        else:
            for bi in (0, 1):
                fill(bs[bi], bi, step)
            if step == 1:
                red.start([bs[1]])               # not first in the sequence: must wait for bucket 0
            if step == 2:
                red.start([bs[0]])
        s.prototypes.prototypes.grad = torch.full((4, 8), float(rank + 1 + step))
        red.finish()
        assert all(b.grad_ready_hook is None for b in bs), "hooks must not outlive the step that armed them"
        mean_scale = sum(r + 1 for r in range(world)) / world
        for bi, b in enumerate(bs):
            want = torch.arange(b.total, dtype=torch.float32) * mean_scale + 100.0 * bi + step
            assert torch.allclose(b.gflat, want), f"rank {rank} step {step}: bucket {bi} was paired with another bucket"
        assert torch.allclose(s.prototypes.prototypes.grad, torch.full((4, 8), mean_scale + step))
    # a capture-time warm-up pass (quiet scope) must neither reduce nor leave anything marked
    with red.quiet():
        for b in s.buckets():
            b.publish()
        assert not red.pending
    out.put((rank, True))
    dist.destroy_process_group()


def test_ranks_on_different_paths_pair_the_same_buckets():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mixed_paths, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]


# ---- the training loop of train.py under data parallelism, odd number of batches --------------------------------------

class _ToyDataset(torch.utils.data.Dataset):
    """Ragged clips whose samples all equal the item index (so a step can tell which items it was given)."""

    def __init__(self, n):
        self.len = [800 + 160 * (i % 5) for i in range(n)]

    def __len__(self):
        return len(self.len)

    def __getitem__(self, i):
        return torch.full((self.len[i],), float(i)), f"t{i}", i % 4

    def lengths(self):
        return self.len


class _ToyEngine:
    """Stand-in for train.HipEngine with the same interface: every step averages a rank-dependent "gradient" through the
    product's GradReducer (gloo), counts scheduler steps and records the items it saw."""

    def __init__(self, rank, world, steps_per_epoch, epochs):
        from ser_amd.system import GradReducer
        self.rank, self.world, self.start_epoch = rank, world, 0
        self.sys = _FakeSystem()
        for b in self.sys.buckets():
            b.ensure()
        self.red = GradReducer(self.sys, overlap=False)
        self.total_steps, self.sched_steps, self.seen, self.batch_lens = steps_per_epoch * epochs, 0, [], []

    def begin_epoch(self, epoch):
        self.seen.append([])

    def train_step(self, audio_list, text_list, labels, epoch, step):
        ids = [int(w[0]) for w in audio_list]
        self.seen[-1] += ids
        self.batch_lens.append(sorted({int(w.numel()) for w in audio_list}))
        self.red.arm()
        for b in self.sys.buckets():
            b.gflat.fill_(float(sum(ids)))
            b.publish()
        self.sys.prototypes.prototypes.grad = torch.full((4, 8), float(self.rank))
        self.red.finish()                      # blocks forever if the other rank runs a different number of steps
        self.sched_steps += 1
        return torch.tensor(0.5)

    def end_epoch_health(self, loss):
        assert loss is not None

    def begin_eval(self):
        pass

    def predict(self, audio_list, text_list, want_features):
        return torch.tensor([int(w[0]) % 4 for w in audio_list]), (torch.zeros(len(audio_list), 2) if want_features else None)

    def fit_weibull(self, feats, gold):
        self.fitted = feats.shape[0]

    def checkpoint(self, epoch, f1):
        return dict(epoch=epoch, f1=f1)


def _loop_worker(rank, world, port, n_items, batch, out, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ser_amd import train as T
    args = T.build_parser().parse_args(["--epochs", "2", "--batch_size", str(batch), "--save_dir", tmp, "--seed", "3"])
    ds = _ToyDataset(n_items)
    sampler, tl, vl = T.make_loaders(args, ds, _ToyDataset(6), rank, world)
    eng = _ToyEngine(rank, world, len(sampler), args.epochs)
    f1 = T.run(args, eng, sampler, tl, vl, rank, world, log=lambda *_: None)
    out.put((rank, eng.sched_steps, eng.total_steps, eng.seen, eng.batch_lens, f1))
    dist.destroy_process_group()


def test_train_loop_finishes_on_an_odd_batch_count_with_equal_steps_per_rank(tmp_path):
    """35 items in batches of 4 = 9 global batches on 2 ranks (the case that used to strand rank 0 in an all-reduce):
    both ranks take 5 steps per epoch, together they see every item, the schedule length equals the steps one rank
    takes, and the loop returns."""
    n_items, batch, world = 35, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, n_items, batch, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    steps = [r[1] for r in res]
    assert steps == [10, 10] and all(r[2] == 10 for r in res)           # 5 steps x 2 epochs on every rank = schedule length
    for epoch in range(2):
        seen = sorted(i for r in res for i in r[3][epoch])
        assert set(seen) == set(range(n_items)), "an item was never visited"
        assert len(seen) == 10 * batch - 1 or len(seen) >= n_items       # tail padded by wrapping around
    assert res[0][3][0] != res[0][3][1], "the epochs must be shuffled differently"
    # length bucketing: most batches hold a single clip length
    single = sum(1 for r in res for ls in r[4] if len(ls) == 1)
    assert single >= 0.5 * sum(len(r[4]) for r in res)
    assert abs(res[0][5] - res[1][5]) < 1e-12


def test_sampler_shards_are_equal_disjoint_and_cover_the_corpus():
    from ser_amd.data.sampler import ShardedBucketBatchSampler
    world, n, bs = 8, 7442, 16          # BASELINE config 4: 7 442 utterances in batches of 16 x 8 -> 466 global batches
    lengths = [16000 * 3 for _ in range(n)]
    shards = []
    for r in range(world):
        s = ShardedBucketBatchSampler(lengths, bs, world, r, seed=1)
        s.set_epoch(2)
        shards.append(list(s))
        assert len(shards[-1]) == len(s) == 59                           # ceil(466 / 8): the same on every rank
    flat = [i for sh in shards for b in sh for i in b]
    assert set(flat) == set(range(n))
    assert len(flat) - n <= (59 * world - 466) * bs                      # only the wrapped-around tail repeats
    s0 = ShardedBucketBatchSampler(lengths, bs, world, 0, seed=1)
    s0.set_epoch(3)
    assert list(s0) != shards[0], "a new epoch draws a new permutation"
    s0.set_epoch(2)
    assert list(s0) == shards[0], "the permutation depends on (seed, epoch) only, not on global RNG state"
