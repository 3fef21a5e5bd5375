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


def _stage_sums(sysm, batch):
    """Checksums (float64 sums) of the stages of one eager, dropout-free forward: encoder outputs, fused vector, logits, loss."""
    with torch.no_grad():
        wave, ids, mask, labels = batch
        a_enc, t_enc = sysm.encode_frozen(wave, ids, mask)
        loss, logits = sysm.loss_from_encoded(a_enc, t_enc, mask, labels)
        torch.cuda.synchronize()
        return dict(a_enc=a_enc.double().sum().item(), t_enc=t_enc.double().sum().item(), logits=logits.double().sum().item(),
                    loss=loss.item())


def _diagnose(sds0, wc, xc, batch, seq_loss, pipe_loss, sums_a, sums_b, sys_a, sys_b):
    """Which path is wrong?  The CPU oracle's loss for the same batch on the INITIAL weights decides; the report carries
    what is needed to localise the stage without running anything again."""
    import __graft_entry__ as ge
    from oracle import ser_oracle as O
    from ser_amd import _engines as E
    wave, ids, mask, labels = [t.cpu() for t in batch]
    a_cfg, t_cfg = ge.oracle_cfgs(wc, xc)
    ref = O.full_forward(sds0, list(wave), ids, mask, a_cfg, t_cfg, num_layers=3, heads=2, use_openmax=False, training=True)
    want = O.train_loss(ref["logits"], ref["unc"], ref["fused"], sds0["prototypes"]["prototypes"], labels, 4).item()
    dev_seq, dev_pipe = abs(seq_loss - want), abs(pipe_loss - want)
    culprit = "eager sequential path (sys_a)" if dev_seq > dev_pipe else "graph-pipelined path (sys_b)"
    aborts = [[int(sc[1]) for sc in getattr(m.classifier, "_stack_cache", (None,) * 4)[3] or []] for m in (sys_a, sys_b)]
    return (f"first losses differ: sequential {seq_loss!r} vs pipelined {pipe_loss!r}; CPU oracle {want!r} -> the {culprit} deviates "
            f"(|seq - oracle| = {dev_seq:.3e}, |pipe - oracle| = {dev_pipe:.3e}).  Stage checksums of a fresh eager forward before "
            f"any step - sys_a {sums_a} / sys_b {sums_b}.  GEMM plans {dict((k, v[0][1]) for k, v in E._TUNE_RANKED.items())}; "
            f"persistent-stack abort words {aborts}")


def _worker_pipelined(rank, world, port, q):
    """The timed schedule under data parallelism: PipelinedStepper with grouped encoder passes and the head graph
    captured in two pieces with the classifier bucket's all-reduce between them (what `reducer.early` selects under
    RCCL; forced here over gloo)."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SER_DP_EARLY="1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import __graft_entry__ as ge
        from ser_amd.system import GradReducer, PipelinedStepper, TrainStepper
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        sys_a, wc, xc = ge._small_system(dev)
        sys_b, _, _ = ge._small_system(dev)
        sys_b.load_state_dict(sys_a.state_dict())
        sys_a.train(); sys_b.train()
        sds0 = {k: {n: v.detach().cpu().clone() for n, v in getattr(sys_a, k).state_dict().items()} for k in sys_a.CKPT_KEYS}
        n, G = 9, 2
        batches = [[t.to(dev) for t in _batch(300 + 10 * i + rank)] for i in range(n)]
        sums_a, sums_b = _stage_sums(sys_a, batches[0]), _stage_sums(sys_b, batches[0])
        assert sums_a == sums_b, f"two replicas with the same weights disagree on one eager forward: {sums_a} vs {sums_b}"
        # reference: one batch at a time, eager, all buckets reduced after backward
        oa = sys_a.make_optimizer(lr=1e-3)
        ra = GradReducer(sys_a, overlap=False)
        seq = TrainStepper(sys_a, oa, None, ra, use_graph=False)
        seq_losses = [seq.step(*b).item() for b in batches]
        # the timed schedule
        ob = sys_b.make_optimizer(lr=1e-3)
        rb = GradReducer(sys_b)
        assert rb.early
        pipe = PipelinedStepper(sys_b, ob, None, rb, group=G)
        assert pipe.split
        for j in range(pipe.prime):
            pipe.feed(*batches[j])
        pipe_losses = [pipe.step(*batches[i]).item() for i in range(pipe.prime, n)]
        pipe_losses += [l.item() for l in pipe.drain()]
        torch.cuda.synchronize()
        sys_a.check_persistent_kernels(); sys_b.check_persistent_kernels()
        # same arithmetic up to the summation order of the reduction and of the two consumers of `fused`: the first loss
        # (before any update) is identical, later ones drift by rounding amplified through AdamW (lr 1e-3); a bucket that
        # was not averaged, or averaged twice, would move every parameter by O(lr) per step instead
        if seq_losses[0] != pipe_losses[0]:
            raise AssertionError(_diagnose(sds0, wc, xc, batches[0], seq_losses[0], pipe_losses[0], sums_a, sums_b, sys_a, sys_b))
        assert seq_losses == pytest.approx(pipe_losses, abs=1e-3), (seq_losses, pipe_losses)
        fa = torch.cat([p.detach().reshape(-1) for p in sys_a.parameters() if p.requires_grad])
        fb = torch.cat([p.detach().reshape(-1) for p in sys_b.parameters() if p.requires_grad])
        assert (fa - fb).abs().mean().item() < 5e-5, f"pipelined DP schedule drifted from sequential DP stepping: mean |diff| {(fa - fb).abs().mean().item()}"
        parts = [torch.empty_like(fb) for _ in range(world)]
        dist.all_gather(parts, fb)
        assert torch.equal(parts[0], parts[1]), "replicas diverged under the pipelined schedule"
        q.put((rank, "ok"))
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
        raise


def test_two_ranks_pipelined_split_backward_schedule():
    """Two ranks (two processes sharing the one GPU, gloo) step through the timed schedule and through eager sequential
    stepping from the same weights.  The first loss - forward only - must be identical: round 2 saw it differ rarely and
    allowed a retry; the cause was the lazy flattening of the text-side parameter buckets on a side stream
    (models/_flat.py, tests/test_gpu_system.py::test_first_forward_on_busy_streams_keeps_every_parameter), fixed in round 3.
    No retry: a mismatch reports which path deviates from the CPU oracle."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(400)
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert [r[1] for r in res] == ["ok", "ok"], res
