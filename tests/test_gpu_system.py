"""End-to-end GPU parity of one training step of the assembled hot path (ref train.py:145-177)
against the CPU oracle with torch autograd: logits, loss, gradients, one AdamW update; and
graph-captured stepping == eager stepping."""
import copy

import numpy as np
import pytest
import torch

from oracle import ser_oracle as O

pytestmark = pytest.mark.gpu


def _batch(seed, B=4, T=4000, S=9, vocab=1000, C=4):
    g = torch.Generator().manual_seed(seed)
    wave = 0.1 * torch.randn(B, T, generator=g)
    ids = torch.randint(4, vocab, (B, S), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    mask = torch.ones(B, S)
    ids[1, S - 3:] = 1
    ids[1, S - 4] = 2
    mask[1, S - 3:] = 0
    labels = torch.randint(0, C, (B,), generator=g)
    return wave, ids, mask, labels


def _oracle_step(sds, wave, ids, mask, labels, a_cfg, t_cfg):
    """Oracle forward + autograd backward over the trainable head (encoders frozen)."""
    leaf = {k: {n: v.clone().requires_grad_(v.dtype.is_floating_point and not n.startswith("encoder."))
                for n, v in sd.items()} for k, sd in sds.items()}
    out = O.full_forward(leaf, list(wave), ids, mask, a_cfg, t_cfg, num_layers=3, heads=2, use_openmax=False, training=True)
    loss = O.train_loss(out["logits"], out["unc"], out["fused"], leaf["prototypes"]["prototypes"], labels, 4)
    loss.backward()
    return out, loss, leaf


def test_train_step_matches_oracle():
    import __graft_entry__ as ge
    dev = torch.device("cuda:0")
    sysm, wc, xc = ge._small_system(dev)
    sysm.train()
    a_cfg, t_cfg = ge.oracle_cfgs(wc, xc)
    wave, ids, mask, labels = _batch(7)
    sds = {k: {n: v.detach().cpu().clone() for n, v in getattr(sysm, k).state_dict().items()} for k in sysm.CKPT_KEYS}
    out, ref_loss, leaf = _oracle_step(sds, wave, ids, mask, labels, a_cfg, t_cfg)

    loss, logits = sysm.loss(wave.to(dev), ids.to(dev), mask.to(dev), labels.to(dev))
    loss.backward()
    assert (logits.cpu() - out["logits"]).abs().max().item() < 1e-3          # north-star tolerance
    assert torch.equal(logits.argmax(1).cpu(), out["logits"].argmax(1))      # class indices bit-exact
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    worst = 0.0
    for key in sysm.CKPT_KEYS:
        named = dict(getattr(sysm, key).named_parameters())
        for n, v in leaf[key].items():
            if v.grad is None or n not in named:
                continue
            got = named[n].grad
            if got is None:
                assert v.grad.abs().max().item() == 0.0, f"{key}.{n}: missing gradient"
                continue
            denom = max(v.grad.abs().max().item(), 1e-3)
            rel = (got.cpu() - v.grad).abs().max().item() / denom
            worst = max(worst, rel)
            assert rel < 2e-2, f"{key}.{n}: gradient differs from the oracle (rel {rel:.3e})"
    # one AdamW step, lr 1e-3 with the reference's group multipliers
    opt = sysm.make_optimizer(lr=1e-3)
    opt.step()
    lr_mult = dict(audio_encoder=0.1, text_encoder=0.1, cross=1, pool_a=1, pool_t=1, fusion=1, prototypes=1)
    wd = dict(audio_encoder=.025, text_encoder=.025, cross=.05, pool_a=.05, pool_t=.05, fusion=.05, prototypes=.05)
    for key in ("cross", "fusion", "prototypes", "audio_encoder"):
        named = dict(getattr(sysm, key).named_parameters())
        for n, v in leaf[key].items():
            if v.grad is None or n not in named or named[n].grad is None:
                continue
            want, _, _ = O.adamw_step(v.detach(), v.grad, torch.zeros_like(v), torch.zeros_like(v), 1, 1e-3 * lr_mult[key], wd[key])
            # Adam's first step moves every weight by ~lr*sign(g): compare the update, not the weight
            upd_got, upd_want = named[n].detach().cpu() - v.detach(), want - v.detach()
            big = v.grad.abs() > 1e-6
            if big.any():
                assert (upd_got - upd_want)[big].abs().max().item() < 2e-5, f"{key}.{n}: AdamW update differs"


def test_graph_step_equals_eager_step():
    import __graft_entry__ as ge
    from ser_amd.system import TrainStepper
    from ser_amd.optim import WarmupCosine
    dev = torch.device("cuda:0")
    sys_a, wc, xc = ge._small_system(dev)
    sys_b, _, _ = ge._small_system(dev)
    sys_b.load_state_dict(sys_a.state_dict())
    sys_c, _, _ = ge._small_system(dev)
    sys_c.load_state_dict(sys_a.state_dict())
    steppers = []
    for s, graph, split in ((sys_a, False, False), (sys_b, True, False), (sys_c, True, True)):
        s.train()
        opt = s.make_optimizer(lr=1e-3)
        steppers.append(TrainStepper(s, opt, WarmupCosine(opt, 10, 0.0), use_graph=graph, split_backward=split))
    for it in range(3):
        batch = [t.to(dev) for t in _batch(100 + it)]
        la = steppers[0].step(*batch)
        lb = steppers[1].step(*batch)
        lc = steppers[2].step(*batch)        # the two-piece backward used under data parallelism
        torch.cuda.synchronize()
        assert abs(la.item() - lb.item()) < 1e-6, f"step {it}: eager {la.item()} vs graph {lb.item()}"
        assert abs(la.item() - lc.item()) < 1e-6, f"step {it}: eager {la.item()} vs split graph {lc.item()}"
    for (n, pa), (_, pb), (_, pc) in zip(sys_a.named_parameters(), sys_b.named_parameters(), sys_c.named_parameters()):
        assert torch.equal(pa, pb), f"{n}: parameters diverged between eager and graph stepping"
        assert torch.equal(pa, pc), f"{n}: parameters diverged between eager and split-graph stepping"


@pytest.mark.parametrize("group", [1, 2, 3])
def test_pipelined_stepping_is_bit_identical_to_sequential(group):
    """group = consecutive batches whose frozen-encoder forward runs as ONE pass (all their clips in every GEMM launch),
    issued one to two groups ahead of the head steps that consume it: the same losses and parameters, bit for bit, as
    stepping one batch at a time - a clip's features do not depend on what it is batched with.  11 batches: the stream
    does not end on a group boundary, so `drain()` also trains on a partly filled slot."""
    import __graft_entry__ as ge
    from ser_amd.system import PipelinedStepper, TrainStepper
    dev = torch.device("cuda:0")
    sys_a, _, _ = ge._small_system(dev)
    sys_b, _, _ = ge._small_system(dev)
    sys_b.load_state_dict(sys_a.state_dict())
    sys_a.train(); sys_b.train()
    oa, ob = sys_a.make_optimizer(lr=1e-3), sys_b.make_optimizer(lr=1e-3)
    seq = TrainStepper(sys_a, oa, use_graph=False)
    pipe = PipelinedStepper(sys_b, ob, group=group)
    n = 11
    batches = [[t.to(dev) for t in _batch(200 + i)] for i in range(n)]
    seq_losses = [seq.step(*b).item() for b in batches]
    assert pipe.prime == 2 * group
    for j in range(pipe.prime):
        pipe.feed(*batches[j])
    pipe_losses = [pipe.step(*batches[i]).item() for i in range(pipe.prime, n)]
    pipe_losses += [l.item() for l in pipe.drain()]
    torch.cuda.synchronize()
    assert pipe.pending == 0
    assert seq_losses == pipe_losses, (seq_losses, pipe_losses)
    for (nm, pa), (_, pb) in zip(sys_a.named_parameters(), sys_b.named_parameters()):
        assert torch.equal(pa, pb), f"{nm}: pipelined stepping diverged from sequential stepping"


def test_pipelined_and_graph_stepping_with_the_reference_default_front_end():
    """The audio encoder's gate flags on (the reference's default AudioEncoder(): quality gates + audio conditioning before
    Wav2Vec2, their projected features fused into the sequence): the front end's kernels are part of the captured encoder
    graph (PipelinedStepper) / step graph (TrainStepper), the raw features travel with the batch to the head step.  Same
    losses and parameters, bit for bit, as eager stepping through the same modules."""
    import __graft_entry__ as ge
    from ser_amd.system import PipelinedStepper, TrainStepper
    dev = torch.device("cuda:0")
    systems = [ge._small_system(dev, gates=True)[0] for _ in range(3)]
    for s in systems[1:]:
        s.load_state_dict(systems[0].state_dict())
    for s in systems:
        s.train()
    assert systems[0].gates_on() and len(systems[0].buckets()) == 8          # the gate bucket is one of them
    opts = [s.make_optimizer(lr=1e-3) for s in systems]
    seq = TrainStepper(systems[0], opts[0], use_graph=False)
    graph = TrainStepper(systems[1], opts[1], use_graph=True)
    pipe = PipelinedStepper(systems[2], opts[2], group=2)
    n = 7
    g = torch.Generator().manual_seed(5)
    batches = []
    for i in range(n):
        b = [t.to(dev) for t in _batch(400 + i)]
        t_ = torch.arange(b[0].shape[1], device=dev) / 16000.0
        b[0] = b[0] + 0.2 * torch.sin(2 * 3.14159265 * (180.0 + 20 * i) * t_)[None]        # something voiced for the gates to accept
        batches.append(b)
    lid = systems[0].language_features(None, batches[0][0].shape[0]).to(dev)
    seq_losses = [seq.step(*b, lid).item() for b in batches]
    graph_losses = [graph.step(*b, lid).item() for b in batches]
    for j in range(pipe.prime):
        pipe.feed(*batches[j], lid)
    pipe_losses = [pipe.step(*batches[i], lid).item() for i in range(pipe.prime, n)] + [l.item() for l in pipe.drain()]
    torch.cuda.synchronize()
    assert seq_losses == graph_losses, (seq_losses, graph_losses)
    assert seq_losses == pipe_losses, (seq_losses, pipe_losses)
    moved = 0
    for (nm, pa), (_, pb), (_, pc) in zip(*(s.named_parameters() for s in systems)):
        assert torch.equal(pa, pb) and torch.equal(pa, pc), f"{nm}: stepping paths diverged with the front end on"
    gate = systems[0].audio_encoder._gate_flat
    assert gate.gflat.abs().sum().item() > 0, "the gate modules must receive gradients"


def test_grouped_head_launches_match_the_two_stream_head():
    """`_ops.GROUPED_HEAD` (off by default: slower beside the encoder pass): both cross-attention directions, both adapters and
    both poolings as grouped launches on one stream (ser_linear_fwd_group / ser_linear_dgrad_group).  Same tile code per
    problem: the forward - and with it the loss - is bit-identical to the two-stream head; gradients agree to fp32 rounding
    (the three input-gradient contributions of a sequence are accumulated in a different order)."""
    import __graft_entry__ as ge
    from ser_amd import _ops as OP
    dev = torch.device("cuda:0")
    sys_a, _, _ = ge._small_system(dev, train_dropout=True)
    sys_b, _, _ = ge._small_system(dev, train_dropout=True)
    sys_b.load_state_dict(sys_a.state_dict())
    sys_a.train(); sys_b.train()
    batch = [t.to(dev) for t in _batch(31)]
    la, _ = sys_a.loss(*batch)
    la.backward()
    prev = OP.GROUPED_HEAD
    OP.GROUPED_HEAD = True
    try:
        lb, _ = sys_b.loss(*batch)
        lb.backward()
    finally:
        OP.GROUPED_HEAD = prev
    torch.cuda.synchronize()
    assert la.item() == lb.item()
    for (n, pa), (_, pb) in zip(sys_a.named_parameters(), sys_b.named_parameters()):
        if pa.grad is None:
            assert pb.grad is None, n
            continue
        scale = max(pa.grad.abs().max().item(), 1e-12)
        assert (pa.grad - pb.grad).abs().max().item() <= 2e-5 * scale, f"{n}: grouped head gradient differs"


def test_first_forward_on_busy_streams_keeps_every_parameter():
    """Regression for round 2's rare first-loss mismatch.  The trainable buckets used to be flattened lazily inside the first
    forward; the text-side ones (text adapter, text pooling) on SERSystem's side stream.  Re-pointing `p.data` frees the old
    parameter block to its allocation stream at once, the main stream's next bucket (same size = exact fit for the freed,
    merged block) reused it, and with the device lagging behind the host - here forced by a long spin on the main stream -
    the side stream's queued copy then read the sibling's zero fill or weights.  Every parameter must survive bit for bit."""
    import __graft_entry__ as ge
    dev = torch.device("cuda:0")
    sysm, _, _ = ge._small_system(dev)
    sysm.train()
    want = {n: p.detach().clone() for n, p in sysm.named_parameters()}
    batch = [t.to(dev) for t in _batch(77)]
    torch.cuda.synchronize()
    if hasattr(torch.cuda, "_sleep"):
        torch.cuda._sleep(400_000_000)        # ~0.2 s on the main stream: everything below queues up behind it
    loss, _ = sysm.loss(*batch)
    torch.cuda.synchronize()
    bad = [n for n, p in sysm.named_parameters() if not torch.equal(p.detach(), want[n])]
    assert not bad, f"parameters changed by the first forward: {bad[:6]}"
    # and the pieces flattened by hand on a side stream, as the lazy path did
    from ser_amd.models.pooling import AttentiveStatsPooling
    pa, pt = AttentiveStatsPooling(128).to(dev), AttentiveStatsPooling(128).to(dev)
    ref = {n: p.detach().clone() for n, p in pt.named_parameters()}
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    if hasattr(torch.cuda, "_sleep"):
        torch.cuda._sleep(400_000_000)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pt._flat.ensure()
    pa._flat.ensure()
    torch.cuda.synchronize()
    for n, p in pt.named_parameters():
        assert torch.equal(p.detach(), ref[n]), f"pool_t.{n} corrupted by a bucket flattened on another stream"


def test_variable_length_clips_pad_like_reference():
    import __graft_entry__ as ge
    dev = torch.device("cuda:0")
    sysm, wc, xc = ge._small_system(dev)
    sysm.eval()
    a_cfg, _ = ge.oracle_cfgs(wc, xc)
    g = torch.Generator().manual_seed(3)
    waves = [0.1 * torch.randn(n, generator=g) for n in (4000, 3200, 4000, 2400)]
    seq, mask = sysm.audio_encoder(waves, None)
    sd = {n: v.detach().cpu() for n, v in sysm.audio_encoder.state_dict().items()}
    want, wmask = O.audio_encoder_forward(sd, waves, a_cfg)
    assert seq.shape == want.shape and torch.equal(mask.cpu(), wmask)
    assert (seq.cpu() - want).abs().max().item() < 5e-4


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


def test_train_step_with_dropout_matches_oracle_with_the_same_masks():
    """Training-mode parity INCLUDING dropout: the oracle restates the mask generator (O.dropout_mult) and applies the
    masks of the same step at the same 13 + 2 x depth sites; logits, loss and every gradient must then agree with the
    HIP path (fused masks inside the persistent classifier and attention kernels) as tightly as without dropout."""
    import __graft_entry__ as ge
    from ser_amd import _ops as OP
    dev = torch.device("cuda:0")
    sysm, wc, xc = ge._small_system(dev, train_dropout=True)
    sysm.train()
    a_cfg, t_cfg = ge.oracle_cfgs(wc, xc)
    wave, ids, mask, labels = _batch(11)
    sds = {k: {n: v.detach().cpu().clone() for n, v in getattr(sysm, k).state_dict().items()} for k in sysm.CKPT_KEYS}

    loss, logits = sysm.loss(wave.to(dev), ids.to(dev), mask.to(dev), labels.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    state = int(sysm._drop_state.item())
    assert state == sysm.dropout_seed + 1
    # the generator itself: device multipliers == the oracle's restatement, bit for bit
    for site, shape, p in ((7, (4, 64), 0.15), (3, (4, 2, 12, 9), 0.1), (21, (16, 512), 0.15)):
        dev_m = OP.dropout_(torch.ones(shape, device=dev), (sysm._drop_state, p), site).cpu()
        assert torch.equal(dev_m, O.dropout_mult(state, site, shape, p)), (site, shape)

    plan = O.DropoutPlan(state, p_cross=sysm.cross.dropout.p, p_fusion=0.1, p_classifier=0.15)
    leaf = {k: {n: v.clone().requires_grad_(v.dtype.is_floating_point and not n.startswith("encoder."))
                for n, v in sd.items()} for k, sd in sds.items()}
    out = O.full_forward(leaf, list(wave), ids, mask, a_cfg, t_cfg, num_layers=3, heads=2, use_openmax=False, training=True,
                         drop=plan)
    ref_loss = O.train_loss(out["logits"], out["unc"], out["fused"], leaf["prototypes"]["prototypes"], labels, 4)
    ref_loss.backward()
    nodrop = O.full_forward(sds, list(wave), ids, mask, a_cfg, t_cfg, num_layers=3, heads=2, use_openmax=False, training=True)
    assert (out["logits"] - nodrop["logits"]).abs().max().item() > 1e-2, "the plan must actually drop something"

    assert (logits.cpu() - out["logits"]).abs().max().item() < 1e-3
    assert torch.equal(logits.argmax(1).cpu(), out["logits"].argmax(1))
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    for key in sysm.CKPT_KEYS:
        named = dict(getattr(sysm, key).named_parameters())
        for n, v in leaf[key].items():
            if v.grad is None or n not in named or named[n].grad is None:
                continue
            denom = max(v.grad.abs().max().item(), 1e-3)
            rel = (named[n].grad.cpu() - v.grad).abs().max().item() / denom
            assert rel < 2e-2, f"{key}.{n}: gradient differs from the oracle under dropout (rel {rel:.3e})"


def test_graph_stepping_over_several_input_shapes_equals_eager_stepping():
    """Real manifests give a few distinct (clip length, token count, batch size) shapes; TrainStepper keeps one captured
    graph set per shape but ONE optimizer graph.  The parameters outside the flat gradient buckets (the prototypes) must
    therefore receive their gradient in the same tensor whatever graph set ran - otherwise the optimizer graph reads the
    first shape's stale gradient.  Alternating three shapes, also with the caller clearing gradients between steps:
    bit-identical to eager stepping."""
    import __graft_entry__ as ge
    from ser_amd.system import TrainStepper
    dev = torch.device("cuda:0")
    sys_a, _, _ = ge._small_system(dev)
    sys_b, _, _ = ge._small_system(dev)
    sys_b.load_state_dict(sys_a.state_dict())
    sys_a.train(); sys_b.train()
    oa, ob = sys_a.make_optimizer(lr=1e-3), sys_b.make_optimizer(lr=1e-3)
    eager, graph = TrainStepper(sys_a, oa, use_graph=False), TrainStepper(sys_b, ob, use_graph=True)
    shapes = [dict(B=4, T=4000, S=9), dict(B=4, T=3200, S=7), dict(B=2, T=4000, S=9)]
    for it in range(7):
        batch = [t.to(dev) for t in _batch(400 + it, **shapes[it % 3])]
        la, lb = eager.step(*batch), graph.step(*batch)
        torch.cuda.synchronize()
        assert la.item() == lb.item(), f"step {it} (shape {shapes[it % 3]}): eager {la.item()} vs graph {lb.item()}"
        if it == 3:
            ob.zero_grad(set_to_none=True)          # what a hand-written loop would do between steps
    assert len(graph._graphs) == 3
    for (n, pa), (_, pb) in zip(sys_a.named_parameters(), sys_b.named_parameters()):
        assert torch.equal(pa, pb), f"{n}: parameters diverged between eager and multi-shape graph stepping"
