"""Training-mode dropout of the head (csrc: ser_dropout, the fused sites in persist.hip / head.hip): mask statistics,
determinism, fused-vs-standalone agreement, gradient consistency, behaviour under hipGraph replay.  Parity with the
reference's golden vectors is defined with dropout off, so these tests check the implementation
against itself and against the definition y = x * m / (1 - p)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def OP():
    import ser_amd  # noqa: F401
    from ser_amd import _ops
    assert torch.cuda.is_available()
    return _ops


def _state(v):
    return torch.full((1,), v, dtype=torch.int64, device="cuda")


def test_mask_statistics_and_determinism(OP):
    x = torch.ones(1 << 20, device="cuda")
    for p in (0.1, 0.15, 0.5):
        y = OP.dropout_(x.clone(), (_state(7), p), site=3)
        kept = y != 0
        assert abs(float(kept.float().mean()) - (1 - p)) < 3e-3
        np.testing.assert_allclose(y[kept].cpu().numpy(), 1 / (1 - p), rtol=1e-6)
        assert abs(float(y.mean()) - 1.0) < 5e-3                      # expectation preserved
    a = OP.dropout_(x.clone(), (_state(7), 0.1), site=3)
    assert torch.equal(a, OP.dropout_(x.clone(), (_state(7), 0.1), site=3))
    for other in ((_state(8), 0.1, 3), (_state(7), 0.1, 4), (_state(7 + (1 << 32)), 0.1, 3)):
        b = OP.dropout_(x.clone(), (other[0], other[1]), site=other[2])
        agree = float(((a != 0) == (b != 0)).float().mean())
        assert abs(agree - (0.9 * 0.9 + 0.1 * 0.1)) < 5e-3, "masks of different (state, site) must be independent"
    # no visible structure along the index: lag-1 and lag-512 correlations of the keep mask
    k = (a != 0).float() - 0.9
    for lag in (1, 512, 4096):
        assert abs(float((k[:-lag] * k[lag:]).mean()) / 0.09) < 5e-3
    assert OP.dropout_(x.clone(), None, site=3) is not None and torch.equal(OP.dropout_(x.clone(), None, 3), x)


@pytest.mark.parametrize("rows,D,depth", [(16, 512, 6), (5, 128, 3)])
def test_classifier_dropout_fused_stack_matches_standalone_launches(OP, rows, D, depth):
    """The persistent stack kernels apply the 2 x depth block dropouts inside their epilogues / operand reads; the
    per-Linear path applies the same masks with standalone launches.  Same generator, same sites -> same function."""
    from ser_amd.models.classifier import AdvancedOpenMaxClassifier
    torch.manual_seed(rows + D)
    m = AdvancedOpenMaxClassifier(input_dim=D, num_labels=4, num_layers=depth, base_dim=D, dropout=0.2).cuda().train()
    with torch.no_grad():
        for prm in m.parameters():
            prm.add_(0.05 * torch.randn_like(prm))
    x = torch.randn(rows, D, device="cuda", requires_grad=True)
    gl, gu = torch.randn(rows, 4, device="cuda"), torch.randn(rows, 1, device="cuda")
    st = _state(99)

    def run(use_stack, drop):
        OP.USE_STACK = use_stack
        try:
            m.zero_grad(set_to_none=True)
            x.grad = None
            with OP.dropout_scope(st if drop else None):
                logits, unc, _ = m(x, use_openmax=False, return_uncertainty=True)
            torch.autograd.backward([logits, unc], [gl, gu])
            torch.cuda.synchronize()
            return logits.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        finally:
            OP.USE_STACK = True

    a, b, off = run(True, True), run(False, True), run(True, False)
    assert all(int(sc[1]) == 0 for sc in m._stack_cache[3])
    assert (a[0] - off[0]).abs().max().item() > 1e-2, "dropout must change the training forward"

    def same(u, v, what):
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=2e-4, atol=3e-5 * (float(v.abs().max()) + 1e-12), err_msg=what)

    same(a[0], b[0], "logits")
    same(a[1], b[1], "input gradient")
    for k in a[2]:
        same(a[2][k], b[2][k], k)
    again = run(True, True)
    assert torch.equal(a[0], again[0]), "same state -> same masks"


def _directional_check(fn, params, rel=4e-2, eps=(2e-2, 5e-3, 1e-3)):
    """<grad, v> against central differences of the deterministic function fn(*params) -> scalar at several step sizes.
    The function is only piecewise smooth (ReLU kinks, dropout masks) and fp32, so a single step size is either biased by
    curvature / kinks (large step) or by rounding (small step): the analytic value has to agree with one of the
    estimates, or lie inside the range they span."""
    for prm in params:
        prm.grad = None
    out = fn()
    out.backward()
    ana = 0.0
    vs = []
    gen = torch.Generator().manual_seed(4242)       # CPU generator: the direction must not depend on what earlier tests drew from the device RNG
    for prm in params:
        v = torch.randn(prm.shape, generator=gen).to(prm.device)
        v /= v.norm()
        vs.append(v)
        ana += float((prm.grad * v).sum())
    nums = []
    with torch.no_grad():
        for e in eps:
            for prm, v in zip(params, vs):
                prm.add_(e * v)
            fp = float(fn())
            for prm, v in zip(params, vs):
                prm.sub_(2 * e * v)
            fm = float(fn())
            for prm, v in zip(params, vs):
                prm.add_(e * v)
            nums.append((fp - fm) / (2 * e))
    tol = lambda num: rel * max(abs(ana), abs(num)) + 1e-3
    close = any(abs(ana - num) <= tol(num) for num in nums)
    inside = min(nums) - tol(min(nums)) <= ana <= max(nums) + tol(max(nums))
    assert close or inside, f"analytic {ana} vs numeric {nums} at steps {eps}"


def test_fusion_and_cross_attention_gradients_are_consistent_under_dropout(OP):
    import ser_amd.models as M
    torch.manual_seed(1)
    st = _state(5)
    fus = M.FusionLayer(96, 96, 64).cuda().train()
    av, tv = torch.randn(8, 96, device="cuda", requires_grad=True), torch.randn(8, 96, device="cuda", requires_grad=True)
    w = torch.randn(8, 64, device="cuda")

    def f_fus():
        with OP.dropout_scope(st):
            return (fus(av, tv) * w).sum()
    with OP.dropout_scope(None):
        base = fus(av, tv).detach()
    with OP.dropout_scope(st):
        assert (fus(av, tv).detach() - base).abs().max().item() > 1e-3
    _directional_check(f_fus, [av, tv])

    from ser_amd.models.cross_attention import CrossModalAttention
    cr = CrossModalAttention(64, 64, shared_dim=64, num_heads=2, dropout=0.2).cuda().train()
    a, t = torch.randn(2, 40, 64, device="cuda", requires_grad=True), torch.randn(2, 9, 64, device="cuda", requires_grad=True)
    wa, wt = torch.randn(2, 40, 64, device="cuda"), torch.randn(2, 9, 64, device="cuda")

    def f_cr():
        with OP.dropout_scope(st):
            ya, yt = cr(a, t, torch.ones(2, 40, device="cuda"), torch.ones(2, 9, device="cuda"))
        return (ya * wa).sum() + (yt * wt).sum()
    with OP.dropout_scope(None):
        ya0, _ = cr(a, t, torch.ones(2, 40, device="cuda"), torch.ones(2, 9, device="cuda"))
    with OP.dropout_scope(st):
        ya1, _ = cr(a, t, torch.ones(2, 40, device="cuda"), torch.ones(2, 9, device="cuda"))
    assert (ya1 - ya0).abs().max().item() > 1e-3
    _directional_check(f_cr, [a, t])


def test_classifier_gradients_consistent_under_dropout_on_both_paths(OP):
    """Directional-derivative check of the whole classifier under a fixed mask: the persistent stack (M <= 16) and the
    launch-per-Linear path that larger batches take (M = 24 here)."""
    from ser_amd.models.classifier import AdvancedOpenMaxClassifier
    torch.manual_seed(3)
    st = _state(17)
    gen = torch.Generator().manual_seed(5)          # inputs from a CPU generator (a central difference through ReLU kinks is
    for rows in (8, 24):                            # only meaningful for a fixed, known-smooth draw)
        m = AdvancedOpenMaxClassifier(input_dim=64, num_labels=4, num_layers=2, base_dim=64, dropout=0.2).cuda().train()
        x = torch.randn(rows, 64, generator=gen).cuda().requires_grad_()
        wl, wu = torch.randn(rows, 4, generator=gen).cuda(), torch.randn(rows, 1, generator=gen).cuda()
        assert OP.stack_supported(2, rows, 64) == (rows <= 16)

        def f():
            with OP.dropout_scope(st):
                logits, unc, _ = m(x, use_openmax=False, return_uncertainty=True)
            return (logits * wl).sum() + (unc * wu).sum()
        _directional_check(f, [x] + [p for p in m.deep_classifier.parameters()][:6], rel=5e-2)


@pytest.mark.parametrize("rows", [8, 16, 24])
def test_classifier_dropout_against_torch_autograd_with_the_same_masks(OP, rows):
    """The multipliers of a site are observable (dropout of a ones tensor), so the whole classifier under dropout can be
    restated with torch ops in float64 and differentiated by autograd: outputs and every gradient must match the HIP
    path (persistent stack for rows <= 16, launch-per-Linear beyond) to fp32 accuracy."""
    import torch.nn.functional as F
    from ser_amd.models.classifier import AdvancedOpenMaxClassifier
    torch.manual_seed(11 + rows)
    D, depth, p = 128, 3, 0.25
    m = AdvancedOpenMaxClassifier(input_dim=D, num_labels=4, num_layers=depth, base_dim=D, dropout=p).cuda().train()
    with torch.no_grad():
        for prm in m.parameters():
            prm.add_(0.05 * torch.randn_like(prm))
    st = _state(1234)
    x = torch.randn(rows, D, device="cuda", requires_grad=True)
    gl, gu = torch.randn(rows, 4, device="cuda"), torch.randn(rows, 1, device="cuda")
    with OP.dropout_scope(st):
        logits, unc, _ = m(x, use_openmax=False, return_uncertainty=True)
    torch.autograd.backward([logits, unc], [gl, gu])
    torch.cuda.synchronize()
    got = {k: prm.grad.double().clone() for k, prm in m.named_parameters() if prm.grad is not None}
    gx = x.grad.double().clone()

    def mult(site, shape):
        return OP.dropout_(torch.ones(shape, device="cuda"), (st, p), site).double()

    sites = m._drop_sites
    P = {k: v.detach().double().requires_grad_() for k, v in m.named_parameters()}
    xd = x.detach().double().requires_grad_()
    dc = "deep_classifier."
    ln = lambda t_, pre: F.layer_norm(t_, (t_.shape[1],), P[pre + ".weight"], P[pre + ".bias"], 1e-5)
    lin = lambda t_, pre: t_ @ P[pre + ".weight"].t() + P[pre + ".bias"]
    h = F.relu(ln(lin(xd, dc + "input_projection.0"), dc + "input_projection.1")) * mult(sites[0], (rows, D))
    for i in range(depth):
        x1 = ln(h, dc + f"layer_norms.{i}")
        u = ln(x1, dc + f"residual_layers.{i}.block.0")
        a = F.relu(lin(u, dc + f"residual_layers.{i}.block.1")) * mult(sites[3] + 2 * i, (rows, D))
        h = x1 + lin(a, dc + f"residual_layers.{i}.block.4") * mult(sites[3] + 2 * i + 1, (rows, D))
    f = F.relu(ln(lin(h, dc + "output_projection.0"), dc + "output_projection.1")) * mult(sites[1], (rows, D // 2))
    ref_logits = lin(f, dc + "output_projection.4")
    uh = F.relu(lin(f, "uncertainty_head.0")) * mult(sites[2], (rows, 64))
    ref_unc = torch.sigmoid(lin(uh, "uncertainty_head.3"))
    torch.autograd.backward([ref_logits, ref_unc], [gl.double(), gu.double()])

    def close(a, b, what):
        a, b = a.detach(), b.detach()
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-5 * (float(b.abs().max()) + 1e-12), err_msg=what)

    close(logits.double(), ref_logits.detach(), "logits")
    close(unc.double(), ref_unc.detach(), "uncertainty")
    close(gx, xd.grad, "input gradient")
    for k, g_ in got.items():
        if P[k].grad is not None:
            close(g_, P[k].grad, k)


@pytest.mark.parametrize("Sa,St,heads,E", [(40, 9, 2, 64), (300, 20, 2, 64)])
def test_cross_attention_dropout_against_torch_autograd_with_the_same_masks(OP, Sa, St, heads, E):
    """Attention-probability dropout (inside the xattn kernels: LDS-staged path for S <= 256, generic beyond) and the
    block-output dropout, restated with torch ops in float64 using the observed multipliers of each site."""
    import torch.nn.functional as F
    from ser_amd.models.cross_attention import CrossModalAttention
    torch.manual_seed(Sa + St)
    B, D, p = 2, 64, 0.2
    m = CrossModalAttention(D, D, shared_dim=E, num_heads=heads, dropout=p).cuda().train()
    st = _state(77)
    a = torch.randn(B, Sa, D, device="cuda", requires_grad=True)
    t_ = torch.randn(B, St, D, device="cuda", requires_grad=True)
    am, tm = torch.ones(B, Sa, device="cuda"), torch.ones(B, St, device="cuda")
    tm[0, -2:] = 0
    ga, gt = torch.randn(B, Sa, D, device="cuda"), torch.randn(B, St, D, device="cuda")
    with OP.dropout_scope(st):
        ya, yt = m(a, t_, am, tm)
    torch.autograd.backward([ya, yt], [ga, gt])
    torch.cuda.synchronize()
    got = {k: prm.grad.double().clone() for k, prm in m.named_parameters() if prm.grad is not None}
    g_a, g_t = a.grad.double().clone(), t_.grad.double().clone()

    def mult(site, shape):
        return OP.dropout_(torch.ones(shape, device="cuda"), (st, p), site).double()

    P = {k: v.detach().double().requires_grad_() for k, v in m.named_parameters()}
    ad, td = a.detach().double().requires_grad_(), t_.detach().double().requires_grad_()
    lin = lambda x_, pre: x_ @ P[pre + ".weight"].t() + P[pre + ".bias"]
    hd = E // heads

    def direction(xq, xkv, kmask, q, k, v, attn, out, norm, site_out, site_attn):
        Sq, Sk = xq.shape[1], xkv.shape[1]
        Wi, bi = P[attn + ".in_proj_weight"], P[attn + ".in_proj_bias"]
        Q = lin(xq, q) @ Wi[:E].t() + bi[:E]
        K = lin(xkv, k) @ Wi[E:2 * E].t() + bi[E:2 * E]
        V = lin(xkv, v) @ Wi[2 * E:].t() + bi[2 * E:]
        split = lambda z, S: z.view(B, S, heads, hd).transpose(1, 2)
        sc = split(Q, Sq) @ split(K, Sk).transpose(-1, -2) / hd ** 0.5
        sc = sc.masked_fill(kmask[:, None, None, :] == 0, float("-inf"))
        Pm = torch.softmax(sc, -1) * mult(site_attn, (B, heads, Sq, Sk))
        ctx = (Pm @ split(V, Sk)).transpose(1, 2).reshape(B, Sq, E)
        o = lin(lin(ctx, attn + ".out_proj"), out) * mult(site_out, (B * Sq, D)).view(B, Sq, D)
        return F.layer_norm(xq + o, (D,), P[norm + ".weight"], P[norm + ".bias"], 1e-5)

    s_ = m._drop_sites
    ra = direction(ad, td, tm.double(), "q_a", "k_t", "v_t", "attn_a", "out_a", "norm_a", s_[0], s_[2])
    rt = direction(td, ad, am.double(), "q_t", "k_a", "v_a", "attn_t", "out_t", "norm_t", s_[1], s_[3])
    torch.autograd.backward([ra, rt], [ga.double(), gt.double()])

    def close(u, v, what):
        # the key biases have an exactly-zero gradient (softmax is shift invariant): absolute floor at fp32 round-off
        u, v = u.detach(), v.detach()
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=3e-4, atol=3e-5 * float(v.abs().max()) + 5e-6, err_msg=what)

    close(ya.double(), ra, "audio output")
    close(yt.double(), rt, "text output")
    close(g_a, ad.grad, "audio input gradient")
    close(g_t, td.grad, "text input gradient")
    for k, g_ in got.items():
        close(g_, P[k].grad, k)


def test_system_dropout_is_deterministic_and_replays_draw_new_masks():
    import __graft_entry__ as ge
    from ser_amd.system import TrainStepper
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    B, T, S = 4, 4000, 9
    wave = (0.1 * torch.randn(B, T, generator=g)).to(dev)
    ids = torch.randint(4, 1000, (B, S), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    ids, mask, labels = ids.to(dev), torch.ones(B, S, device=dev), torch.randint(0, 4, (B,), generator=g).to(dev)
    runs = {}
    for name, (graph, drop) in {"eager": (False, True), "eager2": (False, True), "graph": (True, True), "off": (False, False)}.items():
        sysm, _, _ = ge._small_system(dev, train_dropout=drop)
        sysm.train()
        opt = sysm.make_optimizer(1e-3)
        stp = TrainStepper(sysm, opt, None, None, use_graph=graph)
        losses = [float(stp.step(wave, ids, mask, labels))]
        s0 = int(sysm._drop_state.item()) if drop else None
        losses += [float(stp.step(wave, ids, mask, labels)) for _ in range(3)]
        if drop:   # the generator state advances by exactly one per step, also when the step is a hipGraph replay
            assert int(sysm._drop_state.item()) == s0 + 3, name
        runs[name] = losses
    assert runs["eager"] == runs["eager2"], "same seed, same sites -> same masks"
    assert all(abs(a - b) > 1e-5 for a, b in zip(runs["eager"], runs["off"])), "dropout must be active in training steps"
    assert all(abs(a - b) > 1e-5 for a, b in zip(runs["graph"], runs["off"])), "dropout must be active in replayed steps"
    # in eval mode the same system is deterministic and dropout-free
    sysm.eval()
