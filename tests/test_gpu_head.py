"""GPU parity of the trainable head (forward AND backward, through the C ABI) against the golden
vectors captured from the reference modules (tests/golden/*.npz)."""
import numpy as np
import pytest
import torch

from tests.helpers import load_npz, split_fixture, t

pytestmark = pytest.mark.gpu

FWD = dict(atol=3e-5, rtol=2e-4)
BWD = dict(atol=3e-4, rtol=2e-3)


@pytest.fixture(scope="module")
def M():
    import ser_amd  # noqa: F401
    import ser_amd.models as models
    assert torch.cuda.is_available()
    return models


def _close(got, want, msg="", **tol):
    np.testing.assert_allclose(got.detach().cpu().numpy(), np.asarray(want), err_msg=msg, **tol)


def _check_param_grads(module, grads):
    named = dict(module.named_parameters())
    for k, gw in grads.items():
        p = named[k]
        if p.grad is None:
            assert float(gw.abs().max()) == 0.0, f"{k}: reference has a non-zero gradient, build has none"
            continue
        _close(p.grad, gw.numpy(), msg=k, **BWD)


def test_cross_attention(M):
    from ser_amd.models.cross_attention import CrossModalAttention
    sd, gr, r = split_fixture(load_npz("cross.npz"))
    m = CrossModalAttention(128, 128, shared_dim=64, num_heads=int(r["heads"]))
    m.load_state_dict(sd)
    m = m.cuda()
    a, tt = t(r["a"]).cuda().requires_grad_(), t(r["t"]).cuda().requires_grad_()
    ae, te = m(a, tt, t(r["a_mask"]).cuda(), t(r["t_mask"]).cuda())
    _close(ae, r["a_enh"], **FWD)
    _close(te, r["t_enh"], **FWD)
    torch.autograd.backward([ae, te], [t(r["g_a_enh"]).cuda(), t(r["g_t_enh"]).cuda()])
    _close(a.grad, r["grad_a"], **BWD)
    _close(tt.grad, r["grad_t"], **BWD)
    _check_param_grads(m, gr)


def test_pooling(M):
    from ser_amd.models.pooling import AttentiveStatsPooling
    sd, gr, r = split_fixture(load_npz("pool.npz"))
    m = AttentiveStatsPooling(128, hidden_dim=32)
    m.load_state_dict(sd)
    m = m.cuda()
    x = t(r["x"]).cuda().requires_grad_()
    y = m(x, t(r["mask"]).cuda())
    _close(y, r["y"], **FWD)
    y.backward(t(r["g_y"]).cuda())
    _close(x.grad, r["grad_x"], **BWD)
    _check_param_grads(m, gr)


def test_fusion(M):
    sd, gr, r = split_fixture(load_npz("fusion.npz"))
    m = M.FusionLayer(256, 256, 64)
    m.load_state_dict(sd)
    m = m.cuda()
    a, tv = t(r["a_vec"]).cuda().requires_grad_(), t(r["t_vec"]).cuda().requires_grad_()
    f = m(a, tv)
    _close(f, r["fused"], **FWD)
    f.backward(t(r["g_fused"]).cuda())
    _close(a.grad, r["grad_a_vec"], **BWD)
    _close(tv.grad, r["grad_t_vec"], **BWD)
    _check_param_grads(m, gr)


def test_classifier_train_and_openmax(M):
    from ser_amd.models.classifier import AdvancedOpenMaxClassifier
    sd, gr, r = split_fixture(load_npz("classifier.npz"))
    C, Lyr = int(r["num_labels"]), int(r["num_layers"])
    m = AdvancedOpenMaxClassifier(input_dim=64, num_labels=C, num_layers=Lyr, base_dim=64, dropout=0.15)
    m.load_state_dict(sd)
    m = m.cuda().train()
    x = t(r["x"]).cuda().requires_grad_()
    logits, unc, anchor = m(x, use_openmax=False, return_uncertainty=True)
    _close(logits, r["logits"], **FWD)
    _close(unc, r["unc"], **FWD)
    assert float(anchor) == 0.0
    torch.autograd.backward([logits, unc], [t(r["g_logits"]).cuda(), t(r["g_unc"]).cuda()])
    _close(x.grad, r["grad_x"], **BWD)
    _check_param_grads(m, gr)
    # class indices bit-exact
    assert torch.equal(logits.argmax(1).cpu(), t(r["logits"]).argmax(1))
    m.eval()
    with torch.no_grad():
        _close(m(t(r["x"]).cuda(), use_openmax=True), r["logits_openmax_unfitted"], **FWD)
        m.fit_weibull(t(r["fit_feats"]).cuda(), t(r["fit_labels"]).cuda())
        for k in ("weibull_alpha", "weibull_beta", "weibull_tau", "activation_vectors"):
            _close(getattr(m, k), r["fitted." + k], atol=1e-6, rtol=1e-5)
        _close(m(t(r["x"]).cuda(), use_openmax=True), r["logits_openmax_fitted"], **FWD)


def test_second_backward_accumulates(M):
    sd, gr, r = split_fixture(load_npz("fusion.npz"))
    m = M.FusionLayer(256, 256, 64)
    m.load_state_dict(sd)
    m = m.cuda()
    for _ in range(2):
        f = m(t(r["a_vec"]).cuda(), t(r["t_vec"]).cuda())
        f.backward(t(r["g_fused"]).cuda())
    named = dict(m.named_parameters())
    for k, gw in gr.items():
        _close(named[k].grad, 2 * gw.numpy(), msg=k, atol=6e-4, rtol=2e-3)
    m.zero_grad(set_to_none=True)
    f = m(t(r["a_vec"]).cuda(), t(r["t_vec"]).cuda())
    f.backward(t(r["g_fused"]).cuda())
    for k, gw in gr.items():
        _close(named[k].grad, gw.numpy(), msg=k, **BWD)


def test_train_loss_value_and_grads(M):
    from ser_amd.models.losses import TrainLoss, LabelSmoothingCrossEntropy, ClassBalancedFocalLoss
    from ser_amd.models.prototypes import PrototypeMemory
    r = load_npz("losses.npz")
    lg, un, fu = (t(r[k]).cuda().requires_grad_() for k in ("logits", "unc", "fused"))
    lab = t(r["labels"]).cuda()
    C = lg.shape[1]
    pm = PrototypeMemory(C, fu.shape[1]).cuda()
    with torch.no_grad():
        pm.prototypes.copy_(t(r["prototypes"]))
    crit = TrainLoss(C)
    total = crit(lg, un, fu, pm.prototypes, lab, use_proto=True)
    parts = crit.last_parts.cpu().numpy()
    assert abs(float(total) - float(r["total"])) < 2e-5
    for i, k in enumerate(("total", "ce", "focal", "unc_loss", "proto")):
        assert abs(parts[i] - float(r[k])) < 1e-4, (k, parts[i], float(r[k]))
    total.backward()
    _close(lg.grad, r["grad_logits"], atol=2e-6, rtol=1e-3)
    _close(un.grad, r["grad_unc"], atol=1e-7, rtol=1e-3)
    _close(fu.grad, r["grad_fused"], atol=1e-7, rtol=1e-3)
    _close(pm.prototypes.grad, r["grad_prototypes"], atol=1e-7, rtol=1e-3)
    # reference-style separate criteria
    assert abs(float(LabelSmoothingCrossEntropy(0.1)(t(r["logits"]).cuda(), lab)) - float(r["ce"])) < 2e-5
    assert abs(float(ClassBalancedFocalLoss(num_classes=C)(t(r["logits"]).cuda(), lab)) - float(r["focal"])) < 1e-4   # value ~2.9: 3e-5 relative
    assert abs(float(pm.prototype_loss(t(r["fused"]).cuda(), lab)) - float(r["proto"])) < 1e-4


def test_adamw_matches_torch_trajectory(M):
    from ser_amd import _ops as O
    r = load_npz("adamw.npz")
    for name, mult, wd, col in (("p1", 1.5, 0.06, 0), ("p2", 1.0, 0.05, 1)):
        p = t(r[f"{name}_0"]).cuda().contiguous()
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for s in range(4):
            base_lr = float(r["lrs"][s, col]) / mult
            step = s + 1
            hyper = torch.tensor([base_lr, 1 - 0.9 ** step, (1 - 0.999 ** step) ** 0.5], dtype=torch.float32).cuda()
            O.adamw_(p, t(r[f"g{name[1]}_{s}"]).cuda().contiguous(), m, v, hyper, mult, wd)
            _close(p, r[f"{name}_{s + 1}"], atol=2e-6, rtol=2e-5)


@pytest.mark.parametrize("M_,N,K", [(16, 512, 512), (3184, 256, 768), (50, 1, 128), (7, 4, 256), (130, 70, 33)])
def test_gemm_f32_three_forms(M, M_, N, K):
    """y = x W^T, dx = dy W, dW = dy^T x through ser_linear_*: token-level shapes run on the split-bf16 MFMA kernel
    (three products per MAC, ~2^-16 relative per product), M <= 16 and odd shapes on the exact fp32 MFMA kernels."""
    from ser_amd import _ops as O
    g = torch.Generator().manual_seed(3)
    x, dy = torch.randn(M_, K, generator=g), torch.randn(M_, N, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5            # realistic fan-in scaling
    b = torch.randn(N, generator=g)
    y = O.linear_fwd(x.cuda(), w.cuda(), b.cuda(), O.ACT_TANH)
    _close(y, torch.tanh(x.double() @ w.double().t() + b.double()).float(), atol=3e-5, rtol=1e-4)
    dx = O.linear_dgrad(dy.cuda(), w.cuda())
    _close(dx, (dy.double() @ w.double()).float(), atol=5e-5, rtol=1e-4)
    dW, db = torch.zeros(N, K).cuda(), torch.zeros(N).cuda()
    O.linear_wgrad(dy.cuda(), x.cuda(), dW, db, accumulate=False)
    O.linear_wgrad(dy.cuda(), x.cuda(), dW, db, accumulate=True)
    ref = 2 * (dy.double().t() @ x.double())
    _close(dW, ref.float(), atol=3e-5 * float(ref.abs().max()), rtol=2e-4)
    _close(db, 2 * dy.double().sum(0).float(), atol=1e-5 * M_ ** 0.5 * 4, rtol=2e-4)


@pytest.mark.parametrize("hd", [32, 64])
@pytest.mark.parametrize("Sq,Sk", [(199, 32), (32, 199), (300, 40), (40, 300), (7, 256), (256, 5), (199, 199)])
def test_xattn_core_fast_and_generic_paths(M, Sq, Sk, hd):
    """softmax(QK^T/sqrt(d) + mask) V and its backward: S <= 256 runs the LDS-staged kernels (32-wide heads: the cross-modal
    attention; 64-wide: the encoders' self-attention in the fine-tuning path), longer sequences the generic ones; both
    against an fp64 autograd reference."""
    from ser_amd import _ops as O
    B, heads = 2, 4
    E = heads * hd
    g = torch.Generator().manual_seed(Sq * 1000 + Sk)
    q, k, v = (torch.randn(B * n, E, generator=g) for n in (Sq, Sk, Sk))
    mask = torch.ones(B, Sk)
    mask[1, max(1, Sk // 2):] = 0
    dctx = torch.randn(B * Sq, E, generator=g)
    ctx, P = O.xattn_fwd(q.cuda(), k.cuda(), v.cuda(), mask.cuda(), B, Sq, Sk, heads)
    dq, dk, dv = O.xattn_bwd(dctx.cuda(), q.cuda(), k.cuda(), v.cuda(), P, B, Sq, Sk, heads)
    qd, kd, vd = (t_.double().requires_grad_() for t_ in (q, k, v))
    qh = qd.view(B, Sq, heads, hd).transpose(1, 2)
    kh = kd.view(B, Sk, heads, hd).transpose(1, 2)
    vh = vd.view(B, Sk, heads, hd).transpose(1, 2)
    s = qh @ kh.transpose(2, 3) / hd ** 0.5
    s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * Sq, E)
    ref.backward(dctx.double())
    _close(ctx, ref.detach().float(), atol=2e-5, rtol=1e-4)
    _close(dq, qd.grad.float(), atol=5e-5, rtol=1e-3)
    _close(dk, kd.grad.float(), atol=5e-5, rtol=1e-3)
    _close(dv, vd.grad.float(), atol=5e-5, rtol=1e-3)


def test_gate_feature_fusion_learnable_path(M):
    """ref audio_encoder.py:115-132 driven with raw feature vectors (the DSP that produces them is out of scope):
    forward and all gradients vs the golden vectors captured from the reference's own layers."""
    from transformers import Wav2Vec2Config
    from ser_amd.models import AudioEncoder
    sd, gr, r = split_fixture(load_npz("gate_fusion.npz"))
    wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                        conv_dim=[64] * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
    ae = AudioEncoder(hf_config=wc, adapter_dim=32)           # gate flags on (the reference default)
    missing, unexpected = ae.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("encoder.") for k in missing)
    ae = ae.cuda()
    seq = t(r["seq"]).cuda().requires_grad_()
    out = ae.fuse_gate_features(seq, t(r["q_raw"]).cuda(), t(r["c_raw"]).cuda())
    _close(out, r["out"], **FWD)
    out.backward(t(r["g_out"]).cuda())
    _close(seq.grad, r["grad_seq"], **BWD)
    named = dict(ae.named_parameters())
    for k, gw in gr.items():
        assert named[k].grad is not None, k
        _close(named[k].grad, gw.numpy(), msg=k, **BWD)


@pytest.mark.parametrize("rows,D,depth", [(16, 512, 35), (5, 512, 4), (16, 64, 3), (3, 128, 2), (1, 256, 1)])
def test_classifier_persistent_stack_equals_per_block_launches(M, rows, D, depth):
    """The one-launch walk of the residual stack (csrc/persist.hip) against the launch-per-Linear path: same
    arithmetic, but the LayerNorm row sums are accumulated in a different order (operand layout + LDS reduction instead
    of one wave per row), so logits and gradients agree to a few ulps per block (checked at 2e-5 / 1e-4 relative after
    35 blocks); a second backward must accumulate; the hand-off waits must never have been abandoned."""
    from ser_amd import _ops as OP
    from ser_amd.models.classifier import AdvancedOpenMaxClassifier
    torch.manual_seed(rows * 1000 + D + depth)
    m = AdvancedOpenMaxClassifier(input_dim=D, num_labels=4, num_layers=depth, base_dim=D).cuda().train()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    x = torch.randn(rows, D, device="cuda", requires_grad=True)
    gl, gu = torch.randn(rows, 4, device="cuda"), torch.randn(rows, 1, device="cuda")

    def run(flag, twice=False):
        OP.USE_STACK = flag
        try:
            m.zero_grad(set_to_none=True)
            x.grad = None
            for _ in range(2 if twice else 1):
                logits, unc, _ = m(x, use_openmax=False, return_uncertainty=True)
                torch.autograd.backward([logits, unc], [gl, gu])
            torch.cuda.synchronize()
            return logits.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()
                                                             if p.grad is not None}
        finally:
            OP.USE_STACK = True

    assert OP.stack_supported(depth, rows, D)
    a, b = run(True), run(False)
    assert all(int(sc[1]) == 0 for sc in m._stack_cache[3]), "a hand-off wait was abandoned"
    assert all(int(sc[0]) >= 1 for sc in m._stack_cache[3]), "launch counters must advance"
    # row statistics are summed in a different order than in the per-Linear kernels: agreement to a few ulps
    np.testing.assert_allclose(a[0].cpu().numpy(), b[0].cpu().numpy(), rtol=2e-5, atol=2e-5)

    def same(u, v, what):
        scale = float(v.abs().max()) + 1e-12
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale, err_msg=what)

    same(a[1], b[1], "input gradient")
    assert a[2].keys() == b[2].keys()
    for k in a[2]:
        same(a[2][k], b[2][k], k)
    a2 = run(True, twice=True)
    for k in a2[2]:
        same(a2[2][k], 2 * a[2][k], "accumulated " + k)


@pytest.mark.parametrize("M_,N,K", [(3184, 256, 768), (512, 768, 256), (130, 72, 36)])
def test_backward_token_gemms_in_one_product_mode(M, M_, N, K):
    """`bf16` precision mode: dgrad / wgrad of the token-level Linears round their operands to bf16 (one MFMA product,
    fp32 accumulation).  Against float64 on the bf16-rounded operands they are exact to fp32 accumulation error;
    against the unrounded operands the error is the bf16 operand rounding (~2^-9 relative per factor)."""
    from ser_amd import _lib as L, _ops as OP
    g = torch.Generator().manual_seed(M_ + N)
    dy = torch.randn(M_, N, generator=g).cuda()
    x = torch.randn(M_, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bf = lambda t_: t_.to(torch.bfloat16).double()
    try:
        L.check(L.lib.ser_set_head_backward_products(1), "set")
        assert L.lib.ser_get_head_backward_products() == 1
        dx = OP.linear_dgrad(dy, W)
        dW, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
        OP.linear_wgrad(dy, x, dW, db)
        torch.cuda.synchronize()
    finally:
        L.lib.ser_set_head_backward_products(3)
    if N % 64 == 0 and K % 64 == 0:      # shapes the split-bf16 MFMA kernels take (others fall back to exact fp32 MFMA)
        ref_dx, ref_dW = bf(dy) @ bf(W), bf(dy).t() @ bf(x)
        assert (dx.double() - ref_dx).abs().max().item() < 2e-4 * (N ** 0.5)
        assert (dW.double() - ref_dW).abs().max().item() < 2e-4 * (M_ ** 0.5)
        assert (db.double() - bf(dy).sum(0)).abs().max().item() < 2e-4 * (M_ ** 0.5)
    fullW = dy.double().t() @ x.double()
    assert (dW.double() - fullW).abs().max().item() < 4e-2 * fullW.abs().max().item()
    # and the bf16 rounding itself stays at the 1e-3 level of the result's scale
    full = dy.double() @ W.double()
    assert (dx.double() - full).abs().max().item() < 4e-2 * full.abs().max().item()
    # 3-product default: fp32-equivalent
    dx3 = OP.linear_dgrad(dy, W)
    assert (dx3.double() - full).abs().max().item() < 2e-4 * full.abs().max().item()


@pytest.mark.parametrize("B,C,T", [(5, 4, 1.0), (33, 6, 0.37), (1, 2, 12.5), (64, 64, 2.0)])
def test_eval_consumers_against_torch(B, C, T):
    """/T, softmax, arg-max and the energy score -logsumexp (ref eval.py:192-206, utils.py:11-14) in one launch."""
    import ser_amd  # noqa: F401
    from ser_amd import _ops as OP
    g = torch.Generator().manual_seed(B * 100 + C)
    lg = 3.0 * torch.randn(B, C, generator=g)
    lg[0, :2] = lg[0].max() + 1.0                       # a tie: torch.argmax returns the first index
    pr, pd, en = OP.eval_consumers(lg.cuda(), T)
    z = lg.double() / T
    assert (pr.cpu().double() - torch.softmax(z, 1)).abs().max().item() < 2e-6
    assert torch.equal(pd.cpu(), z.argmax(1))
    assert (en.cpu().double() + torch.logsumexp(z, 1)).abs().max().item() < 1e-5


def test_temperature_grid_against_the_reference_loop():
    """--calibrate (ref eval.py:49-67): the 100 objectives of the grid in one launch, and the temperature the reference's
    loop would pick (first strict minimum)."""
    import ser_amd  # noqa: F401
    from ser_amd import _ops as OP
    from ser_amd import eval as E
    g = torch.Generator().manual_seed(8)
    N, C = 300, 6
    lg = 2.5 * torch.randn(N, C, generator=g)
    y = torch.where(torch.rand(N, generator=g) < 0.6, lg.argmax(1), torch.randint(0, C, (N,), generator=g))
    temps = torch.logspace(-1, 2, 100)
    ece = OP.temperature_grid(lg.cuda(), y.cuda(), temps.cuda()).cpu()
    want, best_t, best = [], 1.0, float("inf")
    for t_ in temps:
        probs = torch.softmax(lg / t_, dim=1)
        conf, preds = probs.max(dim=1)
        e = torch.mean(torch.abs(conf - (preds == y).float()))
        want.append(float(e))
        if e < best:
            best, best_t = e, t_.item()
    assert (ece - torch.tensor(want)).abs().max().item() < 2e-6
    assert abs(E.find_optimal_temperature(lg.cuda(), y.cuda()) - best_t) < 1e-6
