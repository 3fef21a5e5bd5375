"""Pins oracle/ser_oracle.py to the golden vectors captured from the reference
(tests/golden/make_fixtures.py).  CPU only."""
import numpy as np
import torch

from oracle import ser_oracle as O
from tests.helpers import cfg_of, load_npz, split_fixture, t

TOL = dict(atol=2e-5, rtol=1e-4)


def test_audio_encoder():
    z = load_npz("audio_encoder.npz")
    sd, _, r = split_fixture(z)
    seq, mask = O.audio_encoder_forward(sd, [t(r["wave0"]), t(r["wave1"])], cfg_of(r))
    assert seq.shape == r["a_seq"].shape
    np.testing.assert_allclose(seq.numpy(), r["a_seq"], **TOL)
    np.testing.assert_array_equal(mask.numpy(), r["a_mask"])


def test_gate_feature_fusion_fwd_bwd():
    sd, gr, r = split_fixture(load_npz("gate_fusion.npz"))
    p = _leafs(sd)
    seq = t(r["seq"]).requires_grad_()
    out = O.gate_feature_fusion(p, seq, t(r["q_raw"]), t(r["c_raw"]))
    np.testing.assert_allclose(out.detach().numpy(), r["out"], **TOL)
    (out * t(r["g_out"])).sum().backward()
    np.testing.assert_allclose(seq.grad.numpy(), r["grad_seq"], atol=1e-5, rtol=1e-4)
    for k, g in gr.items():
        np.testing.assert_allclose(p[k].grad.numpy(), g.numpy(), atol=1e-4, rtol=1e-3, err_msg=k)


def test_text_encoder():
    z = load_npz("text_encoder.npz")
    sd, _, r = split_fixture(z)
    seq, mask = O.text_encoder_forward(sd, t(r["input_ids"]), t(r["attention_mask"]), cfg_of(r))
    np.testing.assert_allclose(seq.numpy(), r["t_seq"], **TOL)
    np.testing.assert_array_equal(mask.numpy(), r["t_mask"])


def _leafs(sd):
    return {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}


def test_cross_attention_fwd_bwd():
    sd, gr, r = split_fixture(load_npz("cross.npz"))
    p = _leafs(sd)
    a, tt = t(r["a"]).requires_grad_(), t(r["t"]).requires_grad_()
    ae, te = O.cross_attention_forward(p, a, tt, t(r["a_mask"]), t(r["t_mask"]), int(r["heads"]))
    np.testing.assert_allclose(ae.detach().numpy(), r["a_enh"], **TOL)
    np.testing.assert_allclose(te.detach().numpy(), r["t_enh"], **TOL)
    ((ae * t(r["g_a_enh"])).sum() + (te * t(r["g_t_enh"])).sum()).backward()
    np.testing.assert_allclose(a.grad.numpy(), r["grad_a"], atol=1e-4, rtol=1e-3)
    np.testing.assert_allclose(tt.grad.numpy(), r["grad_t"], atol=1e-4, rtol=1e-3)
    for k, g in gr.items():
        np.testing.assert_allclose(p[k].grad.numpy(), g.numpy(), atol=2e-4, rtol=1e-3, err_msg=k)


def test_pooling_fwd_bwd():
    sd, gr, r = split_fixture(load_npz("pool.npz"))
    p = _leafs(sd)
    x = t(r["x"]).requires_grad_()
    y = O.pooling_forward(p, x, t(r["mask"]))
    np.testing.assert_allclose(y.detach().numpy(), r["y"], **TOL)
    (y * t(r["g_y"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), r["grad_x"], atol=1e-4, rtol=1e-3)
    for k, g in gr.items():
        np.testing.assert_allclose(p[k].grad.numpy(), g.numpy(), atol=1e-4, rtol=1e-3, err_msg=k)


def test_fusion_fwd_bwd():
    sd, gr, r = split_fixture(load_npz("fusion.npz"))
    p = _leafs(sd)
    a, tv = t(r["a_vec"]).requires_grad_(), t(r["t_vec"]).requires_grad_()
    f = O.fusion_forward(p, a, tv)
    np.testing.assert_allclose(f.detach().numpy(), r["fused"], **TOL)
    (f * t(r["g_fused"])).sum().backward()
    np.testing.assert_allclose(a.grad.numpy(), r["grad_a_vec"], atol=1e-4, rtol=1e-3)
    for k, g in gr.items():
        np.testing.assert_allclose(p[k].grad.numpy(), g.numpy(), atol=1e-4, rtol=1e-3, err_msg=k)


def test_classifier_fwd_bwd_openmax():
    sd, gr, r = split_fixture(load_npz("classifier.npz"))
    L = int(r["num_layers"])
    p = _leafs(sd)
    x = t(r["x"]).requires_grad_()
    logits, unc, anchor, feats = O.classifier_forward(p, x, L, use_openmax=False, training=False)
    np.testing.assert_allclose(logits.detach().numpy(), r["logits"], **TOL)
    np.testing.assert_allclose(unc.detach().numpy(), r["unc"], **TOL)
    assert float(anchor) == float(r["anchor_loss"]) == 0.0
    ((logits * t(r["g_logits"])).sum() + (unc * t(r["g_unc"])).sum()).backward()
    np.testing.assert_allclose(x.grad.numpy(), r["grad_x"], atol=1e-4, rtol=1e-3)
    for k, g in gr.items():
        got = p[k].grad
        if got is None:   # anchor branch: zero gradient in the reference (loss is identically 0)
            assert float(g.abs().max()) == 0.0, k
            continue
        np.testing.assert_allclose(got.numpy(), g.numpy(), atol=2e-4, rtol=1e-3, err_msg=k)
    with torch.no_grad():
        lo, *_ = O.classifier_forward(sd, t(r["x"]), L, use_openmax=True, training=False)
        np.testing.assert_allclose(lo.numpy(), r["logits_openmax_unfitted"], **TOL)
        fitted = O.fit_weibull(t(r["fit_feats"]), t(r["fit_labels"]), int(r["num_labels"]), sd)
        for k, v in fitted.items():
            np.testing.assert_allclose(v.numpy(), r["fitted." + k], atol=1e-6, rtol=1e-5, err_msg=k)
        sd2 = dict(sd); sd2.update(fitted)
        lo2, *_ = O.classifier_forward(sd2, t(r["x"]), L, use_openmax=True, training=False)
        np.testing.assert_allclose(lo2.numpy(), r["logits_openmax_fitted"], **TOL)


def test_losses_value_and_grads():
    r = load_npz("losses.npz")
    lg, un, fu = (t(r[k]).requires_grad_() for k in ("logits", "unc", "fused"))
    pr = t(r["prototypes"]).requires_grad_()
    lab = t(r["labels"])
    C = lg.shape[1]
    assert abs(float(O.label_smoothing_ce(lg, lab)) - float(r["ce"])) < 1e-5
    assert abs(float(O.class_balanced_focal(lg, lab, C)) - float(r["focal"])) < 1e-5
    assert abs(float(O.prototype_loss(pr, fu, lab)) - float(r["proto"])) < 1e-4
    total = O.train_loss(lg, un, fu, pr, lab, C)
    assert abs(float(total) - float(r["total"])) < 1e-5
    total.backward()
    np.testing.assert_allclose(lg.grad.numpy(), r["grad_logits"], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(un.grad.numpy(), r["grad_unc"], atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(fu.grad.numpy(), r["grad_fused"], atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(pr.grad.numpy(), r["grad_prototypes"], atol=1e-7, rtol=1e-4)


def test_adamw_and_lr_schedule():
    r = load_npz("adamw.npz")
    total, wr = int(r["total_steps"]), float(r["warmup_ratio"])
    lam = [O.lr_lambda(s, total, wr) for s in range(total + 1)]
    np.testing.assert_allclose(lam, r["lambda_values"], atol=1e-7)
    for name, base_lr, wd, col in (("p1", 1.5e-3, 0.06, 0), ("p2", 1e-3, 0.05, 1)):
        p = t(r[f"{name}_0"]); m = torch.zeros_like(p); v = torch.zeros_like(p)
        for s in range(4):
            lr = base_lr * lam[s]
            assert abs(lr - r["lrs"][s, col]) < 1e-12
            p, m, v = O.adamw_step(p, t(r[f"g{name[1]}_{s}"]), m, v, s + 1, lr, wd)
            np.testing.assert_allclose(p.numpy(), r[f"{name}_{s + 1}"], atol=1e-6, rtol=1e-5)


def test_dropout_mask_generator_restatement_properties():
    """O.dropout_mult restates the device mask generator (csrc/ser_common.h: ser_drop_mult); the bit-for-bit check
    against the device is in tests/test_gpu_system.py.  Here: keep rate, scaling, determinism, independence."""
    import numpy as np
    a = O.dropout_mult(0x5EEE, 7, (256, 512), 0.15)
    assert abs(float((a != 0).float().mean()) - 0.85) < 3e-3
    np.testing.assert_allclose(a[a != 0].numpy(), 1 / 0.85, rtol=1e-6)
    assert torch.equal(a, O.dropout_mult(0x5EEE, 7, (256, 512), 0.15))
    for st, site in ((0x5EEF, 7), (0x5EEE, 8), (0x5EEE + (1 << 32), 7)):
        b = O.dropout_mult(st, site, (256, 512), 0.15)
        agree = float(((a != 0) == (b != 0)).float().mean())
        assert abs(agree - (0.85 ** 2 + 0.15 ** 2)) < 5e-3
    assert float(O.dropout_mult(1, 1, (1000,), 0.0).min()) == 1.0


def test_cpu_train_step_applies_the_dropout_plan_deterministically():
    """oracle/cpu_step.py (bench.py's cpu_baseline leg): with a dropout seed the step draws the head's masks from the
    build's generator (state = seed + step) — same seed same losses, other seed / no dropout other losses."""
    import __graft_entry__ as ge
    from oracle.cpu_step import OracleTrainer
    sysm, wc, xc = ge._small_system("cpu")
    sds = {k: {n: v.detach() for n, v in getattr(sysm, k).state_dict().items()} for k in sysm.CKPT_KEYS}
    a_cfg, t_cfg = ge.oracle_cfgs(wc, xc)
    g = torch.Generator().manual_seed(1)
    wave, ids = 0.1 * torch.randn(2, 4000, generator=g), torch.randint(4, 1000, (2, 8), generator=g)
    mask, labels = torch.ones(2, 8), torch.tensor([1, 2])

    def losses(seed):
        tr = OracleTrainer(sds, a_cfg, t_cfg, num_layers=3, heads=2, num_labels=4, dropout_seed=seed)
        return [tr.step(list(wave), ids, mask, labels)[0] for _ in range(2)]

    off, a, b, c = losses(None), losses(7), losses(7), losses(8)
    assert a == b and a != c and a != off
    assert all(np.isfinite(x) for x in off + a + c)


def _check_grads(p, gr):
    """Per tensor: max-abs-err <= 2e-4 * max|grad of the tensor| + 2e-6 * max|grad| over the model (the floor covers
    gradients that are zero in exact arithmetic, e.g. the key bias of a softmax attention)."""
    assert gr, "fixture holds no gradients"
    gmax = max(float(g.abs().max()) for g in gr.values())
    for k, g in gr.items():
        assert p[k].grad is not None, f"no gradient reached {k}"
        err = float((p[k].grad - g).abs().max())
        assert err <= 2e-4 * float(g.abs().max()) + 2e-6 * gmax, f"{k}: max-abs-err {err:.2e} (max|grad| {float(g.abs().max()):.2e})"


def test_audio_encoder_backward_unfrozen():
    """BASELINE config 3 (`freeze_base=False`, reference audio_encoder.py:9-17): autograd through the oracle's Wav2Vec2
    restatement reproduces the reference's gradients for every encoder and adapter parameter."""
    sd, _, r = split_fixture(load_npz("audio_encoder.npz"))
    _, gr, rg = split_fixture(load_npz("audio_encoder_grads.npz"))
    p = _leafs(sd)
    seq, _ = O.audio_encoder_forward(p, [t(r["wave0"]), t(r["wave1"])], cfg_of(r))
    (seq * t(rg["g_out"])).sum().backward()
    _check_grads(p, gr)


def test_text_encoder_backward_unfrozen():
    """BASELINE config 3 (reference text_encoder.py:8-15): same for XLM-R, embeddings included."""
    sd, _, r = split_fixture(load_npz("text_encoder.npz"))
    _, gr, rg = split_fixture(load_npz("text_encoder_grads.npz"))
    p = _leafs(sd)
    seq, _ = O.text_encoder_forward(p, t(r["input_ids"]), t(r["attention_mask"]), cfg_of(r))
    (seq * t(rg["g_out"])).sum().backward()
    _check_grads(p, gr)


def test_cpu_train_step_with_unfrozen_encoders_updates_them():
    """oracle/cpu_step.py with train_encoders=True (BASELINE config 3): encoder parameters move, by their group's
    learning rate (0.1 x), and stay put when frozen."""
    import __graft_entry__ as ge
    from oracle.cpu_step import OracleTrainer
    sysm, wc, xc = ge._small_system("cpu")
    sds = {k: {n: v.detach() for n, v in getattr(sysm, k).state_dict().items()} for k in sysm.CKPT_KEYS}
    a_cfg, t_cfg = ge.oracle_cfgs(wc, xc)
    g = torch.Generator().manual_seed(2)
    wave, ids = 0.1 * torch.randn(2, 4000, generator=g), torch.randint(4, 1000, (2, 8), generator=g)
    mask, labels = torch.ones(2, 8), torch.tensor([0, 3])
    key = "encoder.encoder.layers.0.attention.q_proj.weight"
    moved = {}
    for flag in (False, True):
        tr = OracleTrainer(sds, a_cfg, t_cfg, num_layers=3, heads=2, num_labels=4, lr=1e-3, train_encoders=flag)
        before = tr.sds["audio_encoder"][key].clone()
        loss, _ = tr.step(list(wave), ids, mask, labels)
        assert np.isfinite(loss)
        moved[flag] = float((tr.sds["audio_encoder"][key].detach() - before).abs().max())
    assert moved[False] == 0.0
    assert 0.0 < moved[True] <= 1.01e-4 + 1e-3 * 0.025 * 1e-4 * 10       # one AdamW step: |delta| <= lr x 0.1 (+ decay)
