"""BASELINE config 3 on the GPU: the fine-tuning form of the encoders (reference `freeze_base=False`,
src/models/audio_encoder.py:15-17, src/models/text_encoder.py:13-15).

* every Wav2Vec2 / XLM-R / adapter parameter gradient against the golden gradients captured from the reference
  (tests/golden/{audio,text}_encoder_grads.npz: gradient of sum(out * g) on the recorded inputs), forward against the
  recorded outputs;
* one whole training step of the assembled system with unfrozen encoders against the CPU oracle with autograd
  (logits, loss, sampled gradients everywhere from the conv front end to the classifier) and an AdamW update that moves the
  encoder weights at their group's learning rate."""
import json

import numpy as np
import pytest
import torch

from oracle import ser_oracle as O
from tests.helpers import cfg_of, load_npz, split_fixture, t

pytestmark = pytest.mark.gpu


def _check(named, gr, tol=3e-4):
    assert gr
    gmax = max(float(g.abs().max()) for g in gr.values())
    seen = 0
    for k, g in gr.items():
        if k not in named:
            continue
        p = named[k]
        if p.grad is None:
            assert float(g.abs().max()) == 0.0, f"no gradient reached {k}"
            continue
        err = float((p.grad.cpu() - g).abs().max())
        assert err <= tol * float(g.abs().max()) + 3e-6 * gmax, f"{k}: max-abs-err {err:.2e} (max|grad| {float(g.abs().max()):.2e})"
        seen += 1
    assert seen > 20


def test_audio_encoder_gradients_match_the_reference():
    import ser_amd  # noqa: F401
    from transformers import Wav2Vec2Config
    from ser_amd.models import AudioEncoder
    sd, _, r = split_fixture(load_npz("audio_encoder.npz"))
    _, gr, rg = split_fixture(load_npz("audio_encoder_grads.npz"))
    c = cfg_of(r)
    wc = Wav2Vec2Config(hidden_size=c["hidden"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"], intermediate_size=c["ffn"],
                        conv_dim=list(c["conv_dim"]), conv_kernel=list(c["conv_kernel"]), conv_stride=list(c["conv_stride"]),
                        num_conv_pos_embeddings=c["pos_kernel"], num_conv_pos_embedding_groups=c["pos_groups"], layer_norm_eps=c["eps"])
    ae = AudioEncoder(hf_config=wc, adapter_dim=sd["adapter.0.weight"].shape[0], freeze_base=False, use_quality_gates=False,
                      use_audio_conditioning=False)
    ae.load_state_dict(sd, strict=True)
    ae = ae.cuda().eval()
    assert all(p.requires_grad for p in ae.encoder.parameters())
    waves = [t(r["wave0"]).cuda(), t(r["wave1"]).cuda()]
    seq, mask = ae(waves, ["x", "y"])
    want = t(r["a_seq"])
    assert seq.shape == want.shape
    assert (seq.detach().cpu() - want).abs().max().item() < 2e-4
    (seq * t(rg["g_out"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    _check(dict(ae.named_parameters()), gr)
    assert ae.encoder.masked_spec_embed.grad is None or float(ae.encoder.masked_spec_embed.grad.abs().max()) == 0.0


def test_text_encoder_gradients_match_the_reference():
    import ser_amd  # noqa: F401
    from transformers import XLMRobertaConfig
    from ser_amd.models import TextEncoder
    sd, _, r = split_fixture(load_npz("text_encoder.npz"))
    _, gr, rg = split_fixture(load_npz("text_encoder_grads.npz"))
    c = cfg_of(r)
    xc = XLMRobertaConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                          intermediate_size=c["ffn"], max_position_embeddings=c["max_pos"], type_vocab_size=1, layer_norm_eps=c["eps"],
                          pad_token_id=c["pad_id"], bos_token_id=0, eos_token_id=2)
    te = TextEncoder(hf_config=xc, adapter_dim=sd["adapter.0.weight"].shape[0], freeze_base=False)
    te.load_state_dict(sd, strict=True)
    te = te.cuda().eval()
    seq, mask = te.forward_ids(t(r["input_ids"]).cuda(), t(r["attention_mask"]).cuda())
    assert (seq.detach().cpu() - t(r["t_seq"])).abs().max().item() < 2e-4
    (seq * t(rg["g_out"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    named = dict(te.named_parameters())
    _check(named, gr)
    # nn.Embedding(padding_idx): the pad rows of the word and position tables get no gradient
    pad = c["pad_id"]
    assert float(named["encoder.embeddings.word_embeddings.weight"].grad[pad].abs().max()) == 0.0
    assert float(named["encoder.embeddings.position_embeddings.weight"].grad[pad].abs().max()) == 0.0


@pytest.mark.parametrize("width", ["small", "base"])
def test_full_fine_tune_step_matches_the_oracle(width):
    """width "base": the Base-width layer shapes of BASELINE config 3 (768-d, 12 heads, FFN 3072, the 512-channel conv front end,
    128-tap 16-group positional conv) at 2 layers per encoder, so the oracle's autograd step stays a matter of seconds."""
    import __graft_entry__ as ge
    from transformers import Wav2Vec2Config, XLMRobertaConfig
    import ser_amd  # noqa: F401
    from ser_amd.models import AudioEncoder, TextEncoder
    from ser_amd.system import SERSystem, TrainStepper
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if width == "base":
        wc = Wav2Vec2Config(num_hidden_layers=2)
        xc = XLMRobertaConfig(vocab_size=1000, num_hidden_layers=2, max_position_embeddings=66, type_vocab_size=1, layer_norm_eps=1e-5,
                              pad_token_id=1, bos_token_id=0, eos_token_id=2)
        ae = AudioEncoder(hf_config=wc, freeze_base=False, use_quality_gates=False, use_audio_conditioning=False)
        te = TextEncoder(hf_config=xc, freeze_base=False)
        sysm = SERSystem(ae, te, num_labels=4, num_layers=3).to(dev)
    else:
        wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, conv_dim=[64] * 7,
                            num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
        xc = XLMRobertaConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                              max_position_embeddings=66, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
        ae = AudioEncoder(hf_config=wc, adapter_dim=32, freeze_base=False, use_quality_gates=False, use_audio_conditioning=False)
        te = TextEncoder(hf_config=xc, adapter_dim=32, freeze_base=False)
        sysm = SERSystem(ae, te, num_labels=4, shared_dim=64, num_heads=2, proj_dim=64, num_layers=3, base_dim=64).to(dev)
    sysm.train()
    sysm.train_dropout = False
    g = torch.Generator().manual_seed(5)
    B, T, S = 4, 4000, 9
    wave = 0.1 * torch.randn(B, T, generator=g)
    ids = torch.randint(4, 1000, (B, S), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    mask = torch.ones(B, S)
    ids[1, S - 3:] = 1
    ids[1, S - 4] = 2
    mask[1, S - 3:] = 0
    labels = torch.randint(0, 4, (B,), generator=g)
    sds = {k: {n: v.detach().cpu().clone() for n, v in getattr(sysm, k).state_dict().items()} for k in sysm.CKPT_KEYS}
    a_cfg, t_cfg = ge.oracle_cfgs(wc, xc)
    leaf = {k: {n: v.clone().requires_grad_(v.dtype.is_floating_point) for n, v in sd.items()} for k, sd in sds.items()}
    out = O.full_forward(leaf, list(wave), ids, mask, a_cfg, t_cfg, num_layers=3, heads=8 if width == "base" else 2, use_openmax=False, training=True)
    ref_loss = O.train_loss(out["logits"], out["unc"], out["fused"], leaf["prototypes"]["prototypes"], labels, 4)
    ref_loss.backward()

    opt = sysm.make_optimizer(lr=1e-3)
    stepper = TrainStepper(sysm, opt, use_graph=False)
    opt.zero_grad(set_to_none=True)
    loss, logits = stepper._fwd_bwd(wave.to(dev), ids.to(dev), mask.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    assert (logits.cpu() - out["logits"].detach()).abs().max().item() < 1e-3
    assert torch.equal(logits.argmax(1).cpu(), out["logits"].argmax(1))
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    checked = 0
    for key in ("audio_encoder", "text_encoder", "cross", "classifier"):
        named = dict(getattr(sysm, key).named_parameters())
        for n, v in leaf[key].items():
            if v.grad is None or n not in named or named[n].grad is None:
                continue
            denom = max(v.grad.abs().max().item(), 1e-6)
            rel = (named[n].grad.cpu() - v.grad).abs().max().item() / denom
            assert rel < 2e-2 or (named[n].grad.cpu() - v.grad).abs().max().item() < 1e-7, f"{key}.{n}: gradient differs from the oracle (rel {rel:.3e})"
            checked += 1
    assert checked > 150
    w = sysm.audio_encoder.encoder.encoder.layers[0].attention.q_proj.weight
    c0 = sysm.audio_encoder.encoder.feature_extractor.conv_layers[0].conv.weight
    before, before_c0 = w.detach().clone(), c0.detach().clone()
    opt.step()
    torch.cuda.synchronize()
    moved = (w.detach() - before).abs().max().item()
    assert 0.0 < moved <= 1.01e-4 + 1e-6, f"encoder weights move by lr x 0.1 per AdamW step, got {moved}"
    assert (c0.detach() - before_c0).abs().max().item() > 0.0, "the conv front end is trained too"


def test_encoder_training_noise_matches_the_oracle_with_the_same_draws():
    """The encoders' own training-mode noise (reference: .train() on both encoders, src/train.py:124): HF's hidden /
    attention / activation dropout sites, LayerDrop and SpecAugment in the fine-tuning path, against the oracle applying
    the same masks (the build's generator, restated in the oracle), the same skipped layers and the same masked frames:
    outputs and every parameter gradient."""
    import ser_amd  # noqa: F401
    from transformers import Wav2Vec2Config, XLMRobertaConfig
    from ser_amd import _ops as OP
    from ser_amd.models import AudioEncoder, TextEncoder
    from ser_amd.models._finetune import Noise
    import __graft_entry__ as ge
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256, conv_dim=[64] * 7,
                        num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, layerdrop=0.3, mask_time_prob=0.3,
                        mask_time_length=2, mask_time_min_masks=2)
    xc = XLMRobertaConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                          max_position_embeddings=66, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    ae = AudioEncoder(hf_config=wc, adapter_dim=32, freeze_base=False, use_quality_gates=False, use_audio_conditioning=False).to(dev).train()
    te = TextEncoder(hf_config=xc, adapter_dim=32, freeze_base=False).to(dev).train()
    ae.encoder_train_noise = te.encoder_train_noise = True
    ae.noise_seed = 5                                  # a seed whose first draws skip a layer (asserted below)
    a_cfg, t_cfg = ge.oracle_cfgs(wc, xc)
    g = torch.Generator().manual_seed(3)
    B, T, S = 3, 4000, 9
    wave = 0.1 * torch.randn(B, T, generator=g)
    ids = torch.randint(4, 1000, (B, S), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    mask = torch.ones(B, S)
    ga, gt = torch.randn(B, 12, 128, generator=g), torch.randn(B, S, 128, generator=g)
    state = torch.full((1,), 777, dtype=torch.int64, device=dev)
    with OP.dropout_scope(state):
        a_seq = ae.encode(wave.to(dev))
        t_seq, _ = te.forward_ids(ids.to(dev), mask.to(dev))
    assert a_seq.shape == (B, 12, 128)
    ((a_seq * ga.to(dev)).sum() + (t_seq * gt.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    na, nt = ae._noise, te._noise
    assert na.spec_mask is not None and na.spec_mask.any() and (na.p_hidden, na.p_attn, na.p_act) == (0.1, 0.1, 0.1)

    sda = {n: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for n, v in ae.state_dict().items()}
    sdt = {n: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for n, v in te.state_dict().items()}
    noa = O.EncoderNoise(777, 0, na.p_hidden, na.p_attn, na.p_act, na.p_featproj, na.skip, na.spec_mask)
    not_ = O.EncoderNoise(777, 1, nt.p_hidden, nt.p_attn, 0.0, 0.0)
    x = torch.stack([O.normalise_waveform(w) for w in wave])
    ra = O.adapter(O.wav2vec2_forward(O.sub(sda, "encoder."), x, a_cfg, noa), O.sub(sda, "adapter."))
    rt = O.adapter(O.xlmr_forward(O.sub(sdt, "encoder."), ids, mask, t_cfg, not_), O.sub(sdt, "adapter."))
    ((ra * ga).sum() + (rt * gt).sum()).backward()
    clean = O.adapter(O.wav2vec2_forward(O.sub(sda, "encoder."), x, a_cfg), O.sub(sda, "adapter."))
    assert (clean - ra).abs().max().item() > 0.1, "the noise must change the output"
    assert (a_seq.detach().cpu() - ra.detach()).abs().max().item() < 5e-4
    assert (t_seq.detach().cpu() - rt.detach()).abs().max().item() < 5e-4
    for mod, sd in ((ae, sda), (te, sdt)):
        named = dict(mod.named_parameters())
        gmax = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
        n_checked = 0
        for n, v in sd.items():
            if v.grad is None or n not in named:
                continue
            got = named[n].grad
            if got is None:
                assert float(v.grad.abs().max()) == 0.0, n
                continue
            err = float((got.cpu() - v.grad).abs().max())
            assert err <= 2e-3 * float(v.grad.abs().max()) + 1e-5 * gmax, f"{n}: {err:.2e} vs max {float(v.grad.abs().max()):.2e}"
            n_checked += 1
        assert n_checked > 30
    if na.skip:                                         # a skipped layer's parameters get no gradient
        l = sorted(na.skip)[0]
        gq = ae.encoder.encoder.layers[l].attention.q_proj.weight.grad
        assert gq is None or float(gq.abs().max()) == 0.0
    assert float(ae.encoder.masked_spec_embed.grad.abs().max()) > 0.0, "SpecAugment frames feed gradients to masked_spec_embed"


def test_frozen_encoders_with_training_noise_match_the_oracle():
    """freeze_base=True + encoder_train_noise: what the reference's train.py does to its frozen encoders (.train() on
    both, src/train.py:124) - HF dropout sites, LayerDrop and SpecAugment on, no encoder gradients.  Outputs against the
    oracle with the same draws; the encoder parameters stay without gradients and the adapters train."""
    import ser_amd  # noqa: F401
    from transformers import Wav2Vec2Config, XLMRobertaConfig
    from ser_amd import _ops as OP
    from ser_amd.models import AudioEncoder, TextEncoder
    import __graft_entry__ as ge
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256, conv_dim=[64] * 7,
                        num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, layerdrop=0.3, mask_time_prob=0.3,
                        mask_time_length=2, mask_time_min_masks=2)
    xc = XLMRobertaConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                          max_position_embeddings=66, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    ae = AudioEncoder(hf_config=wc, adapter_dim=32, use_quality_gates=False, use_audio_conditioning=False).to(dev).train()
    te = TextEncoder(hf_config=xc, adapter_dim=32).to(dev).train()
    assert ae.freeze_base and te.freeze_base
    ae.encoder_train_noise = te.encoder_train_noise = True
    ae.noise_seed = 5
    a_cfg, t_cfg = ge.oracle_cfgs(wc, xc)
    g = torch.Generator().manual_seed(3)
    B, T, S = 3, 4000, 9
    wave = 0.1 * torch.randn(B, T, generator=g)
    ids = torch.randint(4, 1000, (B, S), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    mask = torch.ones(B, S)
    state = torch.full((1,), 777, dtype=torch.int64, device=dev)
    with OP.dropout_scope(state):
        a_seq = ae.encode(wave.to(dev))
        t_seq, _ = te.forward_ids(ids.to(dev), mask.to(dev))
    (a_seq.sum() + t_seq.sum()).backward()
    na, nt = ae._noise, te._noise
    assert na.spec_mask is not None and na.spec_mask.any()
    assert all(p.grad is None for p in ae.encoder.parameters()) and all(p.grad is None for p in te.encoder.parameters())
    assert ae.adapter[0].weight.grad is not None and te.adapter[0].weight.grad is not None
    sda = {n: v.detach().cpu().clone() for n, v in ae.state_dict().items()}
    sdt = {n: v.detach().cpu().clone() for n, v in te.state_dict().items()}
    noa = O.EncoderNoise(777, 0, na.p_hidden, na.p_attn, na.p_act, na.p_featproj, na.skip, na.spec_mask)
    not_ = O.EncoderNoise(777, 1, nt.p_hidden, nt.p_attn, 0.0, 0.0)
    x = torch.stack([O.normalise_waveform(w) for w in wave])
    ra = O.adapter(O.wav2vec2_forward(O.sub(sda, "encoder."), x, a_cfg, noa), O.sub(sda, "adapter."))
    rt = O.adapter(O.xlmr_forward(O.sub(sdt, "encoder."), ids, mask, t_cfg, not_), O.sub(sdt, "adapter."))
    clean = O.adapter(O.wav2vec2_forward(O.sub(sda, "encoder."), x, a_cfg), O.sub(sda, "adapter."))
    assert (clean - ra).abs().max().item() > 0.1, "the noise must change the output"
    assert (a_seq.detach().cpu() - ra).abs().max().item() < 5e-4
    assert (t_seq.detach().cpu() - rt).abs().max().item() < 5e-4
    # eval mode: back on the bf16 engine, no noise
    ae.eval()
    with torch.no_grad():
        e = ae.encode(wave.to(dev))
    assert (e.cpu() - clean).abs().max().item() < 5e-4


def test_captured_fine_tune_steps_equal_eager_steps_with_the_same_draws():
    """BASELINE config 3 as hipGraphs: LayerDrop and SpecAugment are host decisions (hf modeling_wav2vec2.py:700-703, :1293-1302),
    so the captured step computes every layer, discards a dropped one by a select on a device word and gates its parameters'
    AdamW update with the same word (torch.optim.AdamW leaves a parameter without a gradient untouched).  Against eager stepping
    with the same seeds - which really skips the layer and really has `grad is None` - losses and every parameter must agree
    bit for bit over steps that include dropped layers, and the capture's warm-up passes must not use up draws."""
    import ser_amd  # noqa: F401
    from transformers import Wav2Vec2Config, XLMRobertaConfig
    from ser_amd.models import AudioEncoder, TextEncoder
    from ser_amd.system import SERSystem, TrainStepper
    dev = torch.device("cuda:0")

    def build():
        torch.manual_seed(0)
        wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256, conv_dim=[64] * 7,
                            num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, layerdrop=0.3, mask_time_prob=0.3,
                            mask_time_length=2, mask_time_min_masks=2)
        xc = XLMRobertaConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                              max_position_embeddings=66, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
        ae = AudioEncoder(hf_config=wc, adapter_dim=32, freeze_base=False, use_quality_gates=False, use_audio_conditioning=False)
        te = TextEncoder(hf_config=xc, adapter_dim=32, freeze_base=False)
        sysm = SERSystem(ae, te, num_labels=4, shared_dim=64, num_heads=2, proj_dim=64, num_layers=3, base_dim=64).to(dev)
        sysm.train()
        for m in (sysm.audio_encoder, sysm.text_encoder):
            m.encoder_train_noise, m.noise_seed = True, 5
        return sysm

    g = torch.Generator().manual_seed(11)
    B, S = 3, 9
    batches = []
    for T in (4000, 4000, 4800, 4000, 4800, 4000):     # two input shapes: the second graph set accumulates into the first one's gradient tensors
        ids = torch.randint(4, 1000, (B, S), generator=g)
        ids[:, 0], ids[:, -1] = 0, 2
        batches.append([0.1 * torch.randn(B, T, generator=g).to(dev), ids.to(dev), torch.ones(B, S).to(dev),
                        torch.randint(0, 4, (B,), generator=g).to(dev)])
    runs = {}
    for mode in ("eager", "graph"):
        sysm = build()
        opt = sysm.make_optimizer(lr=1e-3)
        st = TrainStepper(sysm, opt, use_graph=(mode == "graph"))
        losses, skipped = [], []
        for b in batches:
            losses.append(st.step(*b).clone())
            skipped.append(sorted(sysm.audio_encoder._noise.skip))
        torch.cuda.synchronize()
        runs[mode] = (torch.stack([l.reshape(()) for l in losses]).cpu(), skipped,
                      {n: p.detach().cpu().clone() for n, p in sysm.named_parameters()}, opt)
    assert runs["eager"][1] == runs["graph"][1], "the capture must not use up LayerDrop draws"
    assert any(runs["eager"][1]) and not all(runs["eager"][1]), f"the steps must include dropped and kept layers: {runs['eager'][1]}"
    assert torch.equal(runs["eager"][0], runs["graph"][0]), f"losses differ: {runs['eager'][0]} vs {runs['graph'][0]}"
    for n, v in runs["eager"][2].items():
        assert torch.equal(v, runs["graph"][2][n]), f"{n}: captured and eager stepping diverged ({(v - runs['graph'][2][n]).abs().max():.3e})"
    # a layer dropped in the first step keeps its initial weights through that step in both modes: covered by the equality above,
    # and the optimizer's moments of such a parameter stay untouched too
    oe, og = runs["eager"][3], runs["graph"][3]
    assert oe.t == og.t == len(batches)


@pytest.mark.parametrize("S", [12, 199, 230])
def test_positional_conv_node_on_the_resident_slab_kernel(S):
    """`_PosConvDirect` (Base geometry: 128 taps, 16 groups of 48): GELU(conv(z) + b) + z and all three gradients against float64
    autograd of torch's grouped conv1d with padding 64 and the last frame dropped (hf modeling_wav2vec2.py:326-368)."""
    import ser_amd  # noqa: F401
    from ser_amd.models._finetune import _PosConvDirect
    dev = torch.device("cuda:0")
    B, H, G, K = 2, 768, 16, 128
    g = torch.Generator().manual_seed(S)
    z = torch.randn(B * S, H, generator=g)
    Wp = torch.randn(H, H // G, K, generator=g) / (K * H // G) ** 0.5          # pre-activations of unit scale
    bias = 0.1 * torch.randn(H, generator=g)
    dout = torch.randn(B * S, H, generator=g)
    zd, wd, bd = (t_.double().requires_grad_() for t_ in (z, Wp, bias))
    pre = torch.nn.functional.conv1d(zd.view(B, S, H).transpose(1, 2), wd, bd, padding=K // 2, groups=G)[:, :, :S].transpose(1, 2)
    ref = (torch.nn.functional.gelu(pre) + zd.view(B, S, H)).reshape(B * S, H)
    ref.backward(dout.double())
    zg, wg, bg = (t_.to(dev).requires_grad_() for t_ in (z, Wp, bias))
    out = _PosConvDirect.apply(zg, wg, bg, B, S, K, G)
    out.backward(dout.to(dev))
    torch.cuda.synchronize()
    assert (out.detach().cpu() - ref.detach().float()).abs().max().item() < 5e-5       # three bf16 products per multiply, K Cg = 6144 terms
    for got, want, name in ((zg.grad, zd.grad, "dz"), (wg.grad, wd.grad, "dW"), (bg.grad, bd.grad, "db")):
        err = (got.cpu() - want.float()).abs().max().item()
        assert err < 2e-4 * max(1.0, want.abs().max().item()), f"{name}: {err:.3e} (max {want.abs().max().item():.3e})"


@pytest.mark.parametrize("M,N", [(1592, 192), (3300, 192), (1592, 2304)])
def test_tile_path_linear_and_conv_gradients_with_split_k(M, N):
    """`_Linear` and `_ConvPad` on the MFMA tile kernel at row counts where the weight gradient's K range (the rows) is cut into
    4 / 8 slices (`ser_gemm_bf16_nt_splitk` + slice-order sum), and at a layer width (2 304 outputs) where the input gradient's K
    range is cut too: outputs and all gradients against float64."""
    import ser_amd  # noqa: F401
    from ser_amd.models._finetune import SLACK, _ConvPad, _Linear
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + N)
    K = 256
    x, W, b, dy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    xg, Wg, bg = (t_.to(dev).requires_grad_() for t_ in (x, W, b))
    y = _Linear.apply(xg, Wg, bg, None)
    y.backward(dy.to(dev))
    xd, Wd, bd = (t_.double().requires_grad_() for t_ in (x, W, b))
    (xd @ Wd.t() + bd).backward(dy.double())
    assert (y.detach().cpu() - (x.double() @ W.double().t() + b.double()).float()).abs().max().item() < 5e-5      # three bf16 products per multiply
    for got, want, name in ((xg.grad, xd.grad, "dx"), (Wg.grad, Wd.grad, "dW"), (bg.grad, bd.grad, "db")):
        assert (got.cpu() - want.float()).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item()), name
    # a long-K forward (the FFN's second Linear, K = 3072): split K + bias in the slice sum
    K2 = 3072
    x2, W2l, b2 = torch.randn(M, K2, generator=g), torch.randn(192, K2, generator=g) / K2 ** 0.5, torch.randn(192, generator=g)
    y2 = _Linear.apply(x2.to(dev).requires_grad_(), W2l.to(dev).requires_grad_(), b2.to(dev).requires_grad_(), None)
    assert (y2.detach().cpu() - (x2.double() @ W2l.double().t() + b2.double()).float()).abs().max().item() < 5e-5
    # conv: C_in = C_out = 64, kernel 3, stride 2 on one padded buffer
    Cin, Cout, k, s_ = 64, 64, 3, 2
    rows_out = M
    xin = torch.randn(s_ * rows_out + SLACK, Cin, generator=g)
    W2 = torch.randn(Cout, k * Cin, generator=g) / (k * Cin) ** 0.5
    dyc = torch.randn(rows_out + SLACK, Cout, generator=g)
    dyc[rows_out:] = 0
    xg2, Wg2 = xin.to(dev).requires_grad_(), W2.to(dev).requires_grad_()
    yc = _ConvPad.apply(xg2, Wg2, k, s_, rows_out)
    yc.backward(dyc.to(dev))
    xd2, Wd2 = xin.double().requires_grad_(), W2.double().requires_grad_()
    win = torch.stack([xd2[s_ * m:s_ * m + k].reshape(-1) for m in range(rows_out)])          # [rows_out, k Cin]
    ref = win @ Wd2.t()
    ref.backward(dyc[:rows_out].double())
    assert (yc[:rows_out].detach().cpu() - ref.detach().float()).abs().max().item() < 5e-5
    assert (Wg2.grad.cpu() - Wd2.grad.float()).abs().max().item() < 1e-4 * max(1.0, Wd2.grad.abs().max().item())
    assert (xg2.grad.cpu() - xd2.grad.float()).abs().max().item() < 1e-4
