"""CPU-only tests of the host-side logic around the kernels: flat parameter buckets, optimizer
planning and schedule, state_dict key parity with the reference, CLI flags, data-feed helpers."""
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, load_npz

import ser_amd  # noqa: F401


def _manifest():
    return json.load(open(os.path.join(GOLDEN, "state_dict_manifest.json")))


def test_state_dict_keys_match_reference_full_size():
    from ser_amd.models import FusionLayer
    from ser_amd.models.classifier import AdvancedOpenMaxClassifier
    from ser_amd.models.cross_attention import CrossModalAttention
    from ser_amd.models.pooling import AttentiveStatsPooling
    from ser_amd.models.prototypes import PrototypeMemory
    man = _manifest()
    shapes = lambda m: {k: list(v.shape) for k, v in m.state_dict().items()}
    assert shapes(AdvancedOpenMaxClassifier(512, 4, 35, 512, 0.15)) == man["classifier"]
    assert shapes(CrossModalAttention(768, 768, 256, 8)) == man["cross"]
    assert shapes(FusionLayer(1536, 1536, 512)) == man["fusion"]
    assert shapes(AttentiveStatsPooling(768)) == man["pool"]
    assert shapes(PrototypeMemory(4, 512)) == man["prototypes"]


def test_encoder_wrappers_keep_reference_keys():
    from transformers import Wav2Vec2Config, XLMRobertaConfig
    from ser_amd.models import AudioEncoder, TextEncoder
    man = _manifest()
    wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                        conv_dim=[64] * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
    ae = AudioEncoder(hf_config=wc, adapter_dim=32, use_quality_gates=False, use_audio_conditioning=False)
    assert {k: list(v.shape) for k, v in ae.state_dict().items()} == man["audio_encoder_small"]
    xc = XLMRobertaConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                          intermediate_size=256, max_position_embeddings=66, type_vocab_size=1, pad_token_id=1)
    te = TextEncoder(hf_config=xc, adapter_dim=32)
    assert {k: list(v.shape) for k, v in te.state_dict().items()} == man["text_encoder_small"]
    # with the gate flags on, the reference's extra learnable keys exist too
    ae2 = AudioEncoder(hf_config=wc, adapter_dim=32)
    keys = set(ae2.state_dict())
    for k in ("quality_gates.quality_projection.0.weight", "quality_gates.quality_projection.3.bias", "quality_fusion.0.weight",
              "audio_conditioning.conditioning_projection.0.weight", "conditioning_fusion.0.weight", "combined_fusion.0.weight"):
        assert k in keys
    assert ae2.state_dict()["combined_fusion.0.weight"].shape == (128, 148)
    from ser_amd.models.frontend import AudioConditioningModule, FrontEndQualityGates
    assert isinstance(ae2.quality_gates, FrontEndQualityGates) and isinstance(ae2.audio_conditioning, AudioConditioningModule)
    from ser_amd._lib import SerHipError
    with pytest.raises((AssertionError, SerHipError)):     # the front end is device kernels: no device / CPU tensors -> refused, no fallback
        ae2([torch.zeros(4000)])
    with pytest.raises(ValueError, match="VAD method"):
        AudioEncoder(hf_config=wc, adapter_dim=32, vad_method="silero")


def test_language_features_follow_the_reference_table():
    """ref quality_gates.py:252-301, :514-517 — pinned by tests/test_oracle_dsp.py through the same numbers."""
    import numpy as np
    from ser_amd.models.frontend import LID_LISTED, LID_OTHER, language_features
    G = np.load(os.path.join(GOLDEN, "dsp_frontend.npz"))
    rows = G["qg.lid"]                     # (langdetect available?, entropy, confidence) for: unavailable, en, ja, hi
    lf = language_features(["some words"] * 3 + [None, "   ", "x"], 6, detector=lambda t: {"some words": "en"}.get(t, "hi"))
    assert lf[0].tolist() == pytest.approx([rows[1][1], rows[1][2]], abs=1e-6)
    assert list(LID_LISTED) == pytest.approx([rows[1][1], rows[1][2]], abs=1e-9)
    assert list(LID_OTHER) == pytest.approx([rows[3][1], rows[3][2]], abs=1e-9)
    assert lf[3].tolist() == [1.0, 0.0] and lf[4].tolist() == [1.5, 0.0] and lf[5].tolist() == pytest.approx(list(LID_OTHER), abs=1e-6)
    none = language_features(["some words"], 1, detector=None)
    from ser_amd.models import frontend as FE
    if FE._langdetect is None:
        assert none[0].tolist() == pytest.approx([rows[0][1], rows[0][2]])
    boom = language_features(["x"], 1, detector=lambda t: 1 / 0)
    assert boom[0].tolist() == [1.5, 0.0]


def test_flat_params_views_roundtrip_and_reflatten():
    from ser_amd.models import FusionLayer
    m = FusionLayer(32, 32, 16)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = m._flat.ensure()
    assert fp.flat.numel() == fp.total and fp.total % 64 == 0
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    for p, o in zip(fp.params, fp.offsets):
        assert p.data_ptr() == fp.flat.data_ptr() + 4 * o and o % 64 == 0
    # load_state_dict writes through the views
    sd = {k: torch.randn_like(v) for k, v in before.items()}
    m.load_state_dict(sd)
    assert torch.equal(fp.flat[fp.offsets[0]:fp.offsets[0] + fp.params[0].numel()].view_as(fp.params[0]), sd["proj_a.0.weight"])
    # a dtype/device move re-points the parameters: ensure() must notice and re-flatten
    m.double().float()
    fp2 = m._flat.ensure()
    assert fp2.params[0].data_ptr() == fp2.flat.data_ptr()
    assert not fp.accumulating()
    fp.publish()
    assert fp.accumulating()


def test_optimizer_plan_follows_reference_groups():
    from transformers import Wav2Vec2Config, XLMRobertaConfig
    from ser_amd.models import AudioEncoder, TextEncoder
    from ser_amd.system import SERSystem
    wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128,
                        conv_dim=[64] * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
    xc = XLMRobertaConfig(vocab_size=100, hidden_size=128, num_hidden_layers=1, num_attention_heads=2,
                          intermediate_size=128, max_position_embeddings=40, type_vocab_size=1, pad_token_id=1)
    s = SERSystem(AudioEncoder(hf_config=wc, adapter_dim=16, use_quality_gates=False, use_audio_conditioning=False),
                  TextEncoder(hf_config=xc, adapter_dim=16), num_labels=4, shared_dim=64, num_heads=2, proj_dim=64,
                  num_layers=2, base_dim=64)
    from ser_amd.models.adapter import adapter_apply  # buckets for the adapters are created lazily
    from ser_amd.models._flat import FlatParams
    for enc in (s.audio_encoder, s.text_encoder):
        enc._adapter_flat = FlatParams(list(enc.adapter.parameters()))
    opt = s.make_optimizer(lr=1e-3)
    opt._build_plan()
    mults = [round(g["lr_mult"], 6) for g, _, _ in opt._plan]
    wds = [g["weight_decay"] for g, _, _ in opt._plan]
    assert mults == [0.1, 0.1, 1, 1, 1, 1, 1.5, 2.0, 1.0, 1]            # ref train.py:72-83
    assert wds == [.025, .025, .05, .05, .05, .05, .06, .04, .05, .05]
    # the classifier's three groups are three contiguous, disjoint segments of ONE flat bucket
    segs = [opt._plan[i][1] for i in (6, 7, 8)]
    assert all(len(x) == 1 for x in segs) and len({id(x[0][0]) for x in segs}) == 1
    (b, s0, e0), (_, s1, e1), (_, s2, e2) = segs[0][0], segs[1][0], segs[2][0]
    assert s0 == 0 and e0 == s1 and e1 == s2 and e2 == b.total
    # frozen encoder weights are not planned; the adapter bucket is; the encoder's unused `pool` head is planned
    # but, exactly like torch (grad is None -> skipped), never updated because it never receives a gradient
    audio_segs = opt._plan[0][1]
    assert audio_segs[0][0] is s.audio_encoder._adapter_flat
    assert [sg[0] for sg in audio_segs[1:]] == [s.audio_encoder.pool._flat]
    assert all(p.grad is None for p in s.audio_encoder.pool.parameters())
    assert opt._plan[9][2] == [s.prototypes.prototypes]


def test_warmup_cosine_matches_reference_lambda():
    from ser_amd.optim import WarmupCosine

    class _Opt:
        lr_factor = 1.0
    r = load_npz("adamw.npz")
    o = _Opt()
    sch = WarmupCosine(o, int(r["total_steps"]), float(r["warmup_ratio"]))
    got = [o.lr_factor]
    for _ in range(int(r["total_steps"])):
        sch.step()
        got.append(o.lr_factor)
    np.testing.assert_allclose(got, r["lambda_values"], atol=1e-7)
    assert got[0] == 0.0     # like torch's LambdaLR, the first optimizer step runs with lr * lambda(0) = 0


def test_cli_flags_match_reference():
    from ser_amd.train import build_parser
    a = build_parser().parse_args([])
    ref_defaults = dict(train_manifest='train_70.jsonl', val_manifest='val_20.jsonl', epochs=5, batch_size=4, lr=1e-4,
                        warmup_ratio=0.1, use_amp=False, augment=False, proto_weight=0.05, save_dir='checkpoints',
                        resume_from=None)
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, k


def test_checkpoint_layout_keys():
    from ser_amd.system import SERSystem
    assert SERSystem.CKPT_KEYS == ("audio_encoder", "text_encoder", "cross", "pool_a", "pool_t", "fusion", "classifier",
                                   "prototypes")     # ref train.py:249-262 (+ optimizer, scheduler, epoch, f1)


def test_speed_perturb_is_length_preserving_lowpass():
    from ser_amd.data.preprocess import add_noise_snr, resample, speed_perturb
    t = torch.arange(16000) / 16000.0
    tone = 0.5 * torch.sin(2 * np.pi * 440 * t)
    for f in (0.9, 0.95, 1.05, 1.1):
        y = speed_perturb(tone, f)
        assert y.shape == tone.shape                       # ref preprocess.py:56-61: duration does not change
        assert (y[200:-200] - tone[200:-200]).abs().max() < 2e-2   # a 440 Hz tone survives the round trip
    assert speed_perturb(tone, 1.0005) is tone
    assert resample(tone[None], 16000, 8000).shape == (1, 8000)
    n = add_noise_snr(tone, 10.0)
    snr = 10 * torch.log10(tone.pow(2).mean() / (n - tone).pow(2).mean())
    assert abs(float(snr) - 10.0) < 0.5 and float(n.abs().max()) <= 1.0


def test_synthetic_dataset_shapes():
    from ser_amd.data.dataset import SyntheticSERDataset
    ds = SyntheticSERDataset(5, seconds=1.0, num_labels=4)
    w, txt, y = ds[3]
    assert w.shape == (16000,) and len(txt.split()) == 30 and 0 <= y < 4


def test_bench_contract_pieces_without_a_gpu():
    """bench.py: defaults finish in minutes at N=1, the synthetic batch has BASELINE config 2's shape, and without a GPU
    the script refuses to run (there is no CPU fallback of the product path)."""
    import subprocess
    import sys
    import bench
    wave, ids, mask, labels = bench.synth_batch(16, 4.0, 32, 250002, 4, seed=1)
    assert wave.shape == (16, 64000) and ids.shape == (16, 32) and mask.shape == (16, 32) and labels.shape == (16,)
    assert int(ids[:, 0].max()) == 0 and int(ids[:, -1].min()) == 2 and int(labels.max()) < 4
    wc, xc = bench.hf_configs()
    assert (wc.hidden_size, wc.num_hidden_layers, xc.hidden_size, xc.num_hidden_layers) == (768, 12, 768, 12)
    if not torch.cuda.is_available():
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "no CPU fallback" in r.stderr and not r.stdout.strip()


def test_reference_checkpoint_round_trip(tmp_path):
    """A checkpoint written by the REFERENCE's modules + torch AdamW / LambdaLR (tests/golden/ref_checkpoint_small.pt, made by
    make_checkpoint_fixture.py; it also carries the gate parameters of the reference's default AudioEncoder()) loads into
    this build: module weights, AdamW moments by parameter name despite the shifted indices, step count, scheduler; and
    the optimizer state exported back in torch's format loads into a torch.optim.AdamW over the same grouping."""
    import os
    import torch
    from transformers import Wav2Vec2Config, XLMRobertaConfig
    import ser_amd  # noqa: F401
    from ser_amd.models import AudioEncoder, TextEncoder
    from ser_amd.optim import WarmupCosine
    from ser_amd.system import SERSystem
    ck = torch.load(os.path.join(os.path.dirname(__file__), "golden", "ref_checkpoint_small.pt"), map_location="cpu", weights_only=False)
    wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, conv_dim=[64] * 7,
                        num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
    xc = XLMRobertaConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                          max_position_embeddings=66, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    ae = AudioEncoder(hf_config=wc, adapter_dim=32, use_quality_gates=False, use_audio_conditioning=False)
    te = TextEncoder(hf_config=xc, adapter_dim=32)
    sysm = SERSystem(ae, te, num_labels=4, shared_dim=64, num_heads=2, proj_dim=64, num_layers=3, base_dim=64)
    msgs = []
    sysm.load_checkpoint_dict(ck, log=msgs.append)
    assert len(msgs) == 1 and "skipping 14 checkpoint entries" in msgs[0]
    for key in sysm.CKPT_KEYS:                                   # every entry this build holds equals the checkpoint's
        own = getattr(sysm, key).state_dict()
        for n, v in own.items():
            assert torch.equal(v, ck[key][n]), f"{key}.{n}"
    # gates-on construction takes the same checkpoint strictly
    ae2 = AudioEncoder(hf_config=wc, adapter_dim=32)
    ae2.load_state_dict(ck["audio_encoder"], strict=True)

    opt = sysm.make_optimizer(lr=1e-3)
    sched = WarmupCosine(opt, 10, 0.0)
    by_name = {(k, n): p for k in sysm.CKPT_KEYS for n, p in getattr(sysm, k).named_parameters()}
    order = sysm.torch_param_order(ck)
    opt.load_state_dict(ck["optimizer"], order=order, params_by_name=by_name)
    sched.load_state_dict(ck["scheduler"])
    assert opt.t == 2 and sched.last_epoch == 2 and abs(opt.base_lr - 1e-3) < 1e-12
    assert abs(opt.lr_factor - sched.factor(2)) < 1e-6
    for name, want in ck["_probe"].items():
        key, n = name.split(".", 1)
        m, v = opt._moments_of(by_name[(key, n)])
        assert torch.allclose(m, want["exp_avg"]) and torch.allclose(v, want["exp_avg_sq"]), name
        assert float(m.abs().max()) > 0
    # export in torch's format and load it into a torch AdamW built over this system with the reference's grouping
    own_order = sysm.torch_param_order()
    exported = opt.torch_state_dict(own_order, by_name)
    c = sysm.classifier
    topt = torch.optim.AdamW([{'params': sysm.audio_encoder.parameters()}, {'params': sysm.text_encoder.parameters()},
                              {'params': sysm.cross.parameters()}, {'params': sysm.pool_a.parameters()},
                              {'params': sysm.pool_t.parameters()}, {'params': sysm.fusion.parameters()},
                              {'params': c.deep_classifier.parameters()}, {'params': c.anchor_clustering.parameters()},
                              {'params': c.uncertainty_head.parameters()}, {'params': sysm.prototypes.parameters()}], lr=1e-3)
    topt.load_state_dict(exported)
    q = sysm.cross.q_a.weight
    assert torch.allclose(topt.state[q]["exp_avg"], ck["_probe"]["cross.q_a.weight"]["exp_avg"])
    assert float(topt.state[q]["step"]) == 2.0


def test_encoder_noise_plan_is_the_same_whether_the_forward_or_the_graph_owner_draws_it():
    """models/_finetune.py Noise: LayerDrop / SpecAugment decisions are host draws.  The eager forward calls plan() itself; a captured
    step has its owner call plan() + stage() before each replay.  Same generator, same order -> same decisions; a saved / restored
    generator state replays them (what the capture's warm-up passes rely on); the frame count used to plan ahead of the forward is the
    conv stack's (hf modeling_wav2vec2.py:1113-1130)."""
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    from ser_amd.models._finetune import Noise, wav2vec2_frames
    cfg = Wav2Vec2Config(hidden_size=64, num_hidden_layers=6, num_attention_heads=2, intermediate_size=128, conv_dim=[32] * 7,
                         num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, layerdrop=0.3, mask_time_prob=0.2,
                         mask_time_length=3, mask_time_min_masks=2)
    model = Wav2Vec2Model(cfg)
    for T in (4000, 16000, 64000, 12345):
        assert wav2vec2_frames(cfg, T) == int(model._get_feat_extract_output_lengths(T))
    a, b = Noise(cfg, 0, seed=9), Noise(cfg, 0, seed=9)
    seq_a, seq_b = [], []
    for step in range(6):
        S = wav2vec2_frames(cfg, 16000)
        a.plan(cfg.num_hidden_layers, 3, S)
        seq_a.append((sorted(a.skip), a.spec_mask.copy()))
        if step == 2:                                  # a capture in the middle of a run: draws made for warm-up are put back
            st = b.state()
            for _ in range(3):
                b.plan(cfg.num_hidden_layers, 3, S)
            b.set_state(st)
        b.plan(cfg.num_hidden_layers, 3, S)
        seq_b.append((sorted(b.skip), b.spec_mask.copy()))
    assert any(s for s, _ in seq_a), "the draws must drop some layer"
    for (sa, ma), (sb, mb) in zip(seq_a, seq_b):
        assert sa == sb and (ma == mb).all()
        assert ma.shape == (3, wav2vec2_frames(cfg, 16000)) and ma.any()
    t = Noise(cfg, 1, seed=9)                          # the text encoder: hidden / attention dropout only, no host decisions
    t.plan(4, 3, 32)
    assert not t.skip and t.spec_mask is None
