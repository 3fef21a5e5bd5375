"""BASELINE configs 4 and 5 as GPU tests.

Config 4 — CREMA-D-shaped: 6 classes, 3 s clips (149 frames), data parallel, --augment, hipGraph steps — through
`train.main` on two ranks that share the one GPU of the test box over gloo (RCCL needs a GPU per rank).
Config 5 — stress shapes: Large-sized layers (1024-d, 16 heads, FFN 4096), 10 s clips (499 frames: the chunked
attention kernels, beyond the resident-K/V ones) + 128 tokens, batch 32 (the launch-per-Linear classifier path, M > 16,
and the generic cross-attention kernels, S > 256) against the CPU oracle."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import ser_oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _config4_worker(rank, world, port, tmp, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0",
                          SER_SINGLE_DEVICE="1", SER_DIST_BACKEND="gloo")
        import ser_amd  # noqa: F401
        from ser_amd import train as T
        # 21 utterances in batches of 4 = 6 global batches -> 3 steps per rank and epoch, last batch partial (a second
        # graph shape); 3 s clips -> 149 frames
        f1 = T.main(["--synthetic", "21", "--synthetic_seconds", "3", "--num_labels", "6", "--epochs", "2", "--batch_size", "4",
                     "--save_dir", os.path.join(tmp, f"ck{rank}"), "--audio_model", os.path.join(tmp, "w2v"),
                     "--text_model", os.path.join(tmp, "xlmr"), "--augment", "--graph", "--warmup_ratio", "0.0"])
        flat = None
        if rank == 0:
            cks = sorted(os.listdir(os.path.join(tmp, "ck0")))
            ck = torch.load(os.path.join(tmp, "ck0", cks[-1]), map_location="cpu", weights_only=False)
            flat = float(sum(v.double().sum() for v in ck["classifier"].values() if v.dtype.is_floating_point))
            assert ck["classifier"]["deep_classifier.output_projection.4.weight"].shape[0] == 6
            assert ck["scheduler"]["total"] == 6, "schedule length = steps one rank takes (3 per epoch x 2 epochs)"
        q.put((rank, "ok", f1, flat))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), None, None))
        raise


def test_config4_two_ranks_augment_graph(tmp_path):
    from tests.test_gpu_cli import _local_models
    tmp = str(tmp_path)
    _local_models(tmp)
    # the synthetic corpus speaks "w<i>": give the local tokenizer those words
    from tokenizers import Tokenizer, models as tkm, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for i in range(150):
        vocab[f"w{i}"] = len(vocab)
    tok = Tokenizer(tkm.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", special_tokens=[("<s>", 0), ("</s>", 2)])
    PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>",
                            pad_token="<pad>").save_pretrained(os.path.join(tmp, "xlmr"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_config4_worker, args=(r, 2, port, tmp, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert [r[1] for r in res] == ["ok", "ok"], res
    assert res[0][2] == res[1][2], "both replicas evaluate the same model: same validation F1"


def _replica_worker(rank, world, port, tmp, extra, q):
    """train.main on one of two ranks; reports a digest of every trainable parameter after training."""
    try:
        import hashlib
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0",
                          SER_SINGLE_DEVICE="1", SER_DIST_BACKEND="gloo")
        import ser_amd  # noqa: F401
        from ser_amd import train as T
        made = []

        class Engine(T.HipEngine):
            def __init__(self, *a, **k):
                super().__init__(*a, **k)
                made.append(self)
        T.HipEngine = Engine
        T.main(["--num_labels", "4", "--epochs", "1", "--batch_size", "4", "--save_dir", os.path.join(tmp, f"ck{rank}"),
                "--audio_model", os.path.join(tmp, "w2v"), "--text_model", os.path.join(tmp, "xlmr"), "--warmup_ratio", "0.0"] + extra)
        eng = made[0]
        torch.cuda.synchronize()
        names = [n for n, p in eng.sys.named_parameters() if p.requires_grad]
        flat = torch.cat([p.detach().reshape(-1) for _, p in eng.sys.named_parameters() if p.requires_grad]).cpu()
        assert torch.isfinite(flat).all()
        per = {n: hashlib.sha1(p.detach().cpu().numpy().tobytes()).hexdigest()[:12] for n, p in eng.sys.named_parameters() if p.requires_grad}
        q.put((rank, "ok", hashlib.sha1(flat.numpy().tobytes()).hexdigest(), per, len(names)))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), None, None, 0))
        raise


def _base_width_models(tmp):
    """Wav2Vec2-Base and XLM-R-Base ARCHITECTURES (768-d, 12 layers, 12 heads, FFN 3072, the 7-layer 512-channel conv
    front end) with random weights; only the text vocabulary is cut to 2 048 rows so that the two ranks' tables fit a test."""
    from transformers import Wav2Vec2Config, Wav2Vec2FeatureExtractor, Wav2Vec2Model, XLMRobertaConfig, XLMRobertaModel
    da, dt = os.path.join(tmp, "w2v"), os.path.join(tmp, "xlmr")
    torch.manual_seed(3)
    Wav2Vec2Model(Wav2Vec2Config()).save_pretrained(da)
    Wav2Vec2FeatureExtractor().save_pretrained(da)
    XLMRobertaModel(XLMRobertaConfig(vocab_size=2048, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                                     pad_token_id=1, bos_token_id=0, eos_token_id=2)).save_pretrained(dt)


def _two_replicas(tmp_path, extra, base_width=False, n="22"):
    from tests.test_gpu_cli import _local_models
    tmp = str(tmp_path)
    _base_width_models(tmp) if base_width else _local_models(tmp)
    from tokenizers import Tokenizer, models as tkm, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for i in range(150):
        vocab[f"w{i}"] = len(vocab)
    tok = Tokenizer(tkm.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", special_tokens=[("<s>", 0), ("</s>", 2)])
    PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>",
                            pad_token="<pad>").save_pretrained(os.path.join(tmp, "xlmr"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, tmp, ["--synthetic", n] + extra, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert [r[1] for r in res] == ["ok", "ok"], res
    if res[0][2] != res[1][2]:
        diff = [n for n in res[0][3] if res[0][3][n] != res[1][3].get(n)]
        raise AssertionError(f"replicas diverged in {len(diff)} of {res[0][4]} trainable tensors, e.g. {diff[:8]}")


def test_two_ranks_graph_steps_with_ragged_and_equal_batches_stay_identical(tmp_path):
    """a ragged batch takes the eager path and arms the reducer's per-bucket hooks; a later equal-length
    batch is captured into a graph whose warm-up passes used to fire those hooks (reducing gradients the replay then
    overwrote, and marking the buckets done).  Mixed 3 s / 4 s corpus with --graph: both replicas must end bit-identical."""
    _two_replicas(tmp_path, ["--synthetic_seconds", "3,4", "--graph", "--epochs", "2"])


def test_config4_base_width_two_ranks_augment_graph(tmp_path):
    """BASELINE config 4 at its own model size: Base-width encoders, 6 classes, 3 s clips (149 frames), --augment, hipGraph
    steps, two data-parallel ranks (sharing the test box's one GPU over gloo).  17 clips in batches of 4: the last batch is
    partial, so one rank captures a second graph shape.  Replicas must end bit-identical."""
    _two_replicas(tmp_path, ["--synthetic_seconds", "3", "--num_labels", "6", "--augment", "--graph"], base_width=True, n="17")


def test_two_ranks_with_gates_reduce_the_gate_parameters(tmp_path):
    """the quality / conditioning projections and their fusion Linear live in the audio encoder's gate
    bucket; it is part of SERSystem.buckets() now, so its gradients are all-reduced and the replicas' copies stay equal."""
    _two_replicas(tmp_path, ["--synthetic_seconds", "3", "--use_quality_gates", "--use_audio_conditioning", "--vad_method", "librosa"])


def test_config5_large_shapes_match_oracle():
    import __graft_entry__ as ge
    from transformers import Wav2Vec2Config, XLMRobertaConfig
    import ser_amd  # noqa: F401
    from ser_amd.models import AudioEncoder, TextEncoder
    from ser_amd.system import SERSystem
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    wc = Wav2Vec2Config(hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096)
    xc = XLMRobertaConfig(vocab_size=2048, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
                          max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0,
                          eos_token_id=2)
    ae = AudioEncoder(hf_config=wc, use_quality_gates=False, use_audio_conditioning=False, precision="bf16x3")
    te = TextEncoder(hf_config=xc, precision="bf16x3")
    sysm = SERSystem(ae, te, num_labels=4).to(dev)
    sysm.train()
    sysm.train_dropout = False
    B, T, S = 32, 160000, 128
    g = torch.Generator().manual_seed(9)
    wave = 0.1 * torch.randn(B, T, generator=g)
    ids = torch.randint(4, 2048, (B, S), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    mask = torch.ones(B, S)
    ids[3, S - 20:] = 1                      # one padded row
    ids[3, S - 21] = 2
    mask[3, S - 20:] = 0
    labels = torch.randint(0, 4, (B,), generator=g)
    loss, logits = sysm.loss(wave.to(dev), ids.to(dev), mask.to(dev), labels.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    assert sysm.audio_encoder.engine().out_len(T) == 499

    torch.set_num_threads(min(16, torch.get_num_threads()))
    sds = {k: {n: v.detach().cpu().clone() for n, v in getattr(sysm, k).state_dict().items()} for k in sysm.CKPT_KEYS}
    a_cfg, t_cfg = ge.oracle_cfgs(wc, xc)
    leaf = {k: {n: v.clone().requires_grad_(v.dtype.is_floating_point and not n.startswith("encoder.") and k in ("classifier", "cross", "fusion"))
                for n, v in sd.items()} for k, sd in sds.items()}
    out = O.full_forward(leaf, list(wave), ids, mask, a_cfg, t_cfg, num_layers=35, heads=8, use_openmax=False, training=True)
    ref_loss = O.train_loss(out["logits"], out["unc"], out["fused"], leaf["prototypes"]["prototypes"], labels, 4)
    ref_loss.backward()
    spread = (out["logits"].max(0).values - out["logits"].min(0).values).max().item()
    err = (logits.detach().cpu() - out["logits"].detach()).abs().max().item()
    print(f"config-5 shapes: logits max-abs-err {err:.3e}, oracle spread {spread:.3e}")
    assert spread > 1e-2
    assert err < 1e-3
    assert torch.equal(logits.argmax(1).cpu(), out["logits"].argmax(1))
    assert abs(loss.item() - ref_loss.item()) < 1e-3
    for key, name in (("classifier", "deep_classifier.input_projection.0.weight"), ("classifier", "deep_classifier.residual_layers.17.block.1.weight"),
                      ("cross", "q_a.weight"), ("cross", "out_t.weight"), ("fusion", "proj_a.0.weight")):
        want = leaf[key][name].grad
        got = dict(getattr(sysm, key).named_parameters())[name].grad.cpu()
        rel = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        assert rel < 2e-2, f"{key}.{name}: gradient differs from the oracle (rel {rel:.3e})"
