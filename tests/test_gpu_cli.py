"""BASELINE config 1 as a test: the train.py / eval.py command lines end to end on the GPU —
RAVDESS-style 4-class manifests, batch_size 2, 1 s synthetic audio + dummy text, --epochs 1 —
with small local HuggingFace model directories (no network), then eval.py on the checkpoint it wrote."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _local_models(tmp):
    from tokenizers import Tokenizer, models as tkm, pre_tokenizers, processors
    from transformers import (PreTrainedTokenizerFast, Wav2Vec2Config, Wav2Vec2FeatureExtractor, Wav2Vec2Model,
                              XLMRobertaConfig, XLMRobertaModel)
    da, dt = os.path.join(tmp, "w2v"), os.path.join(tmp, "xlmr")
    torch.manual_seed(3)
    Wav2Vec2Model(Wav2Vec2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                 conv_dim=[64] * 7, num_conv_pos_embeddings=16,
                                 num_conv_pos_embedding_groups=4)).save_pretrained(da)
    Wav2Vec2FeatureExtractor().save_pretrained(da)
    XLMRobertaModel(XLMRobertaConfig(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                                     intermediate_size=256, max_position_embeddings=66, type_vocab_size=1,
                                     layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)).save_pretrained(dt)
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for i, wd in enumerate("audio sample from the ravdess dataset kids are talking by door dogs sitting".split()):
        vocab.setdefault(wd, len(vocab))
    tok = Tokenizer(tkm.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", special_tokens=[("<s>", 0), ("</s>", 2)])
    PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>",
                            pad_token="<pad>").save_pretrained(dt)
    return da, dt


def _manifests(tmp, n_train=6, n_val=4):
    from scipy.io import wavfile
    os.makedirs(os.path.join(tmp, "datasets", "ravdess"), exist_ok=True)
    rng = np.random.default_rng(0)
    texts = ["kids are talking by the door", "dogs are sitting by the door"]

    def write(name, n, off):
        rows = []
        for i in range(n):
            rel = f"ravdess/clip_{off + i}.wav"
            secs = 1.0 if i % 3 else 0.8                       # mostly 1 s, one shorter clip per manifest (ragged batch)
            wavfile.write(os.path.join(tmp, "datasets", rel), 16000,
                          (0.1 * rng.standard_normal(int(16000 * secs)) * 32767).astype(np.int16))
            rows.append(dict(audio=rel, text=texts[i % 2], label=int(i % 4), dataset="ravdess"))
        p = os.path.join(tmp, name)
        with open(p, "w") as f:
            f.write("\n".join(json.dumps(r) for r in rows) + "\n")
        return p
    return write("train_70.jsonl", n_train, 0), write("val_20.jsonl", n_val, 100)


def test_train_then_eval_cli(tmp_path, capsys):
    import ser_amd  # noqa: F401
    from ser_amd import eval as ser_eval, train as ser_train
    tmp = str(tmp_path)
    da, dt = _local_models(tmp)
    tr, va = _manifests(tmp)
    cwd = os.getcwd()
    os.chdir(tmp)                       # load_audio resolves "datasets/<path>" relative to the working directory
    try:
        f1 = ser_train.main(["--train_manifest", tr, "--val_manifest", va, "--epochs", "1", "--batch_size", "2",
                             "--save_dir", os.path.join(tmp, "ck"), "--audio_model", da, "--text_model", dt,
                             "--warmup_ratio", "0.0", "--augment"])
        cks = sorted(os.listdir(os.path.join(tmp, "ck")))
        assert len(cks) == 1 and cks[0].startswith("epoch_0_f1_") and cks[0].endswith(".pt")
        ck = torch.load(os.path.join(tmp, "ck", cks[0]), map_location="cpu", weights_only=False)
        for k in ("audio_encoder", "text_encoder", "cross", "pool_a", "pool_t", "fusion", "classifier", "prototypes",
                  "optimizer", "scheduler", "epoch", "f1"):                       # ref train.py:249-262
            assert k in ck, k
        assert ck["epoch"] == 0 and 0.0 <= f1 <= 1.0
        assert float(ck["classifier"]["weibull_alpha"][0]) == 2.5             # fit_weibull ran after the last epoch
        ser_eval.main(["--manifest", va, "--checkpoint", os.path.join(tmp, "ck", cks[0]), "--batch_size", "2",
                       "--audio_model", da, "--text_model", dt, "--calibrate", "--val_manifest", va, "--use_tta",
                       "--num_tta", "3"])
        out = capsys.readouterr().out
        assert "Weighted F1 Score" in out and "Optimal temperature" in out
    finally:
        os.chdir(cwd)


def test_train_and_eval_cli_with_the_reference_default_front_end(tmp_path, capsys):
    """--use_quality_gates --use_audio_conditioning: the configuration the reference's own train.py / eval.py build
    (AudioEncoder() defaults, ref train.py:54, eval.py:92) — gates, conditioning and feature fusion on the device."""
    import ser_amd  # noqa: F401
    from ser_amd import eval as ser_eval, train as ser_train
    tmp = str(tmp_path)
    da, dt = _local_models(tmp)
    tr, va = _manifests(tmp)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        gates = ["--use_quality_gates", "--use_audio_conditioning", "--vad_method", "librosa"]
        ser_train.main(["--train_manifest", tr, "--val_manifest", va, "--epochs", "1", "--batch_size", "2",
                        "--save_dir", os.path.join(tmp, "ck"), "--audio_model", da, "--text_model", dt,
                        "--warmup_ratio", "0.0"] + gates)
        cks = sorted(os.listdir(os.path.join(tmp, "ck")))
        ck = torch.load(os.path.join(tmp, "ck", cks[0]), map_location="cpu", weights_only=False)
        init = torch.load(os.path.join(tmp, "ck", cks[0]), map_location="cpu", weights_only=False)["audio_encoder"]
        for k in ("quality_gates.quality_projection.0.weight", "audio_conditioning.conditioning_projection.3.bias",
                  "combined_fusion.0.weight"):                                   # the reference's default-flag keys
            assert k in ck["audio_encoder"], k
        assert torch.isfinite(init["combined_fusion.0.weight"]).all()
        ser_eval.main(["--manifest", va, "--checkpoint", os.path.join(tmp, "ck", cks[0]), "--batch_size", "2",
                       "--audio_model", da, "--text_model", dt] + gates)
        assert "Weighted F1 Score" in capsys.readouterr().out
    finally:
        os.chdir(cwd)


def test_train_cli_full_fine_tune_with_amp_and_graphs(tmp_path):
    """BASELINE config 3 through the command line: --unfreeze_encoders --use_amp --graph (encoders' training-mode noise on, as the
    reference's .train() leaves it: LayerDrop / SpecAugment decisions are staged as device words before each captured step), one
    epoch over equal-length and ragged batches; the encoder weights in the checkpoint have moved."""
    import ser_amd  # noqa: F401
    from ser_amd import train as ser_train
    tmp = str(tmp_path)
    da, dt = _local_models(tmp)
    tr, va = _manifests(tmp, n_train=8, n_val=4)
    from transformers import Wav2Vec2Model
    w0 = Wav2Vec2Model.from_pretrained(da).state_dict()["encoder.layers.0.attention.q_proj.weight"].clone()
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        f1 = ser_train.main(["--train_manifest", tr, "--val_manifest", va, "--epochs", "1", "--batch_size", "2",
                             "--save_dir", os.path.join(tmp, "ck"), "--audio_model", da, "--text_model", dt,
                             "--warmup_ratio", "0.0", "--unfreeze_encoders", "--use_amp", "--graph"])
        cks = sorted(os.listdir(os.path.join(tmp, "ck")))
        assert len(cks) == 1 and 0.0 <= f1 <= 1.0
        ck = torch.load(os.path.join(tmp, "ck", cks[0]), map_location="cpu", weights_only=False)
        w1 = ck["audio_encoder"]["encoder.encoder.layers.0.attention.q_proj.weight"]
        assert torch.isfinite(w1).all() and (w1 - w0).abs().max().item() > 0.0, "the encoders are trained"
    finally:
        os.chdir(cwd)
