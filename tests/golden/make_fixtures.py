#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE modules.

Run in the build container only (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py

What it does
  * imports the reference's own `src/models/*` (AudioEncoder, TextEncoder,
    CrossModalAttention, AttentiveStatsPooling, FusionLayer,
    AdvancedOpenMaxClassifier, losses, PrototypeMemory) and HuggingFace
    transformers (the third-party library that holds the encoder arithmetic),
  * builds them with small random-init configurations (there is no network, so
    no pretrained weights exist here; random weights in the exact parameter
    layouts pin the arithmetic completely),
  * records inputs, state dicts, outputs and gradients as .npz data files.

Only data (inputs / expected outputs) is written; no reference source text.

Import notes: `models/audio_encoder.py` imports `quality_gates.py`,
`audio_conditioning.py` and `text_encoder.py` imports `asr_integration.py`,
which import librosa / soundfile / whisper at module top.  Those packages are
not installed here, and the code paths that use them are disabled
(use_quality_gates=False, use_audio_conditioning=False,
use_asr_integration=False), so empty placeholder modules are registered for
the import statements only.  To show the placeholders have no numerical
effect, the encoder fixtures are cross-checked below against a direct
HuggingFace composition (Wav2Vec2Model / XLMRobertaModel + the adapter math).
"""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    import transformers.models.wav2vec2.modeling_wav2vec2  # noqa: F401
    import transformers.models.whisper  # noqa: F401
    for name in ("librosa", "soundfile", "whisper"):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    import models  # noqa: F401
    from models import AudioEncoder, TextEncoder, FusionLayer
    from models.classifier import AdvancedOpenMaxClassifier
    from models.cross_attention import CrossModalAttention
    from models.pooling import AttentiveStatsPooling
    from models.losses import LabelSmoothingCrossEntropy, ClassBalancedFocalLoss
    from models.prototypes import PrototypeMemory
    return dict(AudioEncoder=AudioEncoder, TextEncoder=TextEncoder, FusionLayer=FusionLayer,
                AdvancedOpenMaxClassifier=AdvancedOpenMaxClassifier, CrossModalAttention=CrossModalAttention,
                AttentiveStatsPooling=AttentiveStatsPooling, LabelSmoothingCrossEntropy=LabelSmoothingCrossEntropy,
                ClassBalancedFocalLoss=ClassBalancedFocalLoss, PrototypeMemory=PrototypeMemory)


def sd_np(module, prefix="sd."):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(module, prefix="grad."):
    return {prefix + k: p.grad.detach().cpu().numpy().copy() for k, p in module.named_parameters() if p.grad is not None}


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1e6:.2f} MB, {len(arrs)} arrays")


# small configurations used by the committed fixtures ------------------------------------------
A_CFG = dict(hidden=128, layers=2, heads=2, ffn=256, conv_dim=(64,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2),
             conv_stride=(5, 2, 2, 2, 2, 2, 2), pos_kernel=16, pos_groups=4, eps=1e-5)
T_CFG = dict(hidden=128, layers=2, heads=2, ffn=256, vocab=1000, max_pos=66, eps=1e-5, pad_id=1)
VOCAB_WORDS = 996


def make_local_models(tmp, a_cfg, t_cfg):
    from transformers import (Wav2Vec2Config, Wav2Vec2Model, Wav2Vec2FeatureExtractor, XLMRobertaConfig,
                              XLMRobertaModel, PreTrainedTokenizerFast)
    from tokenizers import Tokenizer, models as tkm, pre_tokenizers, processors
    da, dt = os.path.join(tmp, "w2v"), os.path.join(tmp, "xlmr")
    torch.manual_seed(11)
    wc = Wav2Vec2Config(hidden_size=a_cfg["hidden"], num_hidden_layers=a_cfg["layers"],
                        num_attention_heads=a_cfg["heads"], intermediate_size=a_cfg["ffn"],
                        conv_dim=list(a_cfg["conv_dim"]), conv_kernel=list(a_cfg["conv_kernel"]),
                        conv_stride=list(a_cfg["conv_stride"]), num_conv_pos_embeddings=a_cfg["pos_kernel"],
                        num_conv_pos_embedding_groups=a_cfg["pos_groups"])
    Wav2Vec2Model(wc).save_pretrained(da)
    Wav2Vec2FeatureExtractor().save_pretrained(da)
    torch.manual_seed(12)
    xc = XLMRobertaConfig(vocab_size=t_cfg["vocab"], hidden_size=t_cfg["hidden"], num_hidden_layers=t_cfg["layers"],
                          num_attention_heads=t_cfg["heads"], intermediate_size=t_cfg["ffn"],
                          max_position_embeddings=t_cfg["max_pos"], type_vocab_size=1, layer_norm_eps=t_cfg["eps"],
                          pad_token_id=1, bos_token_id=0, eos_token_id=2)
    XLMRobertaModel(xc).save_pretrained(dt)
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for i in range(t_cfg["vocab"] - 4):
        vocab[f"w{i}"] = i + 4
    tok = Tokenizer(tkm.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", special_tokens=[("<s>", 0), ("</s>", 2)])
    PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>",
                            pad_token="<pad>").save_pretrained(dt)
    return da, dt


def main():
    R = _import_reference()
    torch.set_grad_enabled(True)
    tmp = tempfile.mkdtemp(prefix="ser_fix_")
    da, dt = make_local_models(tmp, A_CFG, T_CFG)

    # ---------------- audio encoder -----------------------------------------------------------
    ae = R["AudioEncoder"](model_name=da, adapter_dim=32, use_quality_gates=False, use_audio_conditioning=False).eval()
    g = torch.Generator().manual_seed(1234)
    waves = [0.1 * torch.randn(4000, generator=g), 0.1 * torch.randn(3200, generator=g) + 0.03]
    with torch.no_grad():
        a_seq, a_mask = ae(waves, ["x", "y"])
        # cross-check: direct HF composition, no reference wrapper
        for i, w in enumerate(waves):
            x = ((w - w.mean()) / torch.sqrt(w.var(unbiased=False) + 1e-7))[None]
            s = ae.encoder(x).last_hidden_state[0]
            s = s + ae.adapter(s)
            assert torch.allclose(a_seq[i, : s.shape[0]], s, atol=1e-6), "wrapper != direct HF composition"
    save("audio_encoder.npz", cfg=json.dumps(A_CFG), adapter_dim=32, wave0=waves[0].numpy(), wave1=waves[1].numpy(),
         a_seq=a_seq.numpy(), a_mask=a_mask.numpy(), **sd_np(ae))

    # ---------------- learnable gate-feature fusion (ref audio_encoder.py:25-52, :115-132) ----
    # The DSP that produces the raw 8 quality + 12 conditioning features needs librosa/webrtcvad and cannot run
    # here; the learnable layers behind it can: they are driven directly with synthetic raw features.
    torch.manual_seed(13)
    aeg = R["AudioEncoder"](model_name=da, adapter_dim=32, vad_method="librosa").eval()
    seqg = torch.randn(2, 9, 128, requires_grad=True)
    q_raw, c_raw = torch.rand(2, 8), torch.rand(2, 12)
    outs = []
    for i in range(2):
        qf = aeg.quality_gates.quality_projection(q_raw[i][None])[0]
        cf = aeg.audio_conditioning.conditioning_projection(c_raw[i][None])[0]
        feats = torch.cat([qf, cf])[None].expand(9, -1)
        outs.append(aeg.combined_fusion(torch.cat([seqg[i], feats], dim=-1)))
    outg = torch.stack(outs)
    gg = torch.randn_like(outg)
    (outg * gg).sum().backward()
    gsd = {"sd." + k: v.detach().numpy().copy() for k, v in aeg.state_dict().items() if not k.startswith("encoder.")}
    ggr = {"grad." + k: p.grad.detach().numpy().copy() for k, p in aeg.named_parameters()
           if p.grad is not None and not k.startswith("encoder.")}
    save("gate_fusion.npz", seq=seqg.detach().numpy(), q_raw=q_raw.numpy(), c_raw=c_raw.numpy(), out=outg.detach().numpy(),
         g_out=gg.numpy(), grad_seq=seqg.grad.numpy(), **gsd, **ggr)

    # ---------------- text encoder ------------------------------------------------------------
    te = R["TextEncoder"](model_name=dt, adapter_dim=32).eval()
    texts = ["w1 w2 w3 w4 w5 w6", "w10 w20 w30", "w7 w8 w9 w11 w12 w13 w14 w15 w16"]
    enc = te.tokenizer(texts, padding=True, truncation=True, return_tensors="pt")
    with torch.no_grad():
        t_seq, t_mask = te(texts)
        h = te.encoder(**enc).last_hidden_state
        assert torch.allclose(t_seq, h + te.adapter(h), atol=1e-6)
    save("text_encoder.npz", cfg=json.dumps(T_CFG), adapter_dim=32, texts=json.dumps(texts),
         input_ids=enc["input_ids"].numpy(), attention_mask=enc["attention_mask"].numpy(),
         t_seq=t_seq.numpy(), t_mask=t_mask.numpy(), **sd_np(te))

    # ---------------- cross attention (fwd + grads) -------------------------------------------
    torch.manual_seed(21)
    cross = R["CrossModalAttention"](128, 128, shared_dim=64, num_heads=2, dropout=0.1).eval()
    a = torch.randn(3, 12, 128, requires_grad=True)
    t = torch.randn(3, 11, 128, requires_grad=True)
    am = torch.ones(3, 12)
    tm = torch.ones(3, 11); tm[1, 5:] = 0; tm[2, 9:] = 0
    a_enh, t_enh = cross(a, t, am, tm)
    ga, gt = torch.randn_like(a_enh), torch.randn_like(t_enh)
    ((a_enh * ga).sum() + (t_enh * gt).sum()).backward()
    save("cross.npz", heads=2, a=a.detach().numpy(), t=t.detach().numpy(), a_mask=am.numpy(), t_mask=tm.numpy(),
         a_enh=a_enh.detach().numpy(), t_enh=t_enh.detach().numpy(), g_a_enh=ga.numpy(), g_t_enh=gt.numpy(),
         grad_a=a.grad.numpy(), grad_t=t.grad.numpy(), **sd_np(cross), **grads_np(cross))

    # ---------------- pooling -----------------------------------------------------------------
    torch.manual_seed(22)
    pool = R["AttentiveStatsPooling"](128, hidden_dim=32).eval()
    x = torch.randn(3, 11, 128, requires_grad=True)
    y = pool(x, tm)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    save("pool.npz", x=x.detach().numpy(), mask=tm.numpy(), y=y.detach().numpy(), g_y=gy.numpy(),
         grad_x=x.grad.numpy(), **sd_np(pool), **grads_np(pool))

    # ---------------- fusion ------------------------------------------------------------------
    torch.manual_seed(23)
    fus = R["FusionLayer"](256, 256, 64).eval()
    av = torch.randn(5, 256, requires_grad=True)
    tv = torch.randn(5, 256, requires_grad=True)
    f = fus(av, tv)
    gf = torch.randn_like(f)
    (f * gf).sum().backward()
    save("fusion.npz", a_vec=av.detach().numpy(), t_vec=tv.detach().numpy(), fused=f.detach().numpy(), g_fused=gf.numpy(),
         grad_a_vec=av.grad.numpy(), grad_t_vec=tv.grad.numpy(), **sd_np(fus), **grads_np(fus))

    # ---------------- classifier --------------------------------------------------------------
    torch.manual_seed(24)
    C = 4
    clf = R["AdvancedOpenMaxClassifier"](input_dim=64, num_labels=C, num_layers=3, base_dim=64, dropout=0.15).eval()
    xin = torch.randn(6, 64, requires_grad=True)
    logits, unc, anchor = clf(xin, use_openmax=False, return_uncertainty=True)
    gl, gu = torch.randn_like(logits), torch.randn_like(unc)
    ((logits * gl).sum() + (unc * gu).sum() + 0.1 * anchor).backward()
    arrs = dict(x=xin.detach().numpy(), logits=logits.detach().numpy(), unc=unc.detach().numpy(),
                anchor_loss=np.float32(anchor.item()), g_logits=gl.numpy(), g_unc=gu.numpy(), grad_x=xin.grad.numpy())
    arrs.update(sd_np(clf)); arrs.update(grads_np(clf))
    with torch.no_grad():
        arrs["logits_openmax_unfitted"] = clf(xin, use_openmax=True).numpy()
        feats = torch.randn(40, 32).abs()
        labs = torch.arange(40) % C
        clf.fit_weibull(feats, labs)
        arrs["fit_feats"], arrs["fit_labels"] = feats.numpy(), labs.numpy()
        for k in ("weibull_alpha", "weibull_beta", "weibull_tau", "activation_vectors"):
            arrs["fitted." + k] = getattr(clf, k).numpy().copy()
        arrs["logits_openmax_fitted"] = clf(xin, use_openmax=True).numpy()
    save("classifier.npz", num_layers=3, num_labels=C, **arrs)

    # ---------------- losses (value + grads) --------------------------------------------------
    torch.manual_seed(25)
    lg = (torch.randn(16, C) * 4).requires_grad_()          # some entries beyond the +-10 clamp
    with torch.no_grad():
        lg[0, 1] = 12.5; lg[3, 2] = -11.0
    un = torch.rand(16, 1, requires_grad=True)
    fu = (torch.randn(16, 64) * 3).requires_grad_()
    lab = torch.randint(0, C, (16,))
    protos = R["PrototypeMemory"](C, 64)
    ce = R["LabelSmoothingCrossEntropy"](0.1)(lg, lab)
    fo = R["ClassBalancedFocalLoss"](beta=0.9999, gamma=2.0, num_classes=C)(lg, lab)
    ul = torch.mean(un * (lab == lg.argmax(dim=1)).float())
    pl = protos.prototype_loss(fu, lab)
    total = ce + 0.3 * fo + 0.1 * 0.0 + 0.05 * ul + 0.01 * pl
    total.backward()
    save("losses.npz", logits=lg.detach().numpy(), unc=un.detach().numpy(), fused=fu.detach().numpy(), labels=lab.numpy(),
         prototypes=protos.prototypes.detach().numpy(), ce=np.float32(ce.item()), focal=np.float32(fo.item()),
         unc_loss=np.float32(ul.item()), proto=np.float32(pl.item()), total=np.float32(total.item()),
         grad_logits=lg.grad.numpy(), grad_unc=un.grad.numpy(), grad_fused=fu.grad.numpy(),
         grad_prototypes=protos.prototypes.grad.numpy())

    # ---------------- AdamW + LambdaLR (torch.optim, as train.py:72-83,114-121 uses them) ------
    torch.manual_seed(26)
    p1 = torch.nn.Parameter(torch.randn(37, 5)); p2 = torch.nn.Parameter(torch.randn(129))
    opt = torch.optim.AdamW([{"params": [p1], "lr": 1e-3 * 1.5, "weight_decay": 0.06},
                             {"params": [p2], "lr": 1e-3, "weight_decay": 0.05}], weight_decay=0.05)
    total_steps, wr = 20, 0.1
    W = int(total_steps * wr)

    def lam(step):
        if step < W:
            return float(step) / max(1, W)
        prog = (step - W) / max(1, total_steps - W)
        return 0.5 * (1.0 + torch.cos(torch.tensor(prog * 3.1415926535))).item()
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lam)
    rec = dict(p1_0=p1.detach().numpy().copy(), p2_0=p2.detach().numpy().copy(), total_steps=total_steps, warmup_ratio=wr)
    lrs = []
    for s in range(4):
        g1, g2 = torch.randn_like(p1), torch.randn_like(p2)
        p1.grad, p2.grad = g1.clone(), g2.clone()
        lrs.append([grp["lr"] for grp in opt.param_groups])
        opt.step(); sch.step()
        rec[f"g1_{s}"], rec[f"g2_{s}"] = g1.numpy(), g2.numpy()
        rec[f"p1_{s + 1}"], rec[f"p2_{s + 1}"] = p1.detach().numpy().copy(), p2.detach().numpy().copy()
    rec["lrs"] = np.array(lrs, dtype=np.float64)
    rec["lambda_values"] = np.array([lam(s) for s in range(total_steps + 1)], dtype=np.float64)
    save("adamw.npz", **rec)

    # ---------------- full-size key / shape manifests ------------------------------------------
    from transformers import Wav2Vec2Config, Wav2Vec2Model, XLMRobertaConfig, XLMRobertaModel
    man = {}
    with torch.device("meta"):
        w = Wav2Vec2Model(Wav2Vec2Config())
        xr = XLMRobertaModel(XLMRobertaConfig(vocab_size=250002, max_position_embeddings=514, type_vocab_size=1,
                                              layer_norm_eps=1e-5))
        man["wav2vec2-base"] = {k: list(v.shape) for k, v in w.state_dict().items()}
        man["xlm-roberta-base"] = {k: list(v.shape) for k, v in xr.state_dict().items()}
        man["cross"] = {k: list(v.shape) for k, v in R["CrossModalAttention"](768, 768, 256, 8).state_dict().items()}
        man["pool"] = {k: list(v.shape) for k, v in R["AttentiveStatsPooling"](768).state_dict().items()}
        man["fusion"] = {k: list(v.shape) for k, v in R["FusionLayer"](1536, 1536, 512).state_dict().items()}
        man["classifier"] = {k: list(v.shape) for k, v in
                             R["AdvancedOpenMaxClassifier"](512, 4, 35, 512, 0.15).state_dict().items()}
        man["prototypes"] = {k: list(v.shape) for k, v in R["PrototypeMemory"](4, 512).state_dict().items()}
    man["audio_encoder_small"] = {k: list(v.shape) for k, v in ae.state_dict().items()}
    man["text_encoder_small"] = {k: list(v.shape) for k, v in te.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_manifest.json"), "w") as fjs:
        json.dump(man, fjs, indent=0, sort_keys=True)
    print("wrote state_dict_manifest.json")


if __name__ == "__main__":
    main()
