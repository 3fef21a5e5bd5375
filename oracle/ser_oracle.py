"""CPU oracle for the multimodal SER hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 restatement of the reference algorithm
(kananmittal/Multilingual-Multimodal-Speech-Emotion-Recognition, `src/models/*`
plus the HuggingFace `transformers` 5.15.0 Wav2Vec2 / XLM-RoBERTa arithmetic
that the reference delegates to).  It imports nothing from the reference and
nothing from the product package.  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import it; the product path never does.

Pinning: every function below is checked against golden vectors produced by
running the reference's own modules in the build container
(`tests/golden/make_fixtures.py` -> `tests/golden/*.npz`,
checked by `tests/test_oracle_golden.py`).  The reference ships no tests of its
own, so those vectors are the pin.

All functions work on *state dicts with the reference's key names* so that a
reference checkpoint entry can be fed in unchanged.  Dropout is the identity
(eval-mode / p = 0 semantics); SpecAugment and LayerDrop are off.

hf: = transformers/models/... ; ref: = /root/reference/src/...
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# ----------------------------------------------------------------------------
# configuration records (plain dicts; the values HF keeps in config.json)
# ----------------------------------------------------------------------------

def wav2vec2_config(hidden=768, layers=12, heads=12, ffn=3072,
                    conv_dim=(512,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2),
                    conv_stride=(5, 2, 2, 2, 2, 2, 2), pos_kernel=128,
                    pos_groups=16, eps=1e-5):
    return dict(hidden=hidden, layers=layers, heads=heads, ffn=ffn,
                conv_dim=tuple(conv_dim), conv_kernel=tuple(conv_kernel),
                conv_stride=tuple(conv_stride), pos_kernel=pos_kernel,
                pos_groups=pos_groups, eps=eps)


def xlmr_config(hidden=768, layers=12, heads=12, ffn=3072, vocab=250002,
                max_pos=514, eps=1e-5, pad_id=1):
    return dict(hidden=hidden, layers=layers, heads=heads, ffn=ffn, vocab=vocab,
                max_pos=max_pos, eps=eps, pad_id=pad_id)


# ----------------------------------------------------------------------------
# small primitives, written out
# ----------------------------------------------------------------------------

def layer_norm(x: Tensor, g: Tensor, b: Tensor, eps: float) -> Tensor:
    """(x-mean)/sqrt(var_biased+eps)*g+b over the last dim (torch nn.LayerNorm)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def gelu(x: Tensor) -> Tensor:
    """Exact erf GELU (hf: activations.py GELUActivation)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


def sub(sd: SD, prefix: str) -> SD:
    """State-dict view with `prefix` stripped."""
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------
# A0  Training-mode dropout: y = x * m / (1 - p).  No RNG stream can match torch's, so the build defines its own
# counter-based mask generator (csrc/ser_common.h: ser_drop_mult, keyed by a 64-bit state, a layer id and the flat
# element index); this is its restatement, so that the oracle can run a training step with the SAME masks.
# Layer ids are those SERSystem assigns (system.py): cross 1-4 (output a, output t, attention a, attention t),
# fusion 5-6, classifier 7 (input projection), 8 (output projection), 9 (uncertainty head), 16 + 2 i (+1) block i.
# ----------------------------------------------------------------------------

def _mix32(h):
    import numpy as np
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16); h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13); h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def dropout_mult(state: int, site: int, shape, p: float) -> Tensor:
    """Multipliers (0 or 1 / (1 - p)) of one dropout layer for a tensor of `shape` (flat element index = C order)."""
    import numpy as np
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint32)
        lo, hi = np.uint32(state & 0xFFFFFFFF), np.uint32((state >> 32) & 0xFFFFFFFF)
        h = _mix32((idx * np.uint32(0x9E3779B1)).astype(np.uint32) ^ lo)
        h = _mix32((h + np.uint32(site) * np.uint32(0x85EBCA77) + hi).astype(np.uint32))
    thresh = np.uint32(min(np.float32(p) * np.float32(4294967296.0), np.float32(4294967040.0)))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy(np.where(h >= thresh, scale, np.float32(0.0)).astype(np.float32).reshape(shape))


class DropoutPlan:
    """The masks of one training step: `state` = the generator state of that step, p per module as constructed."""

    def __init__(self, state: int, p_cross: float = 0.1, p_fusion: float = 0.1, p_classifier: float = 0.15):
        self.state, self.p_cross, self.p_fusion, self.p_classifier = int(state), p_cross, p_fusion, p_classifier

    def mult(self, site: int, shape, p: float) -> Tensor:
        return dropout_mult(self.state, site, tuple(shape), p)


def mha_core(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float,
             key_bias: Optional[Tensor], p_mult: Optional[Tensor] = None) -> Tensor:
    """softmax(q k^T * scale + key_bias) v per head.  q [B,Sq,D] k,v [B,Sk,D];
    key_bias [B,Sk] additive (0 / -inf) or None."""
    B, Sq, D = q.shape
    Sk = k.shape[1]
    hd = D // heads
    qh = q.view(B, Sq, heads, hd).transpose(1, 2)
    kh = k.view(B, Sk, heads, hd).transpose(1, 2)
    vh = v.view(B, Sk, heads, hd).transpose(1, 2)
    s = (qh @ kh.transpose(2, 3)) * scale
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    if p_mult is not None:                       # attention-probability dropout, [B, heads, Sq, Sk]
        p = p * p_mult
    o = p @ vh
    return o.transpose(1, 2).reshape(B, Sq, D)


# ----------------------------------------------------------------------------
# A1  Wav2Vec2 (hf: wav2vec2/modeling_wav2vec2.py, feature_extraction_wav2vec2.py)
# ----------------------------------------------------------------------------

def normalise_waveform(x: Tensor) -> Tensor:
    """hf: feature_extraction_wav2vec2.py:78-96 zero_mean_unit_var_norm."""
    return (x - x.mean()) / torch.sqrt(x.var(unbiased=False) + 1e-7)


def conv_out_len(L: int, cfg) -> int:
    for k, s in zip(cfg["conv_kernel"], cfg["conv_stride"]):
        L = (L - k) // s + 1
    return L


def wav2vec2_features(sd: SD, x: Tensor, cfg) -> Tensor:
    """Feature encoder, hf: modeling_wav2vec2.py:302-323,254-272,409-419.
    x [B,T] normalised -> [B,S,512] (channels last)."""
    h = x[:, None, :]
    for i, (k, s) in enumerate(zip(cfg["conv_kernel"], cfg["conv_stride"])):
        w = sd[f"feature_extractor.conv_layers.{i}.conv.weight"]
        h = F.conv1d(h, w, None, stride=s)
        if i == 0:
            g = sd["feature_extractor.conv_layers.0.layer_norm.weight"]
            b = sd["feature_extractor.conv_layers.0.layer_norm.bias"]
            mu = h.mean(dim=2, keepdim=True)
            var = ((h - mu) ** 2).mean(dim=2, keepdim=True)
            h = (h - mu) / torch.sqrt(var + 1e-5) * g[None, :, None] + b[None, :, None]
        h = gelu(h)
    return h.transpose(1, 2)


def wav2vec2_pos_conv_weight(sd: SD) -> Tensor:
    """weight_norm(dim=2): W = g * v / ||v||_{dims 0,1}.  hf :326-349."""
    g = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    v = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    nrm = torch.sqrt((v * v).sum(dim=(0, 1), keepdim=True))
    return g * v / nrm


class EncoderNoise:
    """Training-mode noise of one encoder for one step (the reference calls .train() on both encoders, ref train.py:124):
    HF's dropout sites with the build's mask generator (state = the step's generator state; site ids as in
    models/_finetune.py: 1000 + 500 * encoder + 8 * layer + {0 attention probabilities, 1 hidden after attention,
    2 activation, 3 hidden after FFN}, + 400 feature projection, + 401 encoder input, + 402 embeddings), the layers LayerDrop
    skips (hf wav2vec2 :700-703) and the SpecAugment rows (hf :1293-1302) as given sets / masks."""

    def __init__(self, state: int, enc: int, p_hidden=0.1, p_attn=0.1, p_act=0.1, p_featproj=0.0, skip=(), spec_mask=None):
        self.state, self.enc = int(state), int(enc)
        self.p_hidden, self.p_attn, self.p_act, self.p_featproj = p_hidden, p_attn, p_act, p_featproj
        self.skip, self.spec_mask = set(skip), spec_mask

    def site(self, layer, k):
        return 1000 + 500 * self.enc + 8 * layer + k

    def drop(self, x: Tensor, p: float, site: int) -> Tensor:
        return x if p <= 0.0 else x * dropout_mult(self.state, site, tuple(x.shape), p)


def transformer_layer_postln(h: Tensor, p: SD, heads: int, eps: float,
                             names: Dict[str, str], key_bias: Optional[Tensor],
                             noise: Optional["EncoderNoise"] = None, layer: int = 0) -> Tensor:
    """Post-LN block shared by Wav2Vec2 (hf :591-608) and XLM-R (hf xlm_roberta :421-463)."""
    D = h.shape[-1]
    scale = (D // heads) ** -0.5
    q = linear(h, p[names["q"] + ".weight"], p[names["q"] + ".bias"])
    k = linear(h, p[names["k"] + ".weight"], p[names["k"] + ".bias"])
    v = linear(h, p[names["v"] + ".weight"], p[names["v"] + ".bias"])
    pm = None
    if noise is not None and noise.p_attn > 0:
        pm = dropout_mult(noise.state, noise.site(layer, 0), (h.shape[0], heads, h.shape[1], h.shape[1]), noise.p_attn)
    ctx = mha_core(q, k, v, heads, scale, key_bias, pm)
    a = linear(ctx, p[names["o"] + ".weight"], p[names["o"] + ".bias"])
    if noise is not None:
        a = noise.drop(a, noise.p_hidden, noise.site(layer, 1))
    h = layer_norm(h + a, p[names["ln1"] + ".weight"], p[names["ln1"] + ".bias"], eps)
    f = gelu(linear(h, p[names["f1"] + ".weight"], p[names["f1"] + ".bias"]))
    if noise is not None:
        f = noise.drop(f, noise.p_act, noise.site(layer, 2))
    f = linear(f, p[names["f2"] + ".weight"], p[names["f2"] + ".bias"])
    if noise is not None:
        f = noise.drop(f, noise.p_hidden, noise.site(layer, 3))
    return layer_norm(h + f, p[names["ln2"] + ".weight"], p[names["ln2"] + ".bias"], eps)


W2V_NAMES = dict(q="attention.q_proj", k="attention.k_proj", v="attention.v_proj",
                 o="attention.out_proj", ln1="layer_norm",
                 f1="feed_forward.intermediate_dense", f2="feed_forward.output_dense",
                 ln2="final_layer_norm")
XLMR_NAMES = dict(q="attention.self.query", k="attention.self.key", v="attention.self.value",
                  o="attention.output.dense", ln1="attention.output.LayerNorm",
                  f1="intermediate.dense", f2="output.dense", ln2="output.LayerNorm")


def wav2vec2_forward(sd: SD, x: Tensor, cfg, noise: Optional["EncoderNoise"] = None) -> Tensor:
    """Wav2Vec2Model.forward (eval mode; with `noise`, the training-mode sites).  x [B,T] already normalised.  -> [B,S,H]."""
    eps = cfg["eps"]
    feats = wav2vec2_features(sd, x, cfg)
    e = layer_norm(feats, sd["feature_projection.layer_norm.weight"],
                   sd["feature_projection.layer_norm.bias"], eps)
    z = linear(e, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    if noise is not None:
        z = noise.drop(z, noise.p_featproj, 1000 + 400)
        if noise.spec_mask is not None:
            mk = torch.as_tensor(noise.spec_mask, dtype=torch.bool)
            z = torch.where(mk[:, :, None], sd["masked_spec_embed"][None, None, :], z)
    # positional conv embedding, hf :351-368, :689-692
    W = wav2vec2_pos_conv_weight(sd)
    K = cfg["pos_kernel"]
    pc = F.conv1d(z.transpose(1, 2), W, sd["encoder.pos_conv_embed.conv.bias"],
                  padding=K // 2, groups=cfg["pos_groups"])
    if K % 2 == 0:
        pc = pc[:, :, :-1]
    h = z + gelu(pc).transpose(1, 2)
    h = layer_norm(h, sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], eps)
    if noise is not None:
        h = noise.drop(h, noise.p_hidden, 1000 + 401)
    for i in range(cfg["layers"]):
        if noise is not None and i in noise.skip:
            continue
        h = transformer_layer_postln(h, sub(sd, f"encoder.layers.{i}."), cfg["heads"], eps, W2V_NAMES, None, noise, i)
    return h


def adapter(x: Tensor, p: SD) -> Tensor:
    """x + W2 relu(W1 x + b1) + b2   (ref: models/audio_encoder.py:19-21,112)."""
    return x + linear(torch.relu(linear(x, p["0.weight"], p["0.bias"])), p["2.weight"], p["2.bias"])


def gate_feature_fusion(sd: SD, seq: Tensor, q_raw: Tensor, c_raw: Tensor) -> Tensor:
    """Learnable part of the quality-gate / conditioning path (ref models/audio_encoder.py:115-132 with
    quality_gates.py:439-444,554 and audio_conditioning.py:455-460,578): project the raw 8 + 12 features, broadcast
    over frames, concatenate and apply combined_fusion (Linear + ReLU; dropout is the identity).
    seq [B,S,H], q_raw [B,8], c_raw [B,12]."""
    qf = linear(torch.relu(linear(q_raw, sd["quality_gates.quality_projection.0.weight"], sd["quality_gates.quality_projection.0.bias"])),
                sd["quality_gates.quality_projection.3.weight"], sd["quality_gates.quality_projection.3.bias"])
    cf = linear(torch.relu(linear(c_raw, sd["audio_conditioning.conditioning_projection.0.weight"],
                                  sd["audio_conditioning.conditioning_projection.0.bias"])),
                sd["audio_conditioning.conditioning_projection.3.weight"], sd["audio_conditioning.conditioning_projection.3.bias"])
    f = torch.cat([qf, cf], dim=-1)[:, None, :].expand(-1, seq.shape[1], -1)
    return torch.relu(linear(torch.cat([seq, f], dim=-1), sd["combined_fusion.0.weight"], sd["combined_fusion.0.bias"]))


def audio_encoder_forward(sd: SD, waves: Sequence[Tensor], cfg,
                          gate_features: Optional[Sequence[Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """ref: models/audio_encoder.py:54-172 with quality gates / conditioning off
    (or, when `gate_features` (20-d per clip, already projected) is given, the
    combined_fusion branch :115-132).  Per-clip batch-1 encoder calls, zero-pad
    to the longest, mask = ones (:140-166)."""
    enc = sub(sd, "encoder.")
    seqs = []
    for i, w in enumerate(waves):
        x = normalise_waveform(w.float())[None, :]
        s = wav2vec2_forward(enc, x, cfg)[0]
        s = adapter(s, sub(sd, "adapter."))
        if gate_features is not None:
            f = gate_features[i][None, :].expand(s.shape[0], -1)
            s = torch.relu(linear(torch.cat([s, f], dim=-1), sd["combined_fusion.0.weight"],
                                  sd["combined_fusion.0.bias"]))
        seqs.append(s)
    S = max(s.shape[0] for s in seqs)
    out = torch.zeros(len(seqs), S, seqs[0].shape[1])
    for i, s in enumerate(seqs):
        out[i, : s.shape[0]] = s
    return out, torch.ones(len(seqs), S)


# ----------------------------------------------------------------------------
# A2  XLM-RoBERTa (hf: xlm_roberta/modeling_xlm_roberta.py)
# ----------------------------------------------------------------------------

def xlmr_position_ids(ids: Tensor, pad_id: int) -> Tensor:
    """hf :142-155 create_position_ids_from_input_ids."""
    m = (ids != pad_id).int()
    return (torch.cumsum(m, dim=1) * m).long() + pad_id


def xlmr_forward(sd: SD, ids: Tensor, attn_mask: Tensor, cfg, noise: Optional["EncoderNoise"] = None) -> Tensor:
    """XLMRobertaModel.forward (eval; with `noise`, the training-mode dropout sites).  ids [B,S] int64, attn_mask [B,S] 1/0."""
    eps = cfg["eps"]
    pos = xlmr_position_ids(ids, cfg["pad_id"])
    # nn.Embedding(padding_idx=pad) for words and positions (hf modeling_xlm_roberta.py:75-95): same values, and the pad
    # row receives no gradient when the encoder is unfrozen
    e = (F.embedding(ids, sd["embeddings.word_embeddings.weight"], padding_idx=cfg["pad_id"])
         + sd["embeddings.token_type_embeddings.weight"][0]
         + F.embedding(pos, sd["embeddings.position_embeddings.weight"], padding_idx=cfg["pad_id"]))
    h = layer_norm(e, sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], eps)
    if noise is not None:
        h = noise.drop(h, noise.p_hidden, 1000 + 500 + 402)
    key_bias = torch.zeros(attn_mask.shape, dtype=torch.float32)
    key_bias = key_bias.masked_fill(attn_mask == 0, float("-inf"))
    for i in range(cfg["layers"]):
        h = transformer_layer_postln(h, sub(sd, f"encoder.layer.{i}."), cfg["heads"], eps, XLMR_NAMES, key_bias, noise, i)
    return h


def text_encoder_forward(sd: SD, ids: Tensor, attn_mask: Tensor, cfg) -> Tuple[Tensor, Tensor]:
    """ref: models/text_encoder.py:51-57,75-78 (ASR branch off)."""
    h = xlmr_forward(sub(sd, "encoder."), ids, attn_mask, cfg)
    return adapter(h, sub(sd, "adapter.")), attn_mask.float()


# ----------------------------------------------------------------------------
# A3  CrossModalAttention (ref: models/cross_attention.py:32-53;
#     torch nn.functional.multi_head_attention_forward)
# ----------------------------------------------------------------------------

def _cross_dir(x_q: Tensor, x_kv: Tensor, kv_mask: Optional[Tensor], sd: SD,
               q: str, k: str, v: str, attn: str, out: str, norm: str, heads: int,
               drop: Optional["DropoutPlan"] = None, site_out: int = 0, site_attn: int = 0) -> Tensor:
    E = sd[f"{q}.weight"].shape[0]
    qq = linear(x_q, sd[f"{q}.weight"], sd[f"{q}.bias"])
    kk = linear(x_kv, sd[f"{k}.weight"], sd[f"{k}.bias"])
    vv = linear(x_kv, sd[f"{v}.weight"], sd[f"{v}.bias"])
    Wi, bi = sd[f"{attn}.in_proj_weight"], sd[f"{attn}.in_proj_bias"]
    qq = linear(qq, Wi[:E], bi[:E])
    kk = linear(kk, Wi[E:2 * E], bi[E:2 * E])
    vv = linear(vv, Wi[2 * E:], bi[2 * E:])
    key_bias = None
    if kv_mask is not None:
        key_bias = torch.zeros(kv_mask.shape, dtype=torch.float32).masked_fill(kv_mask == 0, float("-inf"))
    pm = None
    if drop is not None:
        pm = drop.mult(site_attn, (x_q.shape[0], heads, x_q.shape[1], x_kv.shape[1]), drop.p_cross)
    ctx = mha_core(qq, kk, vv, heads, (E // heads) ** -0.5, key_bias, pm)
    ctx = linear(ctx, sd[f"{attn}.out_proj.weight"], sd[f"{attn}.out_proj.bias"])
    o = linear(ctx, sd[f"{out}.weight"], sd[f"{out}.bias"])
    if drop is not None:
        o = o * drop.mult(site_out, o.shape, drop.p_cross)
    return layer_norm(x_q + o, sd[f"{norm}.weight"], sd[f"{norm}.bias"], 1e-5)


def cross_attention_forward(sd: SD, a: Tensor, t: Tensor, a_mask: Optional[Tensor],
                            t_mask: Optional[Tensor], heads: int = 8,
                            drop: Optional["DropoutPlan"] = None) -> Tuple[Tensor, Tensor]:
    a_enh = _cross_dir(a, t, t_mask, sd, "q_a", "k_t", "v_t", "attn_a", "out_a", "norm_a", heads, drop, 1, 3)
    t_enh = _cross_dir(t, a, a_mask, sd, "q_t", "k_a", "v_a", "attn_t", "out_t", "norm_t", heads, drop, 2, 4)
    return a_enh, t_enh


# ----------------------------------------------------------------------------
# A4  AttentiveStatsPooling (ref: models/pooling.py:15-28)
# ----------------------------------------------------------------------------

def pooling_forward(sd: SD, x: Tensor, mask: Optional[Tensor]) -> Tensor:
    l = linear(torch.tanh(linear(x, sd["attention.0.weight"], sd["attention.0.bias"])),
               sd["attention.2.weight"], sd["attention.2.bias"]).squeeze(-1)
    if mask is not None:
        l = l.masked_fill(mask == 0, float("-inf"))
    a = torch.softmax(l, dim=-1).unsqueeze(-1)
    mean = (a * x).sum(dim=1)
    var = (a * (x - mean.unsqueeze(1)) ** 2).sum(dim=1)
    return torch.cat([mean, torch.sqrt(var + 1e-6)], dim=-1)


# ----------------------------------------------------------------------------
# A5  FusionLayer (ref: models/fusion.py:18-25)
# ----------------------------------------------------------------------------

def fusion_forward(sd: SD, a_vec: Tensor, t_vec: Tensor, drop: Optional["DropoutPlan"] = None) -> Tensor:
    def mlp(x, p0, p1, site=0):
        h = torch.relu(linear(x, sd[p0 + ".weight"], sd[p0 + ".bias"]))
        if drop is not None and site:
            h = h * drop.mult(site, h.shape, drop.p_fusion)
        return linear(h, sd[p1 + ".weight"], sd[p1 + ".bias"])
    a = mlp(a_vec, "proj_a.0", "proj_a.3", 5)
    t = mlp(t_vec, "proj_t.0", "proj_t.3", 6)
    wa = torch.sigmoid(mlp(a, "gate_a.0", "gate_a.2"))
    wt = torch.sigmoid(mlp(t, "gate_t.0", "gate_t.2"))
    ws = wa + wt + 1e-8
    return (wa / ws) * a + (wt / ws) * t


# ----------------------------------------------------------------------------
# A6  AdvancedOpenMaxClassifier (ref: models/classifier.py:200-305)
# ----------------------------------------------------------------------------

def classifier_features(sd: SD, x: Tensor, num_layers: int, drop: Optional["DropoutPlan"] = None) -> Tensor:
    """Penultimate 256-d features (ref :203-218; same loop as train.py:222-236)."""
    d = "deep_classifier."
    dm = (lambda site, t_: t_ * drop.mult(site, t_.shape, drop.p_classifier)) if drop is not None else (lambda site, t_: t_)
    h = torch.relu(layer_norm(linear(x, sd[d + "input_projection.0.weight"], sd[d + "input_projection.0.bias"]),
                              sd[d + "input_projection.1.weight"], sd[d + "input_projection.1.bias"], 1e-5))
    h = dm(7, h)
    for i in range(num_layers):
        h = layer_norm(h, sd[d + f"layer_norms.{i}.weight"], sd[d + f"layer_norms.{i}.bias"], 1e-5)
        r = d + f"residual_layers.{i}.block."
        u = layer_norm(h, sd[r + "0.weight"], sd[r + "0.bias"], 1e-5)
        u = dm(16 + 2 * i, torch.relu(linear(u, sd[r + "1.weight"], sd[r + "1.bias"])))
        h = h + dm(16 + 2 * i + 1, linear(u, sd[r + "4.weight"], sd[r + "4.bias"]))
    f = linear(h, sd[d + "output_projection.0.weight"], sd[d + "output_projection.0.bias"])
    return dm(8, torch.relu(layer_norm(f, sd[d + "output_projection.1.weight"], sd[d + "output_projection.1.bias"], 1e-5)))


def openmax_adjust(sd: SD, feats: Tensor, logits: Tensor) -> Tensor:
    """ref :240-275."""
    A = sd["activation_vectors"]
    d = torch.sqrt(((feats[:, None, :] - A[None, :, :]) ** 2).sum(dim=-1))
    beta = torch.clamp(sd["weibull_beta"], min=1e-6)
    sx = torch.clamp(d - sd["weibull_tau"][None, :], min=0)
    cdf = 1 - torch.exp(-torch.pow(sx / beta[None, :], sd["weibull_alpha"][None, :]))
    p = torch.clamp(cdf.max(dim=1).values, min=0.0)
    scale = torch.where(p > 0.3, 1 - 0.8 * p, torch.ones_like(p))
    return logits * scale[:, None]


def classifier_forward(sd: SD, x: Tensor, num_layers: int = 35, use_openmax: bool = True,
                       training: bool = False, drop: Optional["DropoutPlan"] = None):
    """-> (logits, uncertainty [B,1], anchor_loss (exactly 0.), features)."""
    f = classifier_features(sd, x, num_layers, drop)
    logits = linear(f, sd["deep_classifier.output_projection.4.weight"], sd["deep_classifier.output_projection.4.bias"])
    u = torch.relu(linear(f, sd["uncertainty_head.0.weight"], sd["uncertainty_head.0.bias"]))
    if drop is not None:
        u = u * drop.mult(9, u.shape, drop.p_classifier)
    unc = torch.sigmoid(linear(u, sd["uncertainty_head.3.weight"], sd["uncertainty_head.3.bias"]))
    if use_openmax and not training:
        logits = openmax_adjust(sd, f, logits)
    # ClassAnchorClustering.compute_clustering_loss == mean(clamp(s - max s, min=0)) == 0 (ref :58-70)
    return logits, unc, torch.zeros(()), f


def fit_weibull(feats: Tensor, labels: Tensor, num_labels: int, sd: SD) -> SD:
    """ref :277-305.  Returns updated copies of the four buffers."""
    out = {k: sd[k].clone() for k in ("weibull_alpha", "weibull_beta", "weibull_tau", "activation_vectors")}
    for c in range(num_labels):
        m = labels == c
        if int(m.sum()) == 0:
            continue
        cf = feats[m]
        mu = cf.mean(dim=0)
        out["activation_vectors"][c] = mu
        d = torch.sqrt(((cf - mu) ** 2).sum(dim=1)).numpy()
        out["weibull_alpha"][c] = 2.5
        out["weibull_beta"][c] = float(d.std() * 1.5)
        out["weibull_tau"][c] = float(d.min() * 0.8)
    return out


# ----------------------------------------------------------------------------
# A7  losses (ref: models/losses.py, models/prototypes.py, train.py:154-168)
# ----------------------------------------------------------------------------

def label_smoothing_ce(logits: Tensor, target: Tensor, smoothing: float = 0.1) -> Tensor:
    C = logits.shape[-1]
    z = logits.clamp(-10.0, 10.0)
    lp = z - torch.logsumexp(z, dim=-1, keepdim=True)
    dist = torch.full_like(lp, smoothing / (C - 1))
    dist.scatter_(1, target[:, None], 1.0 - smoothing)
    return (-(dist * lp).sum(dim=-1)).mean()


def class_balanced_focal(logits: Tensor, target: Tensor, num_classes: int,
                         beta: float = 0.9999, gamma: float = 2.0) -> Tensor:
    counts = torch.bincount(target, minlength=num_classes).float().clamp(min=1.0)
    eff = (1.0 - torch.pow(torch.tensor(beta), counts)).clamp(min=1e-6)
    w = (1.0 - beta) / eff
    w = w / (w.sum() + 1e-8) * num_classes
    z = logits.clamp(-10.0, 10.0)
    lp = z - torch.logsumexp(z, dim=-1, keepdim=True)
    lpt = lp.gather(1, target[:, None]).squeeze(1)
    pt = torch.exp(lpt).clamp(min=1e-6, max=1.0)
    ce = -w[target] * lpt
    return (torch.pow(1.0 - pt, gamma) * ce).mean()


def prototype_loss(protos: Tensor, emb: Tensor, labels: Tensor, margin: float = 0.5) -> Tensor:
    e = emb.clamp(-10.0, 10.0)
    pos = torch.sqrt(((e - protos[labels]) ** 2).sum(dim=1)).mean()
    d = torch.sqrt(((e[:, None, :] - protos[None, :, :]) ** 2).sum(dim=2) + 1e-6)
    own = torch.zeros_like(d, dtype=torch.bool)
    own[torch.arange(e.shape[0]), labels] = True
    nd = d.masked_fill(own, float("inf")).clamp(max=10.0)
    neg = (-torch.logsumexp(-nd, dim=1)).mean()
    return pos + margin - neg


def train_loss(logits: Tensor, unc: Tensor, fused: Tensor, protos: Tensor, labels: Tensor,
               num_classes: int, use_proto: bool = True) -> Tensor:
    """ref train.py:154-168: CE_ls + 0.3 focal + 0.1*0 + 0.05 unc-term + 0.01 proto."""
    loss = label_smoothing_ce(logits, labels) + 0.3 * class_balanced_focal(logits, labels, num_classes)
    correct = (labels == logits.argmax(dim=1)).float()
    loss = loss + 0.05 * (unc * correct).mean()   # [B,1]*[B] broadcasts to [B,B] in the reference
    if use_proto:
        loss = loss + 0.01 * prototype_loss(protos, fused, labels)
    return loss


# ----------------------------------------------------------------------------
# A8  AdamW + LambdaLR (ref: train.py:72-83,114-121; torch.optim.AdamW)
# ----------------------------------------------------------------------------

def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float,
               b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """One decoupled-decay Adam update, torch semantics (single_tensor path)."""
    p = p * (1 - lr * wd)
    m = m + (g - m) * (1 - b1)          # lerp
    v = v * b2 + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def lr_lambda(step: int, total_steps: int, warmup_ratio: float) -> float:
    W = int(total_steps * warmup_ratio)
    if step < W:
        return float(step) / max(1, W)
    prog = (step - W) / max(1, total_steps - W)
    # the reference evaluates the cosine in float32 (torch.cos(torch.tensor(...)).item())
    return 0.5 * (1.0 + torch.cos(torch.tensor(prog * 3.1415926535)).item())


# ----------------------------------------------------------------------------
# the whole path (ref: train.py:145-152) — used by smoke() and the CPU baseline
# ----------------------------------------------------------------------------

def full_forward(sds: Dict[str, SD], waves: Sequence[Tensor], ids: Tensor, attn_mask: Tensor,
                 a_cfg, t_cfg, num_layers: int = 35, heads: int = 8, use_openmax: bool = False,
                 training: bool = True, drop: Optional["DropoutPlan"] = None):
    a_seq, a_mask = audio_encoder_forward(sds["audio_encoder"], waves, a_cfg)
    t_seq, t_mask = text_encoder_forward(sds["text_encoder"], ids, attn_mask, t_cfg)
    a_enh, t_enh = cross_attention_forward(sds["cross"], a_seq, t_seq, a_mask, t_mask, heads, drop)
    a_vec = pooling_forward(sds["pool_a"], a_enh, a_mask)
    t_vec = pooling_forward(sds["pool_t"], t_enh, t_mask)
    fused = fusion_forward(sds["fusion"], a_vec, t_vec, drop)
    logits, unc, anchor, feats = classifier_forward(sds["classifier"], fused, num_layers, use_openmax, training, drop)
    return dict(a_seq=a_seq, t_seq=t_seq, a_enh=a_enh, t_enh=t_enh, a_vec=a_vec, t_vec=t_vec,
                fused=fused, logits=logits, unc=unc, feats=feats)
