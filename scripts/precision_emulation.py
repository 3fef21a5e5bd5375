#!/usr/bin/env python3
"""CPU prediction of the logits error of each encoder precision mode (TEST / ANALYSIS TOOL: it drives the oracle).

The frozen encoders of the HIP path round the OPERANDS of every MFMA product (Linear, conv-as-GEMM, positional conv,
QK^T and PV) to a 16-bit format and accumulate in fp32; residual stream, LayerNorm, softmax, GELU stay fp32.  This
script restates that on the CPU oracle by rounding the operands of the same products, on INITIAL weights at the
benchmark's own size, and prints the logits max-abs-err against the unrounded oracle.  It is how the precision mode of
bench.py was chosen: the cheapest mode whose error stays below the 1e-3 north-star tolerance.

    python scripts/precision_emulation.py --modes bf16 fp16 --batch 4 --seeds 0 1
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.nn.functional as F

import bench
import __graft_entry__ as ge
from oracle import ser_oracle as O


def rounder(fmt):
    if fmt == "fp32":
        return lambda x: x
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[fmt]
    return lambda x: x.to(dt).to(torch.float32)


class Emulate:
    """Context: inside it, the encoder products of the oracle see rounded operands.  `sites` maps a product class
    (conv, featproj, posconv, qkv, attn, out, ffn1, ffn2) to a format; anything missing uses `default`."""

    def __init__(self, default, sites=None):
        self.default, self.sites = default, dict(sites or {})

    def r(self, site):
        return rounder(self.sites.get(site, self.default))

    def __enter__(self):
        self.saved = (O.wav2vec2_features, O.wav2vec2_forward, O.transformer_layer_postln)
        emu = self

        def features(sd, x, cfg):
            h = x[:, None, :]
            for i, (k, s) in enumerate(zip(cfg["conv_kernel"], cfg["conv_stride"])):
                w = sd[f"feature_extractor.conv_layers.{i}.conv.weight"]
                if i == 0:        # conv0 runs in fp32 FMAs on the device (10 taps): no operand rounding
                    h = F.conv1d(h, w, None, stride=s)
                    g, b = sd["feature_extractor.conv_layers.0.layer_norm.weight"], sd["feature_extractor.conv_layers.0.layer_norm.bias"]
                    mu = h.mean(dim=2, keepdim=True)
                    var = ((h - mu) ** 2).mean(dim=2, keepdim=True)
                    h = (h - mu) / torch.sqrt(var + 1e-5) * g[None, :, None] + b[None, :, None]
                else:
                    rr = emu.r("conv")
                    h = F.conv1d(rr(h), rr(w), None, stride=s)
                h = O.gelu(h)
            return h.transpose(1, 2)

        def layer(h, p, heads, eps, names, key_bias):
            D = h.shape[-1]
            scale = (D // heads) ** -0.5
            rq, ra, ro, r1, r2 = emu.r("qkv"), emu.r("attn"), emu.r("out"), emu.r("ffn1"), emu.r("ffn2")
            lin = lambda x, n, rr: O.linear(rr(x), rr(p[names[n] + ".weight"]), p[names[n] + ".bias"])
            q, k, v = lin(h, "q", rq), lin(h, "k", rq), lin(h, "v", rq)
            B, S, _ = q.shape
            hd = D // heads
            qh = ra(q).view(B, S, heads, hd).transpose(1, 2)
            kh = ra(k).view(B, S, heads, hd).transpose(1, 2)
            vh = ra(v).view(B, S, heads, hd).transpose(1, 2)
            s_ = (qh @ kh.transpose(2, 3)) * scale
            if key_bias is not None:
                s_ = s_ + key_bias[:, None, None, :]
            pr = torch.softmax(s_, dim=-1)
            ctx = (ra(pr) @ vh).transpose(1, 2).reshape(B, S, D)
            a = lin(ctx, "o", ro)
            h = O.layer_norm(h + a, p[names["ln1"] + ".weight"], p[names["ln1"] + ".bias"], eps)
            f = O.gelu(lin(h, "f1", r1))
            f = lin(f, "f2", r2)
            return O.layer_norm(h + f, p[names["ln2"] + ".weight"], p[names["ln2"] + ".bias"], eps)

        def w2v_forward(sd, x, cfg):
            eps = cfg["eps"]
            feats = O.wav2vec2_features(sd, x, cfg)
            e = O.layer_norm(feats, sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"], eps)
            rf, rp = emu.r("featproj"), emu.r("posconv")
            z = O.linear(rf(e), rf(sd["feature_projection.projection.weight"]), sd["feature_projection.projection.bias"])
            W = O.wav2vec2_pos_conv_weight(sd)
            K = cfg["pos_kernel"]
            pc = F.conv1d(rp(z).transpose(1, 2), rp(W), sd["encoder.pos_conv_embed.conv.bias"], padding=K // 2, groups=cfg["pos_groups"])
            if K % 2 == 0:
                pc = pc[:, :, :-1]
            h = z + O.gelu(pc).transpose(1, 2)
            h = O.layer_norm(h, sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], eps)
            for i in range(cfg["layers"]):
                h = O.transformer_layer_postln(h, O.sub(sd, f"encoder.layers.{i}."), cfg["heads"], eps, O.W2V_NAMES, None)
            return h

        O.wav2vec2_features, O.wav2vec2_forward, O.transformer_layer_postln = features, w2v_forward, layer
        return self

    def __exit__(self, *a):
        O.wav2vec2_features, O.wav2vec2_forward, O.transformer_layer_postln = self.saved


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", nargs="+", default=["bf16", "fp16"])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--tokens", type=int, default=32)
    ap.add_argument("--seeds", type=int, nargs="+", default=[4321])
    ap.add_argument("--weight-seed", type=int, default=0)
    ap.add_argument("--site", action="append", default=[], help="site=fmt override, e.g. conv=fp32 (applies to every mode)")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--ablate", default=None, help="format given in turn to each single product class (others keep the mode)")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    torch.manual_seed(args.weight_seed)
    sysm, wc, xc = bench.build_system("bf16", "cpu")
    if args.weight_seed != 0:
        pass
    sds = {k: {n: v.detach() for n, v in getattr(sysm, k).state_dict().items()} for k in sysm.CKPT_KEYS}
    a_cfg, t_cfg = ge.oracle_cfgs(wc, xc)
    sites = dict(s.split("=") for s in args.site)
    for seed in args.seeds:
        wave, ids, mask, labels = bench.synth_batch(args.batch, args.seconds, args.tokens, xc.vocab_size, 4, seed)
        with torch.no_grad():
            ref = O.full_forward(sds, list(wave), ids, mask, a_cfg, t_cfg, use_openmax=False, training=True)
            spread = (ref["logits"].max(0).values - ref["logits"].min(0).values).max().item()
            print(f"seed {seed}: oracle logits spread across clips {spread:.3e}")
            runs = [(m, sites) for m in args.modes]
            if args.ablate:
                runs += [(m, dict(sites, **{st: args.ablate})) for m in args.modes
                         for st in ("conv", "featproj", "posconv", "qkv", "attn", "out", "ffn1", "ffn2")]
            for mode, sites in runs:
                with Emulate(mode, sites):
                    out = O.full_forward(sds, list(wave), ids, mask, a_cfg, t_cfg, use_openmax=False, training=True)
                err = (out["logits"] - ref["logits"]).abs().max().item()
                ea = (out["a_seq"] - ref["a_seq"]).abs().max().item()
                et = (out["t_seq"] - ref["t_seq"]).abs().max().item()
                same = bool(torch.equal(out["logits"].argmax(1), ref["logits"].argmax(1)))
                print(f"  {mode:6s} sites={sites}: logits err {err:.3e}  a_seq err {ea:.3e}  t_seq err {et:.3e}  argmax equal {same}", flush=True)


if __name__ == "__main__":
    main()
