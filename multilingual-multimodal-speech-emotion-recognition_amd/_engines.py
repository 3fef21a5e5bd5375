"""Packed-weight engines for the two frozen encoders.

An engine owns the device-resident, kernel-ready form of a HuggingFace state dict (split-bf16
planes, conv taps in channels-last order, weight-norm folded, Q/K/V fused) plus the C structs that
`ser_wav2vec2_forward` / `ser_xlmr_forward` take, and a grow-only workspace.  Packing happens once
per `load_state_dict` (frozen encoders); the forward itself is one C call.
"""
import ctypes as C

import os

import torch

from . import _lib as L


class _Keep:
    """Holds references to every device tensor whose raw pointer sits in a C struct."""

    def __init__(self):
        self.t = []

    def f32(self, x, dev):
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        self.t.append(x)
        return x.data_ptr()

    def split(self, x, dev, want_lo, planar=False):
        """Kernel-ready weight [N, K]: the hi plane alone (one-product mode), or both planes — interleaved in groups of
        32 along K (what the three-product GEMM reads, csrc/ser_common.h) unless `planar` (positional conv)."""
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        if want_lo and not planar:
            w = L.split_bf16_il(x)
            self.t.append(w)
            hi, lo = L.il_ptrs(w)
            return L.SplitW(hi, lo)
        hi, lo = L.split_bf16(x, want_lo)
        self.t += [hi, lo]
        return L.SplitW(hi.data_ptr(), lo.data_ptr() if lo is not None else None)


def _pack_layer(keep, sd, names, dev, want_lo):
    g = lambda k: sd[k]
    lw = L.LayerW()
    qkv_w = torch.cat([g(names["q"] + ".weight"), g(names["k"] + ".weight"), g(names["v"] + ".weight")], dim=0)
    qkv_b = torch.cat([g(names["q"] + ".bias"), g(names["k"] + ".bias"), g(names["v"] + ".bias")], dim=0)
    lw.qkv = keep.split(qkv_w, dev, want_lo)
    lw.qkv_b = keep.f32(qkv_b, dev)
    lw.o = keep.split(g(names["o"] + ".weight"), dev, want_lo)
    lw.o_b = keep.f32(g(names["o"] + ".bias"), dev)
    lw.ln1_g = keep.f32(g(names["ln1"] + ".weight"), dev)
    lw.ln1_b = keep.f32(g(names["ln1"] + ".bias"), dev)
    lw.f1 = keep.split(g(names["f1"] + ".weight"), dev, want_lo)
    lw.f1_b = keep.f32(g(names["f1"] + ".bias"), dev)
    lw.f2 = keep.split(g(names["f2"] + ".weight"), dev, want_lo)
    lw.f2_b = keep.f32(g(names["f2"] + ".bias"), dev)
    lw.ln2_g = keep.f32(g(names["ln2"] + ".weight"), dev)
    lw.ln2_b = keep.f32(g(names["ln2"] + ".bias"), dev)
    return lw


_W2V = dict(q="attention.q_proj", k="attention.k_proj", v="attention.v_proj", o="attention.out_proj", ln1="layer_norm",
            f1="feed_forward.intermediate_dense", f2="feed_forward.output_dense", ln2="final_layer_norm")
_XLMR = dict(q="attention.self.query", k="attention.self.key", v="attention.self.value", o="attention.output.dense",
             ln1="attention.output.LayerNorm", f1="intermediate.dense", f2="output.dense", ln2="output.LayerNorm")


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


class _Workspace:
    def __init__(self):
        self.buf = None

    def get(self, nbytes, dev):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != dev:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        return self.buf


class Wav2Vec2Engine:
    """Kernel-ready Wav2Vec2 (hf wav2vec2/modeling_wav2vec2.py Wav2Vec2Model, eval forward)."""

    def __init__(self, hf_config, state_dict, device, prec=L.PREC_BF16X3):
        c = hf_config
        assert c.feat_extract_norm == "group" and not c.do_stable_layer_norm and not c.conv_bias, \
            "only the wav2vec2-base family (group-norm front end, post-LN encoder) is implemented"
        self.prec = prec
        self.device = torch.device(device)
        self.cfg = L.W2vConfig()
        self.cfg.hidden, self.cfg.layers, self.cfg.heads, self.cfg.ffn = (c.hidden_size, c.num_hidden_layers,
                                                                           c.num_attention_heads, c.intermediate_size)
        self.cfg.n_conv = len(c.conv_dim)
        for i in range(self.cfg.n_conv):
            self.cfg.conv_dim[i], self.cfg.conv_kernel[i], self.cfg.conv_stride[i] = (c.conv_dim[i], c.conv_kernel[i],
                                                                                     c.conv_stride[i])
        self.cfg.pos_kernel, self.cfg.pos_groups = c.num_conv_pos_embeddings, c.num_conv_pos_embedding_groups
        self.cfg.eps = c.layer_norm_eps
        self.hidden = c.hidden_size
        self.ws = _Workspace()
        self.pack(state_dict)

    def pack(self, sd):
        dev, want_lo = self.device, self.prec == L.PREC_BF16X3
        keep = _Keep()
        w = L.W2vWeights()
        n = self.cfg.n_conv
        w.conv0_w = keep.f32(sd["feature_extractor.conv_layers.0.conv.weight"].reshape(self.cfg.conv_dim[0], -1), dev)
        w.gn_g = keep.f32(sd["feature_extractor.conv_layers.0.layer_norm.weight"], dev)
        w.gn_b = keep.f32(sd["feature_extractor.conv_layers.0.layer_norm.bias"], dev)
        for i in range(1, n):
            cw = sd[f"feature_extractor.conv_layers.{i}.conv.weight"]          # [Cout, Cin, k]
            w.conv_w[i] = keep.split(cw.permute(0, 2, 1).reshape(cw.shape[0], -1), dev, want_lo)   # [Cout, k*Cin]
        w.fp_ln_g = keep.f32(sd["feature_projection.layer_norm.weight"], dev)
        w.fp_ln_b = keep.f32(sd["feature_projection.layer_norm.bias"], dev)
        w.fp_w = keep.split(sd["feature_projection.projection.weight"], dev, want_lo)
        w.fp_b = keep.f32(sd["feature_projection.projection.bias"], dev)
        # weight_norm(dim=2) folded once: W = g * v / ||v||_(0,1)   (hf :326-349); frozen => load-time constant
        g0 = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"].float()
        v0 = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"].float()
        pw = g0 * v0 / torch.sqrt((v0 * v0).sum(dim=(0, 1), keepdim=True))        # [H, Cg, K]
        G = self.cfg.pos_groups
        H, Cg, K = pw.shape
        pw = pw.reshape(G, H // G, Cg, K).permute(0, 1, 3, 2)                                   # [g][n][j][c]
        if want_lo:   # three-product mode: pad every tap's Cg channels to a multiple of 32 (zeros) so that the Toeplitz window
            Cgp = (Cg + L.IL_GROUP - 1) // L.IL_GROUP * L.IL_GROUP      # rows of the slab start on a 32-group: interleaved planes
            pw = torch.nn.functional.pad(pw, (0, Cgp - Cg)).reshape(G * (H // G), K * Cgp)
            w.pos_w = keep.split(pw, dev, True)
        else:
            w.pos_w = keep.split(pw.reshape(G * (H // G), K * Cg), dev, False)
        w.pos_b = keep.f32(sd["encoder.pos_conv_embed.conv.bias"], dev)
        w.enc_ln_g = keep.f32(sd["encoder.layer_norm.weight"], dev)
        w.enc_ln_b = keep.f32(sd["encoder.layer_norm.bias"], dev)
        self._layers = (L.LayerW * max(1, self.cfg.layers))()
        for i in range(self.cfg.layers):
            self._layers[i] = _pack_layer(keep, _sub(sd, f"encoder.layers.{i}."), _W2V, dev, want_lo)
        w.layers = C.cast(self._layers, C.POINTER(L.LayerW))
        self.w, self._keep = w, keep

    def out_len(self, T):
        return L.lib.ser_wav2vec2_out_len(C.byref(self.cfg), int(T))

    def forward(self, wave):
        """wave [B,T] fp32 raw clips on the device -> last_hidden_state [B,S,H] fp32."""
        assert wave.is_cuda and wave.dtype == torch.float32 and wave.dim() == 2
        wave = wave.contiguous()
        B, T = wave.shape
        S = self.out_len(T)
        if S <= 0:
            raise L.SerHipError(f"clip of {T} samples is shorter than the conv receptive field")
        nbytes = L.lib.ser_wav2vec2_workspace_bytes(C.byref(self.cfg), B, T, self.prec)
        if nbytes == 0:
            L.check(-1, "ser_wav2vec2_workspace_bytes")
        ws = self.ws.get(nbytes, wave.device)
        out = torch.empty(B, S, self.hidden, dtype=torch.float32, device=wave.device)
        L.check(L.lib.ser_wav2vec2_forward(C.byref(self.cfg), C.byref(self.w), wave.data_ptr(), B, T, self.prec,
                                           out.data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr()),
                "ser_wav2vec2_forward")
        return out


class XlmrEngine:
    """Kernel-ready XLM-RoBERTa (hf xlm_roberta/modeling_xlm_roberta.py XLMRobertaModel, eval forward)."""

    def __init__(self, hf_config, state_dict, device, prec=L.PREC_BF16X3):
        c = hf_config
        self.prec = prec
        self.device = torch.device(device)
        self.cfg = L.XlmrConfig(c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.intermediate_size,
                                c.vocab_size, c.max_position_embeddings, c.pad_token_id, c.layer_norm_eps)
        self.hidden = c.hidden_size
        self.ws = _Workspace()
        self.pack(state_dict)

    def pack(self, sd):
        dev, want_lo = self.device, self.prec == L.PREC_BF16X3
        keep = _Keep()
        w = L.XlmrWeights()
        w.word_emb = keep.f32(sd["embeddings.word_embeddings.weight"], dev)
        w.pos_emb = keep.f32(sd["embeddings.position_embeddings.weight"], dev)
        w.type_emb = keep.f32(sd["embeddings.token_type_embeddings.weight"], dev)
        w.emb_ln_g = keep.f32(sd["embeddings.LayerNorm.weight"], dev)
        w.emb_ln_b = keep.f32(sd["embeddings.LayerNorm.bias"], dev)
        self._layers = (L.LayerW * max(1, self.cfg.layers))()
        for i in range(self.cfg.layers):
            self._layers[i] = _pack_layer(keep, _sub(sd, f"encoder.layer.{i}."), _XLMR, dev, want_lo)
        w.layers = C.cast(self._layers, C.POINTER(L.LayerW))
        self.w, self._keep = w, keep

    def forward(self, ids, attn_mask):
        """ids [B,S] int64, attn_mask [B,S] (1/0) on the device -> last_hidden_state [B,S,H] fp32."""
        assert ids.is_cuda and ids.dtype == torch.int64
        ids = ids.contiguous()
        mask = attn_mask.to(torch.float32).contiguous()
        B, S = ids.shape
        nbytes = L.lib.ser_xlmr_workspace_bytes(C.byref(self.cfg), B, S, self.prec)
        ws = self.ws.get(nbytes, ids.device)
        out = torch.empty(B, S, self.hidden, dtype=torch.float32, device=ids.device)
        L.check(L.lib.ser_xlmr_forward(C.byref(self.cfg), C.byref(self.w), ids.data_ptr(), mask.data_ptr(), B, S,
                                       self.prec, out.data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr()),
                "ser_xlmr_forward")
        return out


# ---- one-time tile tuning -----------------------------------------------------------------------------------------
# At batch 16 an encoder GEMM is a few hundred tiles on 256 CUs, so its time is set by how the tile count divides
# into the resident-workgroup slots; which tile height wins depends on the exact (rows, N, K).  The first eager
# forward at a new batch geometry times the five heights on each of its GEMM shapes (~30 ms in total) and records
# the winners in the library (ser_gemm_tile_hint).  The arithmetic of a tile does not depend on its height, so this
# changes speed only.  Skipped during graph capture (the eager warm-up pass has already run it) and in bf16x3 mode.
_TUNED = set()
_TUNE_RANKED = {}      # (rows, N, K, three_products) -> [(ms, cfg), ...] of the stand-alone timing pass, fastest first
TILE_HEIGHTS = (64, 96, 128, 160, 192)
# interleaved three-product mode: also the single-LDS-buffer tiles (3000 + rows: three workgroups per CU) and the
# 512-thread tiles (csrc/ser_common.h SER_GEMM_CFG_*); which wins depends on the shape (scripts/gemm_il_probe.py --cfgs)
# experiments: SER_GEMM_SKIP_FAMILIES="7,3" leaves the 7xxx (64-column) and 3xxx (single-buffer) tiles out of the timing pass
_SKIP_CFG_FAMILIES = {int(x) for x in os.environ.get("SER_GEMM_SKIP_FAMILIES", "").split(",") if x.strip()}
TILE_CONFIGS_X3 = TILE_HEIGHTS + (3064, 3096, 3128, 1192, 1256, 5128, 6256, 7064, 7096, 7128, 7192)


def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def tune_gemm_shapes(shapes, device, reps=8, three_products=False, collective=False):
    """shapes: iterable of (rows_total, N, K) of bf16 NT GEMMs with N >= 128; `three_products`: the interleaved
    three-product mode (its own table: a k-tile carries 1.5x the MFMA work of the one-product kernel's).

    Data parallel (world > 1): timings differ from GPU to GPU, and a rank on a slower plan sets the pace of every step,
    so plans must be the same everywhere.  The LAZY call from a forward is then a no-op (ranks reach it with different
    batch shapes - ragged and partial batches - so it cannot be a collective; the library's cost model picks the tiles,
    identically on every rank).  `collective=True` is the explicit form for a point every rank reaches with the same
    shapes (PipelinedStepper's capture): all ranks time, rank 0's choices are broadcast and installed everywhere."""
    if torch.cuda.is_current_stream_capturing():
        return
    if os.environ.get("SER_GEMM_TUNE", "0") != "1" and not os.environ.get("SER_GEMM_FORCE_CFG"):
        # default: no timing pass - the library's deterministic rules pick the tiles (csrc/gemm_bf16.hip pick_bm): the same
        # kernels on every rank and in every run.  SER_GEMM_TUNE=1 turns the per-shape timing pass back on (experiments).
        return
    if _world() > 1 and not collective:
        return
    if os.environ.get("SER_GEMM_FORCE_CFG"):      # experiments: one tile configuration for every shape, no timing pass
        for rows, N, K in shapes:
            if N >= 128 and rows > 64:
                L.lib.ser_gemm_tile_hint_mode(int(rows), int(N), int(K), 1 if three_products else 0, int(os.environ["SER_GEMM_FORCE_CFG"]))
        return
    pm = 2 if three_products else 1
    fresh = []
    for rows, N, K in shapes:
        key = (int(rows), int(N), int(K), pm, torch.device(device).index)
        if key in _TUNED or N < 128 or rows <= 64:
            continue
        _TUNED.add(key)
        fresh.append((int(rows), int(N), int(K)))
        g = torch.Generator(device="cpu").manual_seed(0)     # random operands: zeros flatter the clock (DVFS)
        a = (torch.randn(rows, K * pm, generator=g) * 0.5).to(device=device, dtype=torch.bfloat16)
        w = (torch.randn(N, K * pm, generator=g) * 0.05).to(device=device, dtype=torch.bfloat16)
        c = torch.empty(rows, N * pm, dtype=torch.bfloat16, device=device)
        lo = (lambda t: t.data_ptr() + 2 * L.IL_GROUP) if three_products else (lambda t: None)
        best, best_ms, ranked = 0, float("inf"), []
        try:
            for bm in (TILE_CONFIGS_X3 if three_products else TILE_HEIGHTS):
                if bm in (1192, 1256, 5128) and N < 256:
                    continue
                if bm // 1000 in _SKIP_CFG_FAMILIES:
                    continue
                if 7000 <= bm < 8000 and (rows // 64) * (N // 64) > 2048:      # 64-column tiles: only where the 128-column tilings are short of workgroups
                    continue
                L.lib.ser_debug_set_gemm_bm(bm)

                def run():
                    L.check(L.lib.ser_gemm_bf16_nt(a.data_ptr(), lo(a), K, w.data_ptr(), lo(w), K, rows, N, K, None, L.ACT_NONE, None,
                                                   0, None, c.data_ptr(), lo(c), N, L.stream_ptr()), "ser_gemm_bf16_nt")
                run(); run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run()
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1)
                ranked.append((ms, bm))
                if ms < best_ms:
                    best, best_ms = bm, ms
        finally:
            L.lib.ser_debug_set_gemm_bm(0)
        L.lib.ser_gemm_tile_hint_mode(rows, N, K, 1 if three_products else 0, best)
        _TUNE_RANKED[(int(rows), int(N), int(K), bool(three_products))] = sorted(ranked)
    if collective:
        _share_plans(fresh, three_products)


def _share_plans(fresh, three_products):
    """Rank 0's choices for the freshly timed shapes, installed on every rank (results are bit-identical under any plan;
    this is about speed and about reproducible `gemm_plans` in the bench line).  Collective: every rank calls it with the
    same shapes in the same order."""
    import torch.distributed as dist
    if not fresh or _world() < 2:
        return
    mine = [(k, _TUNE_RANKED[k + (bool(three_products),)][0][1]) for k in fresh]
    box = [mine]
    dist.broadcast_object_list(box, src=0)
    for (rows, N, K), cfg in box[0]:
        L.lib.ser_gemm_tile_hint_mode(rows, N, K, 1 if three_products else 0, int(cfg))
        ranked = _TUNE_RANKED.get((rows, N, K, bool(three_products)))
        if ranked is not None and ranked[0][1] != cfg:      # keep the local table consistent with what is installed
            _TUNE_RANKED[(rows, N, K, bool(three_products))] = sorted(ranked, key=lambda r: (r[1] != cfg, r[0]))


def close_runner_ups(margin=1.06, limit=2):
    """Shapes with other configurations within `margin` of the fastest in the stand-alone timing pass:
    [((rows, N, K, three_products), best_cfg, [up to `limit` alternatives, fastest first])].  Stand-alone times this close
    do not decide which one is faster beside the head graph (PipelinedStepper.refine_gemm_plans measures that)."""
    out = []
    for key, ranked in _TUNE_RANKED.items():
        alts = [cfg for ms, cfg in ranked[1:1 + limit] if ms <= margin * ranked[0][0]]
        if alts:
            out.append((key, ranked[0][1], alts))
    return out


def set_plan(key, cfg):
    rows, N, K, three = key
    L.lib.ser_gemm_tile_hint_mode(rows, N, K, 1 if three else 0, int(cfg))


def _w2v_gemm_shapes(cfg, B, T, extra_rows=0):
    lens, t = [], T
    for i in range(cfg.n_conv):
        t = (t - cfg.conv_kernel[i]) // cfg.conv_stride[i] + 1
        lens.append(t)
    H, F = cfg.hidden, cfg.ffn
    shapes = [(B * lens[i], cfg.conv_dim[i], cfg.conv_kernel[i] * cfg.conv_dim[i - 1]) for i in range(1, cfg.n_conv)]
    rows = B * lens[-1]
    shapes.append((rows, H, cfg.conv_dim[cfg.n_conv - 1]))
    rows += extra_rows
    shapes += [(rows, 3 * H, H), (rows, H, H), (rows, F, H), (rows, H, F)]
    return shapes


def tune_pair(a, t, B, T, Bt, St, collective=False):
    """Timing pass for the GEMM shapes of one paired encoder call (B clips of T samples, Bt x St tokens)."""
    if a.cfg.layers == t.cfg.layers and a.cfg.hidden == t.cfg.hidden:
        tune_gemm_shapes(_w2v_gemm_shapes(a.cfg, B, T, extra_rows=Bt * St), a.device, three_products=a.prec == L.PREC_BF16X3,
                         collective=collective)


def forward_pair(audio_engine, text_engine, wave, ids, attn_mask, slot=0):
    """Both frozen encoders in ONE call on the current stream (ser_encoders_forward): when the two models have the
    same depth their layers run in lock-step with one launch per step for both.  -> (a_enc [B,S_a,H], t_enc [B,S_t,H])."""
    a, t = audio_engine, text_engine
    assert a.prec == t.prec, "both encoders must run in the same precision mode"
    assert wave.is_cuda and wave.dtype == torch.float32 and wave.dim() == 2
    assert ids.is_cuda and ids.dtype == torch.int64
    wave, ids = wave.contiguous(), ids.contiguous()
    mask = attn_mask.to(torch.float32).contiguous()
    B, T = wave.shape
    Bt, St = ids.shape
    Sa = a.out_len(T)
    if Sa <= 0:
        raise L.SerHipError(f"clip of {T} samples is shorter than the conv receptive field")
    na = L.lib.ser_wav2vec2_workspace_bytes(C.byref(a.cfg), B, T, a.prec)
    nt = L.lib.ser_xlmr_workspace_bytes(C.byref(t.cfg), Bt, St, t.prec)
    if na == 0 or nt == 0:
        L.check(-1, "ser_*_workspace_bytes")
    # slot > 0: a second workspace pair, for an encoder pass that runs beside another one (PipelinedStepper depth 2)
    wsa = (a.ws if slot == 0 else a.__dict__.setdefault('_ws_slots', {}).setdefault(slot, _Workspace())).get(na, wave.device)
    wst = (t.ws if slot == 0 else t.__dict__.setdefault('_ws_slots', {}).setdefault(slot, _Workspace())).get(nt, ids.device)
    tune_pair(a, t, B, T, Bt, St)
    out_a = torch.empty(B, Sa, a.hidden, dtype=torch.float32, device=wave.device)
    out_t = torch.empty(Bt, St, t.hidden, dtype=torch.float32, device=ids.device)
    L.check(L.lib.ser_encoders_forward(C.byref(a.cfg), C.byref(a.w), wave.data_ptr(), B, T, C.byref(t.cfg), C.byref(t.w),
                                       ids.data_ptr(), mask.data_ptr(), Bt, St, a.prec, out_a.data_ptr(), out_t.data_ptr(),
                                       wsa.data_ptr(), wsa.numel(), wst.data_ptr(), wst.numel(), L.stream_ptr()),
            "ser_encoders_forward")
    return out_a, out_t
