// Module-level forward of the two frozen encoders: host-side orchestration of the HIP kernels.
// One C call launches the whole Wav2Vec2 / XLM-R forward on the caller's stream (no allocation,
// no synchronisation, so the call can be captured into a hipGraph).
#include "ser_common.h"

namespace {

struct Planes {
  bf16_t* hi;
  bf16_t* lo;
};

static Planes take_planes(SerArena& ar, size_t n, bool x3) {
  Planes p;
  p.hi = ar.get<bf16_t>(n);
  p.lo = x3 ? ar.get<bf16_t>(n) : nullptr;
  return p;
}

static SerGemmArgs gemm_args(Planes a, int lda, SerSplitW w, int ldw, int M, int N, int K) {
  SerGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_hi = a.hi; g.a_lo = a.lo;
  g.w_hi = w.hi; g.w_lo = w.lo;
  if (!g.a_lo || !g.w_lo) { g.a_lo = nullptr; g.w_lo = nullptr; }
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw;
  g.nb1 = 1; g.nb2 = 1;
  return g;
}

struct LayerBufs {
  Planes qkv, ctx, h1p, ffn;
  float *t1, *h1, *t2;
};

static void alloc_layer_bufs(SerArena& ar, LayerBufs& lb, size_t rows, int H, int F, bool x3) {
  lb.qkv = take_planes(ar, rows * 3 * H, x3);
  lb.ctx = take_planes(ar, rows * H, x3);
  lb.h1p = take_planes(ar, rows * H, x3);
  lb.ffn = take_planes(ar, rows * F, x3);
  lb.t1 = ar.get<float>(rows * H);
  lb.h1 = ar.get<float>(rows * H);
  lb.t2 = ar.get<float>(rows * H);
}

// Post-LN transformer block (hf wav2vec2 :591-608 / xlm_roberta :421-463), eval mode.
// in: h (fp32) + hp (planes).  out: out_f32 (+ out planes when given).
static int run_layer(const SerLayerW& w, const float* h, Planes hp, const float* key_mask, int B, int S, int H, int F,
                     int heads, float eps, LayerBufs& lb, float* out_f32, Planes outp, hipStream_t st) {
  const int rows = B * S;
  SerGemmArgs g = gemm_args(hp, H, w.qkv, H, rows, 3 * H, H);
  g.bias = w.qkv_b; g.c_hi = lb.qkv.hi; g.c_lo = lb.qkv.lo; g.ldc = 3 * H;
  SER_TRY(ser_launch_gemm_bf16(g, st));
  SER_TRY(ser_launch_self_attention(lb.qkv.hi, lb.qkv.lo, key_mask, B, S, heads, lb.ctx.hi, lb.ctx.lo, st));
  g = gemm_args(lb.ctx, H, w.o, H, rows, H, H);
  g.bias = w.o_b; g.residual = h; g.ldr = H; g.c_f32 = lb.t1; g.ldc = H;
  SER_TRY(ser_launch_gemm_bf16(g, st));
  SER_TRY(ser_launch_layernorm(lb.t1, nullptr, w.ln1_g, w.ln1_b, eps, rows, H, lb.h1, lb.h1p.hi, lb.h1p.lo, st));
  g = gemm_args(lb.h1p, H, w.f1, H, rows, F, H);
  g.bias = w.f1_b; g.act = SER_ACT_GELU; g.c_hi = lb.ffn.hi; g.c_lo = lb.ffn.lo; g.ldc = F;
  SER_TRY(ser_launch_gemm_bf16(g, st));
  g = gemm_args(lb.ffn, F, w.f2, F, rows, H, F);
  g.bias = w.f2_b; g.residual = lb.h1; g.ldr = H; g.c_f32 = lb.t2; g.ldc = H;
  SER_TRY(ser_launch_gemm_bf16(g, st));
  SER_TRY(ser_launch_layernorm(lb.t2, nullptr, w.ln2_g, w.ln2_b, eps, rows, H, out_f32, outp.hi, outp.lo, st));
  return SER_OK;
}

static int w2v_lengths(const SerW2vConfig* c, int T, int* L) {
  int len = T;
  for (int i = 0; i < c->n_conv; ++i) {
    if (len < c->conv_kernel[i]) return -1;
    len = (len - c->conv_kernel[i]) / c->conv_stride[i] + 1;
    L[i] = len;
  }
  return len;
}

static int check_w2v_cfg(const SerW2vConfig* c) {
  SER_REQUIRE(c && c->n_conv >= 2 && c->n_conv <= SER_MAX_CONV, "wav2vec2: n_conv out of range");
  SER_REQUIRE(c->hidden == c->heads * 64, "wav2vec2: head_dim must be 64 (hidden=%d heads=%d)", c->hidden, c->heads);
  SER_REQUIRE(c->hidden % 64 == 0 && c->ffn % 64 == 0, "wav2vec2: hidden/ffn must be multiples of 64");
  SER_REQUIRE(c->hidden % c->pos_groups == 0 && (c->hidden / c->pos_groups) % 8 == 0,
              "wav2vec2: channels per positional-conv group must be a multiple of 8");
  SER_REQUIRE(((c->hidden / c->pos_groups) * c->pos_kernel) % 64 == 0, "wav2vec2: pos-conv K must be a multiple of 64");
  for (int i = 1; i < c->n_conv; ++i)
    SER_REQUIRE((c->conv_kernel[i] * c->conv_dim[i - 1]) % 64 == 0 && c->conv_dim[i - 1] % 8 == 0,
                "wav2vec2: conv layer %d K must be a multiple of 64", i);
  SER_REQUIRE(c->conv_dim[c->n_conv - 1] % 4 == 0 && c->conv_dim[c->n_conv - 1] <= 1024, "wav2vec2: conv_dim too large");
  return SER_OK;
}

// one pass over the arena; with ar.base == nullptr only the size is computed
static int w2v_run(const SerW2vConfig* c, const SerW2vWeights* w, const float* wave, int B, int T, int prec, float* out,
                   SerArena& ar, hipStream_t st, bool dry) {
  const bool x3 = prec == SER_PREC_BF16X3;
  int L[SER_MAX_CONV];
  const int S = w2v_lengths(c, T, L);
  SER_REQUIRE(S > 0, "wav2vec2: clip of %d samples is shorter than the receptive field", T);
  const int H = c->hidden, F = c->ffn, G = c->pos_groups, Cg = H / G, Kp = c->pos_kernel;
  const int nc = c->n_conv, Cl = c->conv_dim[nc - 1];
  const size_t rows = (size_t)B * S;

  void* c0_scratch = ar.take(ser_conv0_scratch_bytes(B, L[0], c->conv_dim[0]));
  Planes xa = take_planes(ar, (size_t)B * L[0] * c->conv_dim[0], x3);
  Planes xb = take_planes(ar, (size_t)B * L[1] * c->conv_dim[1], x3);
  float* feat = ar.get<float>(rows * Cl);
  Planes featn = take_planes(ar, rows * Cl, x3);
  float* z = ar.get<float>(rows * H);
  Planes slab = take_planes(ar, (size_t)B * G * (S + Kp - 1) * Cg, x3);
  float* hsum = ar.get<float>(rows * H);
  float* ha = ar.get<float>(rows * H);
  float* hb = ar.get<float>(rows * H);
  Planes hpa = take_planes(ar, rows * H, x3);
  Planes hpb = take_planes(ar, rows * H, x3);
  LayerBufs lb;
  alloc_layer_bufs(ar, lb, rows, H, F, x3);
  if (dry) return SER_OK;
  if (ar.base && ar.off > ar.cap) {
    ser_set_error("wav2vec2: workspace too small (%zu needed, %zu given)", ar.off, ar.cap);
    return SER_E_WORKSPACE;
  }

  // conv0 + GroupNorm + GELU, fused with the clip normalisation
  SER_TRY(ser_launch_conv0(wave, B, T, w->conv0_w, w->gn_g, w->gn_b, c->conv_dim[0], c->conv_kernel[0],
                           c->conv_stride[0], L[0], xa.hi, xa.lo, c0_scratch, st));
  // conv1..: channels-last strided-row GEMMs (no im2col): A row t = x[t*stride*C : +k*C]
  Planes cur = xa, nxt = xb;
  for (int i = 1; i < nc; ++i) {
    const int Cin = c->conv_dim[i - 1], Cout = c->conv_dim[i], k = c->conv_kernel[i], s = c->conv_stride[i];
    SerGemmArgs g = gemm_args(cur, s * Cin, w->conv_w[i], k * Cin, L[i], Cout, k * Cin);
    g.nb1 = B;
    g.sa1 = (long long)L[i - 1] * Cin;
    g.sc1 = (long long)L[i] * Cout;
    g.act = SER_ACT_GELU;
    g.ldc = Cout;
    if (i == nc - 1) g.c_f32 = feat; else { g.c_hi = nxt.hi; g.c_lo = nxt.lo; }
    SER_TRY(ser_launch_gemm_bf16(g, st));
    Planes t = cur; cur = nxt; nxt = t;
  }
  // feature projection: LN(C) -> Linear(C -> H)   (hf :429-435)
  SER_TRY(ser_launch_layernorm(feat, nullptr, w->fp_ln_g, w->fp_ln_b, c->eps, (int)rows, Cl, nullptr, featn.hi, featn.lo, st));
  {
    SerGemmArgs g = gemm_args(featn, Cl, w->fp_w, Cl, (int)rows, H, Cl);
    g.bias = w->fp_b; g.c_f32 = z; g.ldc = H;
    SER_TRY(ser_launch_gemm_bf16(g, st));
  }
  // positional conv embedding as a (clip, group)-batched sliding-window GEMM + GELU + residual
  SER_TRY(ser_launch_posconv_slab(z, B, S, H, G, Kp, slab.hi, slab.lo, st));
  {
    const long long R = S + Kp - 1;
    SerGemmArgs g = gemm_args(slab, Cg, w->pos_w, Kp * Cg, S, Cg, Kp * Cg);
    g.nb1 = B; g.nb2 = G;
    g.sa1 = (long long)G * R * Cg; g.sa2 = R * Cg;
    g.sw1 = 0; g.sw2 = (long long)Cg * Kp * Cg;
    g.bias = w->pos_b; g.sbias1 = 0; g.sbias2 = Cg;
    g.act = SER_ACT_GELU;
    g.residual = z; g.ldr = H; g.sr1 = (long long)S * H; g.sr2 = Cg;
    g.c_f32 = hsum; g.ldc = H; g.sc1 = (long long)S * H; g.sc2 = Cg;
    SER_TRY(ser_launch_gemm_bf16(g, st));
  }
  SER_TRY(ser_launch_layernorm(hsum, nullptr, w->enc_ln_g, w->enc_ln_b, c->eps, (int)rows, H, ha, hpa.hi, hpa.lo, st));
  float* hin = ha; float* hout = hb;
  Planes pin = hpa, pout = hpb;
  for (int l = 0; l < c->layers; ++l) {
    const bool last = l == c->layers - 1;
    Planes none = {nullptr, nullptr};
    SER_TRY(run_layer(w->layers[l], hin, pin, nullptr, B, S, H, F, c->heads, c->eps, lb, last ? out : hout,
                      last ? none : pout, st));
    float* tf = hin; hin = hout; hout = tf;
    Planes tp = pin; pin = pout; pout = tp;
  }
  if (c->layers == 0) SER_CHECK_HIP(hipMemcpyAsync(out, ha, rows * H * sizeof(float), hipMemcpyDeviceToDevice, st));
  return SER_OK;
}

static int xlmr_run(const SerXlmrConfig* c, const SerXlmrWeights* w, const int64_t* ids, const float* mask, int B, int S,
                    int prec, float* out, SerArena& ar, hipStream_t st, bool dry) {
  const bool x3 = prec == SER_PREC_BF16X3;
  const int H = c->hidden, F = c->ffn;
  const size_t rows = (size_t)B * S;
  int* pos = ar.get<int>(rows);
  float* ha = ar.get<float>(rows * H);
  float* hb = ar.get<float>(rows * H);
  Planes hpa = take_planes(ar, rows * H, x3);
  Planes hpb = take_planes(ar, rows * H, x3);
  LayerBufs lb;
  alloc_layer_bufs(ar, lb, rows, H, F, x3);
  if (dry) return SER_OK;
  if (ar.base && ar.off > ar.cap) {
    ser_set_error("xlmr: workspace too small (%zu needed, %zu given)", ar.off, ar.cap);
    return SER_E_WORKSPACE;
  }
  SER_TRY(ser_launch_xlmr_embed(ids, B, S, w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g, w->emb_ln_b, c->eps, H,
                                c->vocab, c->max_pos, c->pad_id, pos, ha, hpa.hi, hpa.lo, st));
  float* hin = ha; float* hout = hb;
  Planes pin = hpa, pout = hpb;
  for (int l = 0; l < c->layers; ++l) {
    const bool last = l == c->layers - 1;
    Planes none = {nullptr, nullptr};
    SER_TRY(run_layer(w->layers[l], hin, pin, mask, B, S, H, F, c->heads, c->eps, lb, last ? out : hout,
                      last ? none : pout, st));
    float* tf = hin; hin = hout; hout = tf;
    Planes tp = pin; pin = pout; pout = tp;
  }
  if (c->layers == 0) SER_CHECK_HIP(hipMemcpyAsync(out, ha, rows * H * sizeof(float), hipMemcpyDeviceToDevice, st));
  return SER_OK;
}

}  // namespace

extern "C" int ser_wav2vec2_out_len(const SerW2vConfig* cfg, int T) {
  int L[SER_MAX_CONV];
  if (!cfg || cfg->n_conv < 1 || cfg->n_conv > SER_MAX_CONV) return -1;
  return w2v_lengths(cfg, T, L);
}

extern "C" size_t ser_wav2vec2_workspace_bytes(const SerW2vConfig* cfg, int B, int T, int prec) {
  if (check_w2v_cfg(cfg) != SER_OK || B <= 0) return 0;
  SerArena ar(nullptr, 0);
  if (w2v_run(cfg, nullptr, nullptr, B, T, prec, nullptr, ar, nullptr, true) != SER_OK) return 0;
  return ar.off + 256;
}

extern "C" int ser_wav2vec2_forward(const SerW2vConfig* cfg, const SerW2vWeights* w, const float* wave, int B, int T,
                                    int prec, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  SER_TRY(check_w2v_cfg(cfg));
  SER_REQUIRE(w && wave && out && workspace && B > 0, "wav2vec2_forward: null argument");
  SER_REQUIRE(prec == SER_PREC_BF16 || prec == SER_PREC_BF16X3, "wav2vec2_forward: bad precision mode %d", prec);
  SerArena ar(workspace, workspace_bytes);
  return w2v_run(cfg, w, wave, B, T, prec, out, ar, (hipStream_t)stream, false);
}

extern "C" size_t ser_xlmr_workspace_bytes(const SerXlmrConfig* cfg, int B, int S, int prec) {
  if (!cfg || B <= 0 || S <= 0) return 0;
  SerArena ar(nullptr, 0);
  if (xlmr_run(cfg, nullptr, nullptr, nullptr, B, S, prec, nullptr, ar, nullptr, true) != SER_OK) return 0;
  return ar.off + 256;
}

extern "C" int ser_xlmr_forward(const SerXlmrConfig* cfg, const SerXlmrWeights* w, const int64_t* ids,
                                const float* attn_mask, int B, int S, int prec, float* out, void* workspace,
                                size_t workspace_bytes, void* stream) {
  SER_REQUIRE(cfg && w && ids && out && workspace && B > 0 && S > 0, "xlmr_forward: null argument");
  SER_REQUIRE(cfg->hidden == cfg->heads * 64 && cfg->hidden % 64 == 0 && cfg->ffn % 64 == 0,
              "xlmr: head_dim must be 64 and hidden/ffn multiples of 64");
  SER_REQUIRE(prec == SER_PREC_BF16 || prec == SER_PREC_BF16X3, "xlmr_forward: bad precision mode %d", prec);
  SerArena ar(workspace, workspace_bytes);
  return xlmr_run(cfg, w, ids, attn_mask, B, S, prec, out, ar, (hipStream_t)stream, false);
}
