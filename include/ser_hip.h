/* libser_hip.so — C ABI of the MI355X (gfx950) multimodal-SER hot path.
 *
 * The reference (kananmittal/Multilingual-Multimodal-Speech-Emotion-Recognition) has no FFI seam:
 * its hot path is the Python module API under src/models/ executed by torch / transformers.  This
 * header is the seam a maintainer binds instead (ctypes stub shown in INTEGRATION.md).  Every entry
 * point names the reference interface it replaces (paths relative to the reference root; "hf:" =
 * transformers 5.15.0 models/).
 *
 * Conventions
 *   - plain C: raw device pointers into caller-owned (torch-owned) storage, sizes, strides.
 *   - the library never allocates or frees device memory; scratch is a caller-provided workspace
 *     sized by the matching *_workspace_bytes() call.
 *   - every launch is asynchronous on the hipStream_t passed as `void* stream`.
 *   - return 0 (SER_OK) or a negative SER_E_* code; ser_last_error_string() gives the text
 *     (thread-local).
 *   - "split bf16": a tensor stored as two bf16 planes hi = bf16(x), lo = bf16(x - hi).  With both
 *     planes a product costs three bf16 MFMAs and carries ~2^-16 relative error (parity mode); with
 *     lo == NULL it is one plain bf16 MFMA (fast mode).
 */
#ifndef SER_HIP_H
#define SER_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SER_OK 0
#define SER_E_ARG (-1)
#define SER_E_HIP (-2)
#define SER_E_WORKSPACE (-3)

#define SER_ACT_NONE 0
#define SER_ACT_GELU 1 /* exact erf GELU (hf: activations.py GELUActivation) */
#define SER_ACT_RELU 2
#define SER_ACT_TANH 3
#define SER_ACT_SIGMOID 4

#define SER_PREC_BF16 0   /* one bf16 MFMA per product, fp32 accumulate            */
#define SER_PREC_BF16X3 1 /* split-bf16 operands, three MFMAs, ~fp32-accurate       */

#define SER_MAX_CONV 8

const char* ser_last_error_string(void);
int ser_abi_version(void);

/* HIP stream restricted to the compute units whose bits are set in mask[words] (hipExtStreamCreateWithCUMask). */
int ser_stream_create_cu_masked(const uint32_t* mask, int words, void** stream_out);
int ser_stream_destroy(void* stream);

/* ---------------------------------------------------------------------------------------------
 * op-level entry points (used by the parity tests and composed by the module-level calls)
 * ------------------------------------------------------------------------------------------- */

/* Split planes come in two layouts.  PLANAR: hi[n] and lo[n] are separate arrays.  INTERLEAVED: one array of 2 n
 * elements in which every run of 32 hi values is followed by its 32 lo values ([hi 0..31 | lo 0..31 | hi 32..63 | ...]);
 * a 128-byte line of a K-contiguous row then holds both planes of 32 values of k, which lets the three-product GEMM
 * stage exactly the bytes of a one-product k-tile.  An entry point recognises the interleaved layout by
 * lo == hi + 32 (elements): pass the array as `hi` and `hi + 32` as `lo`.  Row lengths must be multiples of 32.
 * Every (hi, lo) argument of the encoder entry points below accepts both layouts; A and W of a GEMM must agree. */

/* x[n] fp32 -> hi[n], lo[n] bf16 planes (lo may be NULL; interleaved when lo == hi + 32, then n % 32 == 0). */
int ser_split_bf16(const float* x, uint16_t* hi, uint16_t* lo, long long n, void* stream);
/* x [R, C] fp32 (row stride ldx) -> split planes of x^T: C rows of Rp >= R values, zero beyond R (Rp % 32 == 0); interleaved
 * when lo == hi + 32, the hi plane alone when lo == NULL.  Gives the K-contiguous NT GEMM the transposed operands of a Linear
 * layer's backward products (dx = dy W, dW = dy^T x) on the fine-tuning path (BASELINE config 3). */
int ser_split_bf16_t(const float* x, int R, int C, long long ldx, uint16_t* hi, uint16_t* lo, int Rp, void* stream);
/* ... both operand forms in one pass over x: planes of x [R, C] (s_hi / s_lo, C % 32 == 0) and of x^T [C, Rp] (t_hi / t_lo);
 * a null lo pointer = that form's hi plane alone (one-product mode).  Used by the fine-tuning Linear layers, whose forward
 * product reads x and W straight and whose backward products read them (and dy) transposed. */
int ser_split_bf16_both(const float* x, int R, int C, long long ldx, uint16_t* s_hi, uint16_t* s_lo, uint16_t* t_hi, uint16_t* t_lo,
                        int Rp, void* stream);
/* ... and, in the same pass, the column sums of every block of 32 rows: colpart[Rp / 32][C] (rows added in increasing order).
 * ser_colsum over colpart (M = Rp / 32) finishes the column sum of x - the bias gradient of a Linear layer whose dy is being
 * split anyway - in a fixed order, without another pass over x. */
int ser_split_bf16_both_colsum(const float* x, int R, int C, long long ldx, uint16_t* s_hi, uint16_t* s_lo, uint16_t* t_hi,
                               uint16_t* t_lo, int Rp, float* colpart, void* stream);
/* ... for many matrices in one launch: `table` = nprob descriptors in device memory, 16 64-bit words each
 * {x, s_hi, s_lo, t_hi, t_lo, R, C, Rp, t_roff, ldx, cover, bias_src, bias_dst, bias_n, blk0, cblocks}: a matrix may be a row block of
 * a vertically fused operand (q | k | v weights: straight planes from its own first row, transposed planes at row offset t_roff of
 * the fused operand's Rp-long rows); cover = rows walked (R, or a stand-alone matrix's padded Rp: zero-filled); optional bias
 * copy; blk0 = the matrix's first workgroup in the launch (ascending from 0), cblocks = C / 32; total_blocks = the sum of
 * (cover / 32) x cblocks.  The fine-tuning encoders refresh every Linear weight's operand planes with it once per step. */
int ser_split_bf16_both_multi(const void* table, int nprob, long long total_blocks, void* stream);

/* C[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual, split-bf16 operands, fp32 accumulate.
 * Replaces every torch.nn.Linear / Conv1d-as-GEMM inside the frozen encoders
 * (hf: wav2vec2/modeling_wav2vec2.py:254-272,429-435,500-572; xlm_roberta/modeling_xlm_roberta.py:211-250,336-398).
 * K % 64 == 0, lda/ldw % 8 == 0 (interleaved operands: K, lda, ldw % 32 == 0; lda/ldw/ldc count LOGICAL elements).
 * Any of c_f32 / (c_hi,c_lo) may be NULL; (c_hi, c_lo = c_hi + 32) writes the interleaved layout. */
int ser_gemm_bf16_nt(const uint16_t* a_hi, const uint16_t* a_lo, int lda, const uint16_t* w_hi,
                     const uint16_t* w_lo, int ldw, int M, int N, int K, const float* bias, int act,
                     const float* residual, int ldr, float* c_f32, uint16_t* c_hi, uint16_t* c_lo,
                     int ldc, void* stream);
/* ... with the K range cut into ksplit (<= 8, <= K / 32) slices, interleaved three-product operands only: slabs = [ksplit][M][N]
 * raw partial sums (no bias / activation / residual), added by the caller in slice order (ser_colsum over [ksplit, M * N]).  Used
 * for the fine-tuning encoders' weight gradients: long K (tokens, conv frames), small output. */
int ser_gemm_bf16_nt_splitk(const uint16_t* a_hi, const uint16_t* a_lo, int lda, const uint16_t* w_hi, const uint16_t* w_lo, int ldw,
                            int M, int N, int K, int ksplit, float* slabs, void* stream);
/* out[M*N] = the ks slabs added in slice order (+ bias[N] per row when given): the consumer of ser_gemm_bf16_nt_splitk for a product
 * that carries a bias (the forward of a Linear layer with a long K) */
int ser_sum_slabs_bias(const float* slabs, int ks, long long n, int N, const float* bias, float* out, void* stream);

/* y = LayerNorm(x (+ x2)) * gamma + beta over the last dim D; fp32 in; fp32 and/or split out.
 * Replaces nn.LayerNorm in hf wav2vec2 :429,:601,:606 and xlm_roberta :336-340,:394-398. */
int ser_layernorm(const float* x, const float* x2, const float* gamma, const float* beta, float eps,
                  int rows, int D, float* y_f32, uint16_t* y_hi, uint16_t* y_lo, void* stream);

/* softmax(Q K^T / sqrt(64) + key_bias) V for head_dim 64, Q/K/V read from the fused QKV
 * projection planes [B*S, 3*H] (q | k | v column blocks).  key_mask [B,S] 1/0 fp32 or NULL.
 * Replaces hf wav2vec2 :438-463 (eager/sdpa attention) and xlm_roberta :211-250.
 * ctx planes [B*S, H]. */
int ser_self_attention(const uint16_t* qkv_hi, const uint16_t* qkv_lo, const float* key_mask, int B,
                       int S, int heads, uint16_t* ctx_hi, uint16_t* ctx_lo, void* stream);

/* Measurement aid (bench.py roofline leg): HIP events around every ser_gemm_bf16 launch between
 * start and stop; stop returns summed kernel ms, algorithmic FLOPs (2*M*N*K) and the launch count. */
int ser_prof_gemm_start(void);
int ser_prof_gemm_stop(double* total_ms, double* total_flops, long long* launches);

/* ---------------------------------------------------------------------------------------------
 * frozen encoders, forward only
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const uint16_t* hi;
  const uint16_t* lo; /* NULL in SER_PREC_BF16 */
} SerSplitW;

typedef struct {
  SerSplitW qkv; /* [3H, H]  rows: q | k | v */
  const float* qkv_b;
  SerSplitW o; /* [H, H] */
  const float* o_b;
  const float *ln1_g, *ln1_b;
  SerSplitW f1; /* [F, H] */
  const float* f1_b;
  SerSplitW f2; /* [H, F] */
  const float* f2_b;
  const float *ln2_g, *ln2_b;
} SerLayerW;

typedef struct {
  int hidden, layers, heads, ffn;
  int n_conv;
  int conv_dim[SER_MAX_CONV], conv_kernel[SER_MAX_CONV], conv_stride[SER_MAX_CONV];
  int pos_kernel, pos_groups;
  float eps;
} SerW2vConfig;

typedef struct {
  const float* conv0_w;          /* [C0, k0] fp32 (in_channels = 1)                          */
  const float *gn_g, *gn_b;      /* GroupNorm(C0 groups) affine                              */
  SerSplitW conv_w[SER_MAX_CONV];/* i >= 1: [C_i, k_i, C_{i-1}] (channels-last taps)         */
  const float *fp_ln_g, *fp_ln_b;
  SerSplitW fp_w;                /* [H, C_last] */
  const float* fp_b;
  SerSplitW pos_w;               /* weight-norm folded, [G, H/G, k, H/G]                     */
  const float* pos_b;            /* [H] */
  const float *enc_ln_g, *enc_ln_b;
  const SerLayerW* layers;       /* host array of `layers` records                           */
} SerW2vWeights;

/* Wav2Vec2Model.forward (eval) on B equal-length raw clips, including the feature extractor's
 * zero-mean/unit-variance normalisation.  Replaces ref src/models/audio_encoder.py:91-110 =
 * hf feature_extraction_wav2vec2.py:78-96 + modeling_wav2vec2.py:1319 (409-419, 429-435, 689-727).
 * wave [B,T] fp32 -> out [B,S,H] fp32 (last_hidden_state). */
size_t ser_wav2vec2_workspace_bytes(const SerW2vConfig* cfg, int B, int T, int prec);
int ser_wav2vec2_out_len(const SerW2vConfig* cfg, int T);
int ser_wav2vec2_forward(const SerW2vConfig* cfg, const SerW2vWeights* w, const float* wave, int B,
                         int T, int prec, float* out, void* workspace, size_t workspace_bytes,
                         void* stream);

typedef struct {
  int hidden, layers, heads, ffn, vocab, max_pos, pad_id;
  float eps;
} SerXlmrConfig;

typedef struct {
  const float* word_emb; /* [vocab, H] fp32 */
  const float* pos_emb;  /* [max_pos, H]    */
  const float* type_emb; /* [1, H]          */
  const float *emb_ln_g, *emb_ln_b;
  const SerLayerW* layers;
} SerXlmrWeights;

/* XLMRobertaModel.forward (eval): embeddings (hf xlm_roberta :75-121, position ids :142-155),
 * post-LN layers (:421-463).  Replaces ref src/models/text_encoder.py:55.
 * ids [B,S] int64, attn_mask [B,S] fp32 1/0 -> out [B,S,H] fp32. */
size_t ser_xlmr_workspace_bytes(const SerXlmrConfig* cfg, int B, int S, int prec);
int ser_xlmr_forward(const SerXlmrConfig* cfg, const SerXlmrWeights* w, const int64_t* ids,
                     const float* attn_mask, int B, int S, int prec, float* out, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Front-end stages on their own (the whole-model entries above call the same launchers):
 * conv0 + GroupNorm + erf-GELU on raw clips, clip normalisation included (hf feature_extraction_wav2vec2.py:78-96 +
 * modeling_wav2vec2.py:302-323) -> channels-last bf16 planes [B, L0, C0], L0 = (T - KW) / ST + 1 (y_lo may be NULL);
 * the zero-padded (clip, group) slab the positional conv GEMM reads (modeling_wav2vec2.py:326-368): rows of H/G
 * channels, or - when slab_lo == slab_hi + 32 (interleaved planes) - rows padded with zeros to a multiple of 32 channels;
 * XLM-R embeddings: position ids + word/type/position gather + LayerNorm (modeling_xlm_roberta.py:75-121,142-155),
 * pos_scratch = B * S ints. */
size_t ser_conv0_workspace_bytes(int B, int L0, int C0);
int ser_conv0_gn_gelu(const float* wave, int B, int T, const float* w, const float* gn_g, const float* gn_b, int C0, int KW,
                      int ST, uint16_t* y_hi, uint16_t* y_lo, void* workspace, size_t workspace_bytes, void* stream);
int ser_posconv_slab(const float* z, int B, int S, int H, int G, int K, uint16_t* slab_hi, uint16_t* slab_lo, void* stream);
int ser_xlmr_embed(const int64_t* ids, int B, int S, const float* word_emb, const float* pos_emb, const float* type_emb,
                   const float* gamma, const float* beta, float eps, int D, int vocab, int max_pos, int pad_id,
                   int* pos_scratch, float* y, uint16_t* y_hi, uint16_t* y_lo, void* stream);

/* Both frozen encoders in one call on one stream (audio_encoder.py:110 + text_encoder.py:55 of the same batch).
 * With equal depth (Base 12 + 12) the transformer layers of the two models run in lock-step, ONE launch per step
 * for both (grouped GEMM / attention / LayerNorm), instead of ~90 small XLM-R launches queueing behind the
 * chip-filling Wav2Vec2 GEMMs on a second stream.  Same results and workspaces as the two separate calls. */
int ser_encoders_forward(const SerW2vConfig* wcfg, const SerW2vWeights* ww, const float* wave, int B, int T,
                         const SerXlmrConfig* xcfg, const SerXlmrWeights* xw, const int64_t* ids,
                         const float* attn_mask, int Bt, int St, int prec, float* out_audio, float* out_text,
                         void* ws_audio, size_t ws_audio_bytes, void* ws_text, size_t ws_text_bytes, void* stream);


/* ---------------------------------------------------------------------------------------------
 * trainable head, fp32 (forward and backward)
 * ------------------------------------------------------------------------------------------- */

/* C[M,N] (+)= act(A.B + bias) + residual with A(m,k) = a[m*sam + k*sak], B(k,n) = b[k*sbk + n*sbn],
 * exact fp32 on v_mfma_f32_16x16x4_f32.  One kernel serves y = x W^T, dx = dy W and dW += dy^T x of
 * every torch.nn.Linear in ref src/models/{audio_encoder.py:19-21, cross_attention.py:15-26,
 * pooling.py:9-13, fusion.py:8-16, classifier.py:77-129,190-197}. */
int ser_gemm_f32(const float* a, long long sam, long long sak, const float* b, long long sbk,
                 long long sbn, int M, int N, int K, const float* bias, int act, const float* residual,
                 int ldr, float* c, int ldc, int accumulate, void* stream);
/* ... with the number of bf16 MFMA products per multiply chosen by the caller: 3 (as above) or 1 (bf16 operands: what bf16
 * autocast computes, train.py --use_amp) */
int ser_gemm_f32_np(const float* a, long long sam, long long sak, const float* b, long long sbk, long long sbn, int M, int N, int K,
                    const float* bias, int act, const float* residual, int ldr, float* c, int ldc, int accumulate, int products,
                    void* stream);

/* The three products of a Linear layer with shape-specialised kernels behind them: M <= 16 rows (the
 * classifier / fusion run at M = batch) stream the weights straight into MFMA operands; wgrad over
 * many tokens splits the reduction over workgroups (workspace: ser_linear_wgrad_workspace_bytes) and
 * folds the bias gradient (column sums of dy) into the same pass. */
int ser_linear_fwd(const float* x, const float* W, const float* bias, int act, const float* residual, int ldr,
                   float* y, int M, int N, int K, void* stream);
/* The same for nprob independent problems of one dependency level in ONE launch (e.g. the six first-level projections of the
 * two directions of ref cross_attention.py:15-17,22-24; both adapters of ref audio_encoder.py:112 / text_encoder.py:57).
 * ptrs: 5 per problem {x, W, bias | NULL, residual | NULL, y}; dims: 5 per problem {M, N, K, act, ldr}.  Results are
 * bit-identical to nprob calls of ser_linear_fwd. */
int ser_linear_fwd_group(const void* const* ptrs, const int* dims, int nprob, void* stream);
/* dx_i[M,K] (+)= dy_i[M,N] W_i[N,K] for nprob problems.  ptrs: 3 per problem {dy, W, dx}; dims: 4 per problem
 * {M, N, K, accumulate}.  Bit-identical to nprob calls of ser_linear_dgrad (without a ReLU mask). */
int ser_linear_dgrad_group(const void* const* ptrs, const int* dims, int nprob, void* stream);
/* up to 32 token-level (M > 16) weight gradients, split over their token dimension, in one GEMM launch + one
 * reduce launch; ptrs = host array {dy, x, dW, db} per problem, dims = host array {M, N, K} per problem */
size_t ser_linear_wgrad_group_workspace_bytes(const int* dims, int nprob);
int ser_linear_wgrad_group(const void* const* ptrs, const int* dims, int nprob, int accumulate, void* workspace,
                           size_t workspace_bytes, void* stream);
/* up to 80 skinny (M <= 16) weight gradients in one launch; ptrs = host array {dy, x, dW, db} per problem,
 * dims = host array {N, K} per problem */
int ser_linear_wgrad_batch(const void* const* ptrs, const int* dims, int nprob, int M, int accumulate, void* stream);
/* first half of a classifier residual block in one launch (M <= 16, K <= 512):
 * x1 = LN(x; g1,b1), u = LN(x1; g2,b2), y = act(u W^T + bias); x1, u and stats[4][M] are written for backward. */
int ser_linear_fwd_ln2(const float* x, const float* W, const float* bias, int act, const float* g1, const float* b1,
                       const float* g2, const float* b2, float eps, float* y1, float* y2, float* stats, float* y,
                       int M, int N, int K, void* stream);
/* The residual stack of the deep classifier in ONE launch per direction (classifier.py:77-89 DeepResidualBlock,
 * :209-216 the 35-block loop), for M <= 16 rows and D <= 512 (a multiple of 16):
 *   x1 = LN(h; g1,b1)   u = LN(x1; g2,b2)   a = relu(u W1^T + c1)   h' = x1 + a W2^T + c2.
 * D/16 resident workgroups own 16 output columns each and hand the M x D activations to each other through
 * device memory as (value, tag) pairs: one 8-byte coherent store per element whose tag names the producing phase;
 * consumers re-read until the tags match (bounded), so a hand-off has no flag, fence or barrier.
 * ptr_table  : device array [L][8] of parameter pointers {g1,b1,g2,b2,W1,c1,W2,c2} per block
 * grad_table : device array [L][4] of gradient pointers {dg1,db1,dg2,db2} per block
 * scratch    : ser_stack_scratch_bytes(D) bytes per direction, 256-byte aligned, zeroed ONCE at allocation
 *              (word 0 counts launches, word 1 is a sticky "wait abandoned" flag, then the exchange ring)
 * Hs[L][M][D] block outputs, X1/U/A[L][M][D] and ST[L][4][M] saved for backward.
 * ser_stack_bwd: DH[L+1][M][D], DH[L] = gradient at the stack output on entry; on return DH[i] = gradient at the input
 * of block i; DA/DU/DX1[L][M][D] feed ser_linear_wgrad_batch and ser_stack_ln_param_bwd. */
int ser_stack_supported(int L, int M, int D);
size_t ser_stack_scratch_bytes(int D);
int ser_stack_fwd(const void* ptr_table, const float* x0, float* Hs, float* X1, float* U, float* A, float* ST, int L,
                  int M, int D, float eps, void* scratch, const void* drop_state, unsigned drop_site, float drop_p,
                  void* stream);
/* drop_state / drop_site / drop_p: the two nn.Dropout layers of every block in training mode (see ser_dropout; site ids
 * drop_site + 2 i and + 2 i + 1 for block i; NULL state or p = 0: identity).  With dropout, DT[L+1][M][D] receives the
 * dropped block-output gradients (operand of dW2 = DT^T a); otherwise it may be NULL and DH plays that role. */
int ser_stack_bwd(const void* ptr_table, const float* x0, const float* Hs, const float* X1, const float* A,
                  const float* ST, float* DH, float* DA, float* DU, float* DX1, int L, int M, int D, void* scratch,
                  const void* drop_state, unsigned drop_site, float drop_p, float* DT, void* stream);
int ser_stack_ln_param_bwd(const void* grad_table, const float* x0, const float* Hs, const float* X1, const float* ST,
                           const float* DU, const float* DX1, int L, int M, int D, int accumulate, void* stream);
/* Precision of the BACKWARD token-level (M > 16) head products (ser_linear_dgrad, ser_linear_wgrad*): 3 MFMA products
 * per multiply (hi*hi + lo*hi + hi*lo, fp32-equivalent; default) or 1 (operands rounded to bf16, fp32 accumulation -
 * what mixed-precision training does for every matmul).  Forward products always use 3.  Process-wide setting. */
int ser_set_head_backward_products(int n);
/* MFMA products per multiply of ser_linear_fwd / ser_linear_fwd_group: 3 (default, fp32-equivalent) or 1 (bf16 operands, fp32
 * accumulation: the arithmetic of the reference's --use_amp bf16 autocast, ref src/train.py:151). */
int ser_set_linear_forward_products(int n);
int ser_get_linear_forward_products(void);
int ser_get_head_backward_products(void);
/* relu_mask (may be NULL): the ReLU OUTPUT of the layer that produced x; when given, dx is multiplied by
 * relu'(mask), i.e. the activation backward is fused into the dgrad epilogue. */
int ser_linear_dgrad(const float* dy, const float* W, const float* relu_mask, float* dx, int M, int N, int K,
                     int accumulate, void* stream);
/* two weight gradients of a classifier block (M <= 16) in one launch */
int ser_linear_wgrad_pair(const float* dya, const float* xa, float* dWa, float* dba, int Na, int Ka,
                          const float* dyb, const float* xb, float* dWb, float* dbb, int Nb, int Kb, int M,
                          int accumulate, void* stream);
size_t ser_linear_wgrad_workspace_bytes(int M, int N, int K);
int ser_linear_wgrad(const float* dy, const float* x, float* dW, float* db, int M, int N, int K, int accumulate,
                     void* workspace, size_t workspace_bytes, void* stream);

/* nn.LayerNorm forward keeping z = x (+ x2), mean, rstd for backward (cross_attention.py:28-29,
 * classifier.py:79,107,118,125); backward gives dx (+ dx_add) and dgamma/dbeta. */
int ser_layernorm_fwd(const float* x, const float* x2, const float* gamma, const float* beta, float eps,
                      int rows, int D, float* y, float* z, float* mean, float* rstd, void* stream);
size_t ser_layernorm_bwd_workspace_bytes(int rows, int D); /* 0 for few rows; else partial dgamma/dbeta slices */
int ser_layernorm_bwd(const float* dy, const float* z, const float* mean, const float* rstd,
                      const float* gamma, const float* dx_add, int rows, int D, float* dx, float* dgamma,
                      float* dbeta, int accumulate_params, void* workspace, void* stream);
/* The hidden dropout, the residual add and the LayerNorm of a post-LN transformer block (hf modeling_wav2vec2.py:591-608,
 * modeling_xlm_roberta.py:300-311) in one pass each way: y = LN(dropout(x) + x2), z = dropout(x) + x2 kept for backward; the mask
 * of element (row, col) is the one ser_dropout gives element row * D + col with the same (state, site, p).  Backward writes
 * dx2 = the LayerNorm input gradient and dx = dx2 times the mask. */
int ser_layernorm_drop_fwd(const float* x, const float* x2, const float* gamma, const float* beta, float eps, int rows, int D, float* y,
                           float* z, float* mean, float* rstd, const void* drop_state, unsigned drop_site, float drop_p, void* stream);
int ser_layernorm_drop_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma, int rows, int D,
                           float* dx, float* dx2, float* dgamma, float* dbeta, int accumulate_params, void* workspace,
                           const void* drop_state, unsigned drop_site, float drop_p, void* stream);

/* the two chained LayerNorms of a classifier block (classifier.py:209-210) fused: forward keeps
 * stats[4][rows] = mean1, rstd1, mean2, rstd2; backward returns dx = LN1'(LN2'(du) + dres) and all four
 * parameter gradients from one single-workgroup launch (rows = batch). */
int ser_layernorm2_fwd(const float* x, const float* g1, const float* b1, const float* g2, const float* b2, float eps,
                       int rows, int D, float* y1, float* y2, float* stats, void* stream);
int ser_layernorm2_bwd(const float* du, const float* dres, const float* x, const float* y1, const float* stats,
                       const float* g1, const float* g2, int rows, int D, float* dx, float* dg1, float* db1,
                       float* dg2, float* db2, int accumulate, void* stream);
int ser_colsum(const float* x, int M, int N, int ld, float* out, int accumulate, void* stream);
/* the same for tall x (thousands of rows): 32 row chunks in parallel, then their sum, in a fixed order; ws from
 * ser_colsum_tall_workspace_bytes(N).  The bias gradient of the fine-tuning encoders' Linear layers. */
size_t ser_colsum_tall_workspace_bytes(int N);
int ser_colsum_tall(const float* x, int M, int N, int ld, float* out, void* ws, void* stream);
int ser_act_fwd(const float* x, int act, long long n, float* y, void* stream);
int ser_act_bwd(const float* dy, const float* y, int act, long long n, float* dx, void* stream);
/* activation followed by dropout in one pass, and its backward (see ser_dropout for the generator arguments) */
int ser_act_drop_fwd(const float* x, int act, long long n, float* y, const void* drop_state, unsigned drop_site, float drop_p,
                     void* stream);
int ser_act_drop_bwd(const float* dy, const float* y, int act, long long n, float* dx, const void* drop_state,
                     unsigned drop_site, float drop_p, void* stream);
int ser_axpby(const float* x, float a, float b, long long n, float* y, void* stream); /* y = a x + b y */
int ser_scale_dev(float* x, const float* s, long long n, void* stream);            /* x *= s[0], s on device */

/* softmax(q k^T / sqrt(hd) + key mask) v of nn.MultiheadAttention (cross_attention.py:41,49;
 * torch nn/functional.py multi_head_attention_forward).  P [B,heads,Sq,Sk] is kept for backward. */
int ser_xattn_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* key_mask, int B,
                  int Sq, int Sk, int heads, int head_dim, float* P, float* ctx, int ldc, const void* drop_state,
                  unsigned drop_site, float drop_p, float* P_dropped, void* stream);
/* drop_*: nn.MultiheadAttention's attention dropout in training mode (ctx = (P * m / (1 - p)) v; P as stored is the
 * softmax output and P_dropped [B, heads, Sq, Sk] receives P * m / (1 - p) for dv in backward; see ser_dropout for the
 * generator; NULL state or p = 0: identity, P_dropped may be NULL). */
int ser_xattn_bwd(const float* dctx, int ldc, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                  const float* P, int B, int Sq, int Sk, int heads, int head_dim, float* dS, float* dq, int lddq,
                  float* dk, int lddk, float* dv, int lddv, const void* drop_state, unsigned drop_site, float drop_p,
                  const float* P_dropped, void* stream);

/* AttentiveStatsPooling core (pooling.py:21-28): masked softmax over time of `logits`, weighted
 * mean and std -> out [B,2D]; alpha [B,S] kept for backward. */
int ser_pool_fwd(const float* x, const float* logits, const float* mask, int B, int S, int D,
                 float* alpha, float* out, void* stream);
int ser_pool_bwd(const float* dout, const float* x, const float* alpha, const float* out, int B, int S,
                 int D, float* dx, float* dalpha_scratch, float* dlogits, void* stream);

/* FusionLayer mix (fusion.py:21-25) from the two gate logits ga, gt [B]. */
int ser_fusion_mix_fwd(const float* a, const float* t, const float* ga, const float* gt, int B, int P,
                       float* out, void* stream);
int ser_fusion_mix_bwd(const float* dout, const float* a, const float* t, const float* ga, const float* gt,
                       int B, int P, float* da, float* dt, float* dga, float* dgt, void* stream);

/* The training loss of train.py:154-168 (label-smoothed CE + w_focal * class-balanced focal +
 * w_unc * uncertainty term + w_proto * prototype loss) and its gradients in one launch.
 * losses[5] = {total, ce, focal, unc, proto}.  grad_scale: device scalar (NULL = 1). */
int ser_train_loss(const float* logits, const float* unc, const float* fused, const float* protos,
                   const int64_t* labels, int B, int C, int D, float smoothing, float cb_beta, float gamma,
                   float w_focal, float w_unc, float w_proto, float margin, int use_proto,
                   const float* grad_scale, float* losses, float* dlogits, float* dunc, float* dfused,
                   float* dprotos, void* stream);

/* OpenMax rescale of eval logits (classifier.py:240-275), in place. */
int ser_openmax(const float* feats, const float* act_vec, const float* walpha, const float* wbeta,
                const float* wtau, int B, int C, int F, float thresh, float reduce, float* logits,
                void* stream);

/* nn.Dropout(p) in training mode (cross_attention.py:28,43,51; fusion.py:9,12; classifier.py:83,85,109,127,195):
 * y = x * m / (1 - p), m ~ Bernoulli(1 - p) from a counter-based generator keyed by (*state, site, element index).
 * `state` is a device word advanced by the host once per training step, `site` names the layer; the backward pass
 * calls the same function on the gradient (the mask is regenerated, never stored).  state == NULL or p == 0: copy. */
int ser_dropout(const float* x, long long n, const void* state, unsigned site, float p, float* y, void* stream);

/* torch.optim.AdamW update of a flat fp32 segment (train.py:72-83,169-177).
 * hyper (device) = {lr, 1 - beta1^t, sqrt(1 - beta2^t)}; effective lr = hyper[0] * lr_mult. */
int ser_adamw(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float lr_mult,
              float weight_decay, float beta1, float beta2, float eps, void* stream);
/* ---------------------------------------------------------------------------------------------
 * fine-tuning path of the encoders (BASELINE config 3: reference freeze_base=False,
 * src/models/audio_encoder.py:15-17, text_encoder.py:13-15).  Products, LayerNorm and attention
 * reuse ser_gemm_f32 / ser_linear_* / ser_layernorm_fwd,bwd / ser_xattn_fwd,bwd; these are the rest.
 * ------------------------------------------------------------------------------------------- */
/* dx = dy * d/dx[x Phi(x)] from the pre-activation x (hf activations.py GELUActivation). */
int ser_gelu_bwd(const float* dy, const float* x, long long n, float* dx, void* stream);
/* ... of y = dropout(gelu(x)) (the activation dropout of hf Wav2Vec2FeedForward :560-574): the mask of the forward pass
 * (state, site, element index) multiplies dy in the same pass */
int ser_gelu_drop_bwd(const float* dy, const float* x, long long n, float* dx, const void* drop_state, unsigned drop_site, float drop_p,
                      void* stream);
/* GroupNorm with one channel per group = normalisation over time per (clip, channel); x [B][Ls][C] channels-last, the
 * first L of a clip's Ls rows are its frames, the rest padding (written as zeros) (hf modeling_wav2vec2.py:302-323).
 * workspace: ser_colnorm_workspace_bytes(B, C). */
size_t ser_colnorm_workspace_bytes(int B, int C);
int ser_colnorm_fwd(const float* x, int B, int L, int Ls, int C, const float* gamma, const float* beta, float eps, float* y,
                    float* mean, float* rstd, void* workspace, void* stream);
int ser_colnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, int B,
                    int L, int Ls, int C, float* dx, float* dgamma, float* dbeta, int accumulate, void* workspace, void* stream);
/* Adjoint of the positional conv's window view: dslab[r][c] = sum_j dwin[r - j][j * Cg + c] (hf :326-368). */
int ser_toeplitz_add(const float* dwin, int rows_win, int K, int Cg, int rows_slab, float* dslab, void* stream);
/* The positional conv of the fine-tuning path on the frozen encoders' resident-slab kernel (csrc/posconv.hip; K = 128 taps, 48 or 64
 * channels per group, S <= 352 / 224 frames: ser_posconv_direct_supported).  ser_posconv_pack: Wp [H][Cg][K] fp32 (hf conv weight
 * after weight-norm) -> the kernel's operand (H * K * 128 bf16: per row and tap the interleaved hi / lo planes of 64 zero-padded
 * channels); flip = 1 packs the operand of the input-gradient correlation (rows = in channels, channels = out channels, taps
 * reversed).  ser_posconv_fwd: out = GELU(conv(z) + bias) + z, raw (optional) = conv(z) + bias.  ser_posconv_dgrad:
 * dz = conv^T(dpre) + add with the flip = 1 operand.  Three bf16 products per multiply (hf modeling_wav2vec2.py:326-368). */
int ser_posconv_direct_supported(int S, int H, int G, int K);
int ser_posconv_pack(const float* wp, int H, int G, int K, int flip, uint16_t* packed, void* stream);
int ser_posconv_fwd(const float* z, const uint16_t* packed, const float* bias, int B, int S, int H, int G, int K, float* out, float* raw,
                    void* stream);
int ser_posconv_dgrad(const float* dpre, const uint16_t* packed_flip, const float* add, int B, int S, int H, int G, int K, float* dz,
                      void* stream);
/* Adjoint of a strided Conv1d's window view (the feature extractor's layers 1-6, hf :266-300, channels-last rows):
 * dx[r][c] = sum over taps j with r - j = s m, 0 <= m < M, of dwin[m][j * Cin + c]; every one of the rows_in rows is written. */
int ser_conv_col2im(const float* dwin, int M, int k, int s, int Cin, long long rows_in, float* dx, void* stream);
/* XLM-R embeddings word[id] + type[0] + pos[pid] and the scatter-add backward; rows whose index equals pad_id leave
 * that table's gradient untouched (nn.Embedding(padding_idx), hf modeling_xlm_roberta.py:75-121). */
int ser_embed_fwd(const int64_t* ids, const int64_t* pos, const float* wemb, const float* pemb, const float* temb,
                  int rows, int D, int vocab, int max_pos, float* e, void* stream);
int ser_embed_bwd(const float* de, const int64_t* ids, const int64_t* pos, int rows, int D, int vocab, int max_pos,
                  int pad_id, float* dw, float* dp, float* dt, void* stream);
/* (x - mean) / sqrt(var + 1e-7) per clip (hf feature_extraction_wav2vec2.py:78-96); stats: B float2 of scratch. */
int ser_wave_normalize(const float* wave, int B, int T, float* out, void* stats, void* stream);

/* Eval-side consumers of the logits (ref src/eval.py:192-206, src/utils.py:11-14): z = logits / temperature,
 * probs = softmax(z), pred = first arg-max, energy = -logsumexp(z); any output may be NULL.  C <= 64. */
int ser_eval_consumers(const float* logits, int B, int C, float temperature, float* probs, int64_t* pred, float* energy,
                       void* stream);
/* --calibrate grid search (ref src/eval.py:49-67): ece[g] = mean_n |max softmax(logits_n / temps[g]) - [argmax_n == label_n]|. */
int ser_temperature_grid(const float* logits, const int64_t* labels, int N, int C, const float* temps, int G, float* ece,
                         void* stream);

/* Quality-gate / audio-conditioning front end of the reference's default AudioEncoder(), batched
 * over B equal-length clips of T >= 2048 samples at 16 kHz, no host synchronisation; one workspace of
 * ser_frontend_workspace_bytes(B, T) serves both calls.  ser_frontend_init uploads the FFT tables (call once, outside any
 * stream capture; the two calls below do it on first use otherwise).
 *
 * ser_quality_gates (ref src/models/quality_gates.py:497-560, vad_method = "librosa"): energy VAD (:111-137), STFT SNR
 * (:189-214), clipping (:216-226), spectral naturalness (:228-247), music / laughter scores (:320-345), abstain policy and
 * quality score (:347-403).  lid[b] = {language entropy, dominant-language confidence} of clip b's transcript as
 * LanguageIdentifier.identify_language returns them (:252-301; {1.0, 0.0} without text, :514-517).  pad_reflect: 0 = the
 * zero centre padding of librosa >= 0.10, 1 = the reflect padding of 0.9.x.
 *   q_raw     [B, 8]  inputs of quality_projection in the reference's order (:544-553)
 *   q_metrics [B, 8]  speech_prob, snr_db, clipping_percent, spectral_naturalness, music_prob, laughter_prob,
 *                     quality_score, decision (may be NULL)
 *   decision  [B]     0 reject, 1 uncertain, 2 accept
 *
 * ser_audio_conditioning (ref src/models/audio_conditioning.py:503-584 with noisereduce / pyloudnorm absent): hum notch
 * (:66-95), high-pass (:107-150), energy SNR + Wiener denoise (:161-215, :244-256), T60 estimate (:274-301), loudness
 * normalisation with compression (:364-440).  decision (may be NULL): clips whose entry is not 2 are processed as all-zero
 * audio, as AudioEncoder.forward substitutes them (src/models/audio_encoder.py:74-77).
 *   out    [B, T]   conditioned clips
 *   c_raw  [B, 12]  inputs of conditioning_projection in the reference's order (:562-575)
 *   c_meta [B, 12]  hpf_cutoff, hum50, hum60, snr_before, snr_after, denoise_gain_db, estimated_t60, lufs_original,
 *                   lufs_adjustment, peak_reduction_db, compression_ratio, noise type (0 unknown, 1 low_frequency,
 *                   2 high_frequency, 3 mid_frequency, 4 white_noise) (may be NULL) */
size_t ser_frontend_workspace_bytes(int B, int T);
int ser_frontend_init(void);
int ser_quality_gates(const float* wave, int B, int T, int sample_rate, const float* lid, int pad_reflect, float* q_raw,
                      float* q_metrics, int* decision, void* workspace, size_t workspace_bytes, void* stream);
int ser_audio_conditioning(const float* wave, const int* decision, int B, int T, int sample_rate, float* out, float* c_raw,
                           float* c_meta, void* workspace, size_t workspace_bytes, void* stream);

/* HIP-event timing of every ser_gemm_f32-family launch between start and stop (measurement aid). */
int ser_prof_gemm_f32_start(void);
int ser_prof_gemm_f32_stop(double* total_ms, double* total_flops, long long* launches);

/* Measured tile height (64..192 rows) for the BN = 128 encoder GEMMs of shape (rows_total, N, K), recorded by the
 * engines' one-time timing pass; `three_products` selects the table of the interleaved three-product mode.  Speed only. */
int ser_gemm_tile_hint(long long rows_total, int N, int K, int bm);
int ser_gemm_tile_hint_mode(long long rows_total, int N, int K, int three_products, int bm);

/* the same update for many flat segments (the ten optimizer groups of train.py:72-83 x their buckets) in one launch per
 * 16 segments; ptrs = host array {p, g, m, v} per segment, n / lr_mult / weight_decay = host arrays per segment */
int ser_adamw_multi(const void* const* ptrs, const long long* n, const float* lr_mult, const float* weight_decay, int nseg,
                    const float* hyper, float beta1, float beta2, float eps, void* stream);
/* ... with an optional device word per segment (gates = host array of device `const int*`, null entries = always update):
 * a segment whose word is non-zero when the kernel runs takes no update, moments included - what torch.optim.AdamW does for a
 * parameter without a gradient.  The graph-captured fine-tuning step computes LayerDrop-skipped layers (hf
 * modeling_wav2vec2.py:700-703) and discards them by a select, so their parameters need this to stay untouched. */
int ser_adamw_multi_gated(const void* const* ptrs, const long long* n, const float* lr_mult, const float* weight_decay,
                          const void* const* gates, int nseg, const float* hyper, float beta1, float beta2, float eps, void* stream);

/* ---- data feed on the device (src/data/preprocess.py:50-73) ---------------------------
 * ser_resample: torchaudio.functional.resample (windowed sinc, Hann window, lowpass_filter_width 6, rolloff 0.99 by
 * default) of x[B, T] into y[B, ser_resample_out_len(T, orig, new)]; speed_perturb is the 16000 -> int(16000 f) ->
 * 16000 round trip of it.  ser_add_noise_snr: y = clamp(x + N(0, mean(x^2) / 10^(snr/10)), -1, 1) per clip, noise from
 * a counter-based generator keyed by (seed, clip, sample); sigma = device scratch [B]. */
int ser_resample_out_len(int T, int orig_freq, int new_freq);
int ser_resample(const float* x, int B, int T, int orig_freq, int new_freq, int lowpass_filter_width, float rolloff, float* y,
                 void* stream);
int ser_add_noise_snr(const float* x, int B, int T, const float* snr_db, unsigned long long seed, float* sigma, float* y,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * tuning and diagnostics (process-wide settings; none changes a result: every tile configuration
 * sums k in the same order per accumulator, tests/test_gpu_interleaved.py).  No reference
 * counterpart: the reference leaves kernel selection to torch / hipBLASLt.
 * ------------------------------------------------------------------------------------------- */

/* Tile plan of the encoder GEMM for one shape (rows over all batch entries, N, K; three_products = interleaved
 * split-bf16 operands): cfg = SER_GEMM_CFG_* (64..192 = rows of a BM x 128 tile, 3xxx single LDS buffer, 7xxx BM x 64,
 * 1xxx/2xxx/5xxx/6xxx 512-thread tiles), ksplit = split-K factor.  Filled by the engines' timing pass
 * (_engines.tune_gemm_shapes); shapes without a plan use the library's cost model. */
int ser_gemm_plan_set(long long rows_total, int N, int K, int three_products, int cfg, int ksplit);
int ser_gemm_plan_get(long long rows_total, int N, int K, int three_products, int* cfg, int* ksplit);
/* Occupancy headroom: an encoder GEMM launches at most pct % of the workgroup slots its tile configuration could hold
 * resident and walks its tiles grid-stride, so the head kernels of the other queues find free slots. */
int ser_set_gemm_occupancy_pct(int pct);
int ser_get_gemm_occupancy_pct(void);

/* experiment / test knobs (scripts/, tests/): force a tile configuration, LDS pipeline depths, extra dynamic LDS, a
 * persistent grid cap, kernel variants; stand-alone probes of single kernels */
int ser_debug_set_gemm_bm(int cfg);
int ser_debug_set_gemm_group_m(int m_tiles_per_super_tile);
int ser_debug_set_gemm_stages(int s128, int s64x128, int s64, int s128x64);
int ser_debug_set_gemm_stages_tall(int s96, int s160, int s192);
int ser_debug_set_gemm_lds_pad(int bytes);
int ser_debug_set_gemm_persist(int cap);
int ser_debug_set_attention_small_variant(int v);
int ser_debug_set_attention_generic(int on);
int ser_debug_set_posconv_gemm(int on);
int ser_debug_set_pair_mask(int m);
int ser_debug_set_f32_bk(int bk);
int ser_debug_set_head_x3(int v);
int ser_debug_set_dgrad16(int v);
int ser_debug_gemm_pair(const uint16_t* a0, const uint16_t* w0, int M0, int N0, int K0, float* c0, const uint16_t* a1,
                        const uint16_t* w1, int M1, int N1, int K1, float* c1, void* stream);
int ser_debug_gemm_batched(const uint16_t* a, const uint16_t* w, int M, int N, int K, int nb, long long sa, long long sw,
                           uint16_t* c, long long sc, int il, void* stream);
int ser_debug_gemm_il_cfg(const uint16_t* a, const uint16_t* w, int M, int N, int K, int cfg, int ksplit, float* c_f32,
                          void* stream);
int ser_debug_posconv(const float* z, const uint16_t* w_il, const float* bias, float* out, int B, int S, int H, int G, int K,
                      int direct, uint16_t* slab_il, void* stream);
int ser_debug_barrier_probe(int G, int rounds, int mode, void* flags, void* data, void* errors, void* stream);
int ser_debug_stack_timeline(void* buf);
/* out[2 b] = shader cycles, out[2 b + 1] = 100 MHz ticks that block b's dependent-FMA loop of `iters` steps took */
int ser_debug_clock_probe(void* out, int blocks, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SER_HIP_H */
