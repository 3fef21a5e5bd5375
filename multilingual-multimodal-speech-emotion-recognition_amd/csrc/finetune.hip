// Kernels that only the fine-tuning path of the encoders needs (BASELINE config 3: reference freeze_base=False,
// src/models/audio_encoder.py:15-17, text_encoder.py:13-15).  Its matrix products, LayerNorms and attention reuse the
// fp32 operators of the trainable head (gemm_f32.hip, head.hip); what is left are the pieces the head does not have:
//   * erf-GELU backward from the pre-activation                       (hf activations.py GELUActivation)
//   * GroupNorm(C groups of one channel) over time, channels-last      (hf modeling_wav2vec2.py:302-323)
//   * overlap-add of the positional conv's window gradients            (hf modeling_wav2vec2.py:326-368)
//   * XLM-R embedding gather-sum and its scatter-add backward          (hf modeling_xlm_roberta.py:75-121, padding_idx rows get no gradient)
//   * clip normalisation as a stand-alone op                           (hf feature_extraction_wav2vec2.py:78-96)
// All HBM-bound, coalesced along the channel / feature axis.
#include "ser_common.h"

namespace {

static inline unsigned ew_grid(long long n, int per_block = 256) {
  long long b = (n + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

// d/dx [x Phi(x)] = Phi(x) + x phi(x)
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, long long n, float* __restrict__ dx,
                                SerDropout drop) {
  // y = dropout(gelu(x)) when `drop` is active: the mask of element i (the forward pass's triple) multiplies dy first
  const bool dropping = drop.state != nullptr && drop.p > 0.f;
  const unsigned long long dst = dropping ? *drop.state : 0ull;
  const unsigned dth = ser_drop_thresh(drop.p);
  const float dsc = 1.0f / (1.0f - drop.p);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * v * v);
    float g = dy[i];
    if (dropping) g *= ser_drop_mult(dst, drop.site, (unsigned)i, dth, dsc);
    dx[i] = g * (cdf + v * pdf);
  }
}

// ---- GroupNorm with one channel per group == per-(clip, channel) normalisation over time; x [B][L][C] channels-last.
// One workgroup = 64 channels of one clip x a slice of the frames; partial sums are combined with atomics into [B][C]
// double accumulators (zeroed by the caller), so the reduction order varies in the last bits only at double precision.
// Ls = rows per clip in memory (>= L, the valid frames): the fine-tuning path keeps every clip on a padded row stride
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ x, int L, int Ls, int C, int rows_per_block,
                                                       double* __restrict__ sum, double* __restrict__ sq) {
  __shared__ double sh[2][4][64];
  const int b = blockIdx.z, c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int t0 = blockIdx.y * rows_per_block, t1 = min(L, t0 + rows_per_block);
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int t = t0 + rg; t < t1; t += 4) {
      const double v = x[((long long)b * Ls + t) * C + c];
      s += v;
      q += v * v;
    }
  sh[0][rg][threadIdx.x & 63] = s;
  sh[1][rg][threadIdx.x & 63] = q;
  __syncthreads();
  if (rg == 0 && c < C) {
    const int l = threadIdx.x & 63;
    atomicAdd(sum + (long long)b * C + c, (sh[0][0][l] + sh[0][1][l]) + (sh[0][2][l] + sh[0][3][l]));
    atomicAdd(sq + (long long)b * C + c, (sh[1][0][l] + sh[1][1][l]) + (sh[1][2][l] + sh[1][3][l]));
  }
}
__global__ void colstats_finish_kernel(const double* __restrict__ sum, const double* __restrict__ sq, int n, int L, float eps,
                                       float* __restrict__ mean, float* __restrict__ rstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double m = sum[i] / L;
  double var = sq[i] / L - m * m;
  if (var < 0.0) var = 0.0;
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}
__global__ void colnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, int L, int Ls, int C, long long n,
                                     float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long row = i / C;
    const long long bc = (row / Ls) * C + c;
    y[i] = (row % Ls) < L ? (x[i] - mean[bc]) * rstd[bc] * gamma[c] + beta[c] : 0.f;      // padding rows of a clip: zero
  }
}
// backward, pass 1: per (clip, channel) s1 = sum_t dy, s2 = sum_t dy * xhat  (double accumulators, zeroed by the caller)
__global__ __launch_bounds__(256) void colnorm_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                int L, int Ls, int C, int rows_per_block, double* __restrict__ s1,
                                                                double* __restrict__ s2) {
  __shared__ double sh[2][4][64];
  const int b = blockIdx.z, c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int t0 = blockIdx.y * rows_per_block, t1 = min(L, t0 + rows_per_block);
  double a = 0.0, q = 0.0;
  if (c < C) {
    const float m = mean[(long long)b * C + c], r = rstd[(long long)b * C + c];
    for (int t = t0 + rg; t < t1; t += 4) {
      const long long o = ((long long)b * Ls + t) * C + c;
      const double g = dy[o];
      a += g;
      q += g * (double)((x[o] - m) * r);
    }
  }
  sh[0][rg][threadIdx.x & 63] = a;
  sh[1][rg][threadIdx.x & 63] = q;
  __syncthreads();
  if (rg == 0 && c < C) {
    const int l = threadIdx.x & 63;
    atomicAdd(s1 + (long long)b * C + c, (sh[0][0][l] + sh[0][1][l]) + (sh[0][2][l] + sh[0][3][l]));
    atomicAdd(s2 + (long long)b * C + c, (sh[1][0][l] + sh[1][1][l]) + (sh[1][2][l] + sh[1][3][l]));
  }
}
// pass 2: dx = gamma * rstd * (dy - s1 / L - xhat * s2 / L)
__global__ void colnorm_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                         const double* __restrict__ s1, const double* __restrict__ s2, int L, int Ls, int C, long long n,
                                         float* __restrict__ dx) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long row = i / C;
    const long long bc = (row / Ls) * C + c;
    const float r = rstd[bc], xh = (x[i] - mean[bc]) * r;
    dx[i] = (row % Ls) < L ? gamma[c] * r * (dy[i] - (float)(s1[bc] / L) - xh * (float)(s2[bc] / L)) : 0.f;
  }
}
// dgamma[c] (+)= sum_b s2[b][c], dbeta[c] (+)= sum_b s1[b][c]
__global__ void colnorm_bwd_param_kernel(const double* __restrict__ s1, const double* __restrict__ s2, int B, int C, int accumulate,
                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, q = 0.0;
  for (int b = 0; b < B; ++b) {
    a += s1[(long long)b * C + c];
    q += s2[(long long)b * C + c];
  }
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)q;
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)a;
}

// ---- overlap-add: dslab[r][c] = sum_j dwin[r - j][j * Cg + c] over the taps j with 0 <= r - j < rows_win
// (the adjoint of viewing slab rows t .. t + K - 1 as one window row of K * Cg values)
__global__ void toeplitz_add_kernel(const float* __restrict__ dwin, int rows_win, int K, int Cg, int rows_slab,
                                    float* __restrict__ dslab) {
  const long long n = (long long)rows_slab * Cg;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cg), r = (int)(i / Cg);
    const int j0 = max(0, r - (rows_win - 1)), j1 = min(K - 1, r);
    float a = 0.f;
    for (int j = j0; j <= j1; ++j) a += dwin[(long long)(r - j) * K * Cg + (long long)j * Cg + c];
    dslab[i] = a;
  }
}

// ---- XLM-R embeddings: e[row] = word[id] + type[0] + pos[pid]
__global__ void embed_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos, const float* __restrict__ wemb,
                                 const float* __restrict__ pemb, const float* __restrict__ temb, int rows, int D, int vocab,
                                 int max_pos, float* __restrict__ e) {
  const int row = blockIdx.x;
  if (row >= rows) return;
  long long id = ids[row], p = pos[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
  for (int d = threadIdx.x; d < D; d += blockDim.x)
    e[(long long)row * D + d] = (wemb[id * D + d] + temb[d]) + pemb[p * D + d];
}
// Scatter-add without atomics (results do not depend on the order workgroups run in: reruns, replicas and a captured step
// agree bit for bit): the first token row that uses an index owns it and adds up the gradients of all rows with that index
// in increasing row order.  Rows whose index is the padding index leave that table untouched (nn.Embedding(padding_idx)); the
// token-type table has one row (type 0), owned by row 0.
__global__ void embed_bwd_kernel(const float* __restrict__ de, const int64_t* __restrict__ ids, const int64_t* __restrict__ pos,
                                 int rows, int D, int vocab, int max_pos, int pad_id, float* __restrict__ dw, float* __restrict__ dp,
                                 float* __restrict__ dt) {
  const int row = blockIdx.x;
  if (row >= rows) return;
  auto wid = [&](int r) { const long long v = ids[r]; return v < 0 ? 0ll : (v >= vocab ? (long long)vocab - 1 : v); };
  auto pid = [&](int r) { const long long v = pos[r]; return v < 0 ? 0ll : (v >= max_pos ? (long long)max_pos - 1 : v); };
  const long long id = wid(row), p = pid(row);
  bool own_w = id != pad_id, own_p = p != pad_id;
  const bool own_t = row == 0;
  for (int r = 0; r < row && (own_w || own_p); ++r) {
    if (wid(r) == id) own_w = false;
    if (pid(r) == p) own_p = false;
  }
  if (!(own_w || own_p || own_t)) return;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float sw = 0.f, sp = 0.f, st = 0.f;
    for (int r = row; r < rows; ++r) {
      const bool mw = own_w && wid(r) == id, mp = own_p && pid(r) == p;
      if (!(mw || mp || own_t)) continue;
      const float g = de[(long long)r * D + d];
      if (mw) sw += g;
      if (mp) sp += g;
      if (own_t) st += g;
    }
    if (own_w) dw[id * D + d] += sw;
    if (own_p) dp[p * D + d] += sp;
    if (own_t) dt[d] += st;
  }
}

// Input gradient of a strided Conv1d from the per-window gradients: dx[r][c] = sum over taps j with (r - j) = s m, 0 <= m < M of
// dwin[m][j Cin + c] (channels-last rows; taps added in increasing j: a fixed order, every dx row written, no atomics).
__global__ void conv_col2im_kernel(const float* __restrict__ dwin, int M, int k, int s, int Cin, long long rows_in,
                                   float* __restrict__ dx) {
  const int c4n = Cin >> 2;
  const long long n = rows_in * c4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c4n;
    const int c = (int)(i % c4n) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < k; ++j) {
      const long long t = r - j;
      if (t < 0 || t % s != 0 || t / s >= M) continue;
      const float4 v = *(const float4*)(dwin + (t / s) * (long long)k * Cin + (long long)j * Cin + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *(float4*)(dx + r * Cin + c) = acc;
  }
}

__global__ void wave_normalize_kernel(const float* __restrict__ wave, const float2* __restrict__ stats, int T, long long n,
                                      float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float2 s = stats[i / T];
    out[i] = (wave[i] - s.x) * s.y;
  }
}

}  // namespace

int ser_launch_wave_stats(const float* wave, int B, int T, void* stats, hipStream_t st);   // elementwise.hip

extern "C" int ser_gelu_bwd(const float* dy, const float* x, long long n, float* dx, void* stream) {
  if (n <= 0) return SER_OK;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, x, n, dx, SerDropout{nullptr, 0u, 0.f});
  SER_LAUNCH_CHECK();
  return SER_OK;
}

/* backward of y = dropout(gelu(x)) (ser_act_drop_fwd with SER_ACT_GELU): dx = dy . mask / (1 - p) . gelu'(x) in one pass */
extern "C" int ser_gelu_drop_bwd(const float* dy, const float* x, long long n, float* dx, const void* drop_state, unsigned drop_site,
                                 float drop_p, void* stream) {
  if (n <= 0) return SER_OK;
  SER_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gelu_drop_bwd: p=%f out of range", drop_p);
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, x, n, dx,
                     SerDropout{(const unsigned long long*)drop_state, drop_site, drop_p});
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" size_t ser_colnorm_workspace_bytes(int B, int C) { return (size_t)2 * B * C * sizeof(double) + 256; }

extern "C" int ser_colnorm_fwd(const float* x, int B, int L, int Ls, int C, const float* gamma, const float* beta, float eps, float* y,
                               float* mean, float* rstd, void* workspace, void* stream) {
  SER_REQUIRE(x && y && mean && rstd && workspace && B > 0 && L > 0 && Ls >= L && C > 0, "colnorm_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  double* sum = (double*)workspace;
  double* sq = sum + (size_t)B * C;
  SER_CHECK_HIP(hipMemsetAsync(workspace, 0, (size_t)2 * B * C * sizeof(double), st));
  const int rpb = 512;
  hipLaunchKernelGGL(colstats_kernel, dim3(ceil_div(C, 64), ceil_div(L, rpb), B), dim3(256), 0, st, x, L, Ls, C, rpb, sum, sq);
  hipLaunchKernelGGL(colstats_finish_kernel, dim3(ceil_div(B * C, 256)), dim3(256), 0, st, sum, sq, B * C, L, eps, mean, rstd);
  const long long n = (long long)B * Ls * C;
  hipLaunchKernelGGL(colnorm_apply_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, mean, rstd, gamma, beta, L, Ls, C, n, y);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_colnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, int B,
                               int L, int Ls, int C, float* dx, float* dgamma, float* dbeta, int accumulate, void* workspace, void* stream) {
  SER_REQUIRE(dy && x && mean && rstd && gamma && workspace && B > 0 && L > 0 && Ls >= L && C > 0, "colnorm_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  double* s1 = (double*)workspace;
  double* s2 = s1 + (size_t)B * C;
  SER_CHECK_HIP(hipMemsetAsync(workspace, 0, (size_t)2 * B * C * sizeof(double), st));
  const int rpb = 512;
  hipLaunchKernelGGL(colnorm_bwd_stats_kernel, dim3(ceil_div(C, 64), ceil_div(L, rpb), B), dim3(256), 0, st, dy, x, mean, rstd, L, Ls, C,
                     rpb, s1, s2);
  const long long n = (long long)B * Ls * C;
  if (dx) hipLaunchKernelGGL(colnorm_bwd_apply_kernel, dim3(ew_grid(n)), dim3(256), 0, st, dy, x, mean, rstd, gamma, s1, s2, L, Ls, C, n, dx);
  if (dgamma && dbeta)
    hipLaunchKernelGGL(colnorm_bwd_param_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, s1, s2, B, C, accumulate, dgamma, dbeta);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_toeplitz_add(const float* dwin, int rows_win, int K, int Cg, int rows_slab, float* dslab, void* stream) {
  SER_REQUIRE(dwin && dslab && rows_win > 0 && K > 0 && Cg > 0 && rows_slab > 0, "toeplitz_add: bad argument");
  hipLaunchKernelGGL(toeplitz_add_kernel, dim3(ew_grid((long long)rows_slab * Cg)), dim3(256), 0, (hipStream_t)stream, dwin, rows_win, K,
                     Cg, rows_slab, dslab);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_embed_fwd(const int64_t* ids, const int64_t* pos, const float* wemb, const float* pemb, const float* temb,
                             int rows, int D, int vocab, int max_pos, float* e, void* stream) {
  SER_REQUIRE(ids && pos && wemb && pemb && temb && e && rows > 0 && D > 0, "embed_fwd: bad argument");
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, ids, pos, wemb, pemb, temb, rows, D, vocab,
                     max_pos, e);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

/* dwin [M, k Cin] -> dx [rows_in, Cin] (every row written); Cin % 4 == 0 */
extern "C" int ser_conv_col2im(const float* dwin, int M, int k, int s, int Cin, long long rows_in, float* dx, void* stream) {
  SER_REQUIRE(dwin && dx && M > 0 && k > 0 && s > 0 && Cin > 0 && Cin % 4 == 0 && rows_in > 0, "conv_col2im: bad argument");
  const long long n = rows_in * (Cin / 4);
  hipLaunchKernelGGL(conv_col2im_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dwin, M, k, s, Cin, rows_in, dx);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

/* dw [vocab, D], dp [max_pos, D], dt [D] must be zero-filled (or hold the gradients to accumulate into) */
extern "C" int ser_embed_bwd(const float* de, const int64_t* ids, const int64_t* pos, int rows, int D, int vocab, int max_pos,
                             int pad_id, float* dw, float* dp, float* dt, void* stream) {
  SER_REQUIRE(de && ids && pos && dw && dp && dt && rows > 0 && D > 0, "embed_bwd: bad argument");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, de, ids, pos, rows, D, vocab, max_pos, pad_id,
                     dw, dp, dt);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

/* out[b][t] = (wave[b][t] - mean_b) / sqrt(var_b + 1e-7); stats: B float2 of scratch */
extern "C" int ser_wave_normalize(const float* wave, int B, int T, float* out, void* stats, void* stream) {
  SER_REQUIRE(wave && out && stats && B > 0 && T > 0, "wave_normalize: bad argument");
  hipStream_t st = (hipStream_t)stream;
  SER_TRY(ser_launch_wave_stats(wave, B, T, stats, st));
  const long long n = (long long)B * T;
  hipLaunchKernelGGL(wave_normalize_kernel, dim3(ew_grid(n)), dim3(256), 0, st, wave, (const float2*)stats, T, n, out);
  SER_LAUNCH_CHECK();
  return SER_OK;
}
