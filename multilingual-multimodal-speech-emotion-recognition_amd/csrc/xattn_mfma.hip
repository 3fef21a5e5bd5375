// Attention core on the fp32 matrix pipe for sequences of up to 256 positions on both sides: the cross-modal attention of the
// head (ref src/models/cross_attention.py:15-26, nn.MultiheadAttention with 8 heads of 32) and the encoders' self-attention in
// the fine-tuning path (hf modeling_wav2vec2.py:438-463, modeling_xlm_roberta.py:211-250; 12 heads of 64).
//
// Same contract as the scalar kernels in head.hip (ser_xattn_fwd / ser_xattn_bwd dispatch here): q, k, v are fp32 [B*S, E]
// column blocks with their own row strides; P = softmax(q k^T / sqrt(hd) + key mask) is stored for backward, attention
// dropout multiplies P by a counter-based mask that backward regenerates; dS = P (dP - sum_j P dP) is stored between the two
// backward kernels.  What changes is the arithmetic unit: every product (q k^T, P v, dctx v^T, dS k, dS^T q, P^T dctx) is a
// chain of v_mfma_f32_16x16x4_f32 - exact fp32 multiplies, fp32 accumulation - on operands staged once per workgroup in LDS.
//
// v_mfma_f32_16x16x4_f32 operand layout (lane l, i = l % 16, g = l / 16): A[i][k = g], B[k = g][j = i], D[4 g + r][j = i] in
// accumulator register r.  A float4 along k per lane feeds four MFMAs (element e of lane group g stands for k = 4 g + e of a
// 16-wide chunk: A and B only have to agree on the order).
//
// LDS rows of [position][head_dim] operands are HD + 4 floats long: 16 lanes on consecutive rows then hit 16 different bank
// quads (float4 reads along head_dim), and the four lane groups of a scalar read (rows 4 g + e) land 16 banks apart.  The
// 16 x Sk probability tile of a wave goes through LDS once, to turn the accumulator layout into the A layout; its rows are
// Sk16 + 4 floats long (4 x odd: conflict-free both ways) and reuse the space of the first product's operand.
#include <cstdint>
#include <cstdlib>

#include "ser_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int XM_MAXS = 256;          // positions per side
constexpr int XM_MAXT = XM_MAXS / 16;

struct XDropM {
  bool on;
  unsigned long long st;
  unsigned site, thresh;
  float scale;
};
SER_DEVFN XDropM xdropm_init(const SerDropout& d) {
  XDropM x;
  x.on = d.state != nullptr && d.p > 0.f;
  x.st = x.on ? *d.state : 0ull;
  x.site = d.site; x.thresh = ser_drop_thresh(d.p); x.scale = 1.0f / (1.0f - d.p);
  return x;
}
SER_DEVFN float xdropm_mult(const XDropM& x, long long idx) {
  return x.on ? ser_drop_mult(x.st, x.site, (unsigned)idx, x.thresh, x.scale) : 1.0f;
}

SER_DEVFN void wave_lds_fence() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// [rows16][HD + 4] <- x[(b S + pos) ld + h HD + d], zero rows for pos >= S.  Eight loads are in flight per thread before the
// first LDS store: a plain load / store loop pays one memory round trip per iteration (the kernel is latency-bound here, with
// one workgroup per CU).
template <int HD>
SER_DEVFN void stage_rows(float* __restrict__ dst, const float* __restrict__ x, int ld, int b, int h, int S, int S16) {
  constexpr int C4 = HD / 4, PK = HD + 4, U = 8;
  const int n = S16 * C4;
  for (int base = threadIdx.x; base < n; base += 256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * 256;
      const int pos = idx / C4, c = idx % C4;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < n && pos < S) v[u] = *(const float4*)(x + ((long long)b * S + pos) * ld + h * HD + c * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * 256;
      if (idx < n) *(float4*)(dst + (idx / C4) * PK + (idx % C4) * 4) = v[u];
    }
  }
}

// MODE 0 (forward):  S = (q / sqrt(hd)) k^T -> P = softmax(S + mask) -> ctx = (P . dropout) v
//   a = q, x1 = k, x2 = v, P out, Pd out (optional: the dropped probabilities, read by backward for dv), out = ctx
// MODE 1 (backward, query side):  dP = (dctx v^T) . dropout -> dS = P (dP - sum_j P dP) -> dq = dS k / sqrt(hd)
//   a = dctx, x1 = v, x2 = k, P in, dS out, out = dq
// grid (ceil(Sq / 64), heads, B), 4 waves x 16 query rows.  dynamic LDS: region 1 (x1, then the waves' tiles) + region 2 (x2)
template <int HD, int MODE>
__global__ __launch_bounds__(256) void xattn_rows_mfma_kernel(const float* __restrict__ a, int lda, const float* __restrict__ x1, int ld1,
                                                              const float* __restrict__ x2, int ld2, const float* __restrict__ kmask,
                                                              int Sq, int Sk, int heads, float* __restrict__ P, float* __restrict__ Pd,
                                                              float* __restrict__ dS, float* __restrict__ out, int ldo, SerDropout drop,
                                                              int r1_words) {
  extern __shared__ float xm_lds[];
  constexpr int PK = HD + 4, NC = HD / 16;
  const XDropM xd = xdropm_init(drop);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, row0 = blockIdx.x * 64 + w * 16;
  const int Sk16 = (Sk + 15) & ~15, nt = Sk16 >> 4, PP = Sk16 + 4;
  float* X1 = xm_lds;
  float* X2 = xm_lds + r1_words;
  float* tile = xm_lds + w * 16 * PP;                 // this wave's 16 x Sk16 tile (region 1, after the first product)
  const float scale = 1.0f / sqrtf((float)HD);
  stage_rows<HD>(X1, x1, ld1, b, h, Sk, Sk16);
  stage_rows<HD>(X2, x2, ld2, b, h, Sk, Sk16);
  // A fragments of the first product: row j of the wave's tile, head dims 16 c + 4 g .. + 3
  float4 af[NC];
  {
    const int ar = min(row0 + j, Sq - 1);
    const float* ap = a + ((long long)b * Sq + ar) * lda + h * HD + g * 4;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      af[c] = *(const float4*)(ap + c * 16);
      if (MODE == 0) { af[c].x *= scale; af[c].y *= scale; af[c].z *= scale; af[c].w *= scale; }
    }
  }
  __syncthreads();
  f32x4 acc[XM_MAXT];
#pragma unroll
  for (int t = 0; t < XM_MAXT; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < nt) {
      const float* bp = X1 + (t * 16 + j) * PK + g * 4;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float4 bf = *(const float4*)(bp + c * 16);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c].x, bf.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c].y, bf.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c].z, bf.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c].w, bf.w, acc[t], 0, 0, 0);
      }
    }
  }
  // element (r, t) of this lane: query row row0 + 4 g + r, key 16 t + j
  const long long prow0 = ((long long)b * heads + h) * Sq;
  if (MODE == 0) {
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < XM_MAXT; ++t)
      if (t < nt) {
        const int key = t * 16 + j;
        const bool valid = key < Sk && (!kmask || kmask[(long long)b * Sk + key] != 0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s = valid ? acc[t][r] : -INFINITY;
          acc[t][r] = s;
          mx[r] = fmaxf(mx[r], s);
        }
      }
    float inv[4], mu[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m = row16_max(mx[r]);
      mu[r] = m == -INFINITY ? 0.f : m;
    }
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < XM_MAXT; ++t)
      if (t < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[t][r] = expf(acc[t][r] - mu[r]);
          sum[r] += acc[t][r];
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = row16_sum(sum[r]);
      inv[r] = s > 0.f ? 1.0f / s : 0.f;
    }
    __syncthreads();                                   // every wave is done with x1: region 1 becomes the tiles
#pragma unroll
    for (int t = 0; t < XM_MAXT; ++t)
      if (t < nt) {
        const int key = t * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = row0 + 4 * g + r;
          float pd = 0.f;
          if (i < Sq && key < Sk) {
            const long long gi = (prow0 + i) * Sk + key;
            const float p = acc[t][r] * inv[r];
            pd = p * xdropm_mult(xd, gi);
            P[gi] = p;
            if (Pd) Pd[gi] = pd;
          }
          tile[(4 * g + r) * PP + key] = pd;
        }
      }
  } else {
    float dot[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 pp[XM_MAXT];
#pragma unroll
    for (int t = 0; t < XM_MAXT; ++t) {
      pp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < nt) {
        const int key = t * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = row0 + 4 * g + r;
          float p = 0.f, dp = 0.f;
          if (i < Sq && key < Sk) {
            const long long gi = (prow0 + i) * Sk + key;
            p = P[gi];
            dp = acc[t][r] * xdropm_mult(xd, gi);
          }
          pp[t][r] = p;
          acc[t][r] = dp;
          dot[r] = fmaf(p, dp, dot[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dot[r] = row16_sum(dot[r]);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < XM_MAXT; ++t)
      if (t < nt) {
        const int key = t * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = row0 + 4 * g + r;
          const float gs = pp[t][r] * (acc[t][r] - dot[r]);
          if (i < Sq && key < Sk) dS[(prow0 + i) * Sk + key] = gs;
          tile[(4 * g + r) * PP + key] = (i < Sq && key < Sk) ? gs : 0.f;
        }
      }
  }
  wave_lds_fence();                                    // the tile is read by the wave that wrote it
  // second product: out[16 x HD] = tile[16 x Sk16] . x2[Sk16 x HD]
  f32x4 o[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kk = 0; kk < nt; ++kk) {
    const float4 pa = *(const float4*)(tile + j * PP + kk * 16 + g * 4);
    const float* bp = X2 + (kk * 16 + g * 4) * PK + j;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.x, bp[c * 16], o[c], 0, 0, 0);
      o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.y, bp[PK + c * 16], o[c], 0, 0, 0);
      o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.z, bp[2 * PK + c * 16], o[c], 0, 0, 0);
      o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.w, bp[3 * PK + c * 16], o[c], 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = row0 + 4 * g + r;
    if (i < Sq) {
      float* op = out + ((long long)b * Sq + i) * ldo + h * HD + j;
#pragma unroll
      for (int c = 0; c < NC; ++c) op[c * 16] = MODE == 0 ? o[c][r] : o[c][r] * scale;
    }
  }
}

// Backward, key side: dk_j = sum_i dS_ij q_i / sqrt(hd), dv_j = sum_i Pv_ij dctx_i (Pv: the dropped probabilities under
// dropout).  grid (ceil(Sk / 64), heads, B), 4 waves x 16 keys; the [Sq16][64] panel of dS (then Pv) and the [Sq16][HD]
// rows of q (then dctx) go through LDS.
template <int HD>
__global__ __launch_bounds__(256) void xattn_bwd_kv_mfma_kernel(const float* __restrict__ dctx, int ldc, const float* __restrict__ q, int ldq,
                                                                const float* __restrict__ Pv, const float* __restrict__ dS, int Sq, int Sk,
                                                                int heads, float* __restrict__ dk, int lddk, float* __restrict__ dv,
                                                                int lddv) {
  extern __shared__ float xm_lds[];
  constexpr int PK = HD + 4, NC = HD / 16, PS = 68;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, key0 = blockIdx.x * 64;
  const int Sq16 = (Sq + 15) & ~15, nq = Sq16 >> 4;
  float* St = xm_lds;                                  // [Sq16][68]: column c = key key0 + c
  float* Xs = xm_lds + Sq16 * PS;                      // [Sq16][HD + 4]
  const float scale = 1.0f / sqrtf((float)HD);
  const long long prow0 = ((long long)b * heads + h) * Sq;
#pragma unroll
  for (int phase = 0; phase < 2; ++phase) {
    const float* panel = phase == 0 ? dS : Pv;
    if (phase) __syncthreads();
    {                                                  // 16 loads in flight per thread (see stage_rows)
      constexpr int U = 16;
      const int n = Sq16 * 64;
      for (int base = tid; base < n; base += 256 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = base + u * 256, i = idx >> 6, c = idx & 63;
          v[u] = (idx < n && i < Sq && key0 + c < Sk) ? panel[(prow0 + i) * Sk + key0 + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = base + u * 256;
          if (idx < n) St[(idx >> 6) * PS + (idx & 63)] = v[u];
        }
      }
    }
    stage_rows<HD>(Xs, phase == 0 ? q : dctx, phase == 0 ? ldq : ldc, b, h, Sq, Sq16);
    __syncthreads();
    f32x4 o[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (key0 + w * 16 < Sk) {                          // wave-uniform
      for (int qq = 0; qq < nq; ++qq) {
        const float* ap = St + (qq * 16 + g * 4) * PS + w * 16 + j;
        const float* bp = Xs + (qq * 16 + g * 4) * PK + j;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float av = ap[e * PS];
#pragma unroll
          for (int c = 0; c < NC; ++c) o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[e * PK + c * 16], o[c], 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + w * 16 + 4 * g + r;
        if (key < Sk) {
          float* op = (phase == 0 ? dk + ((long long)b * Sk + key) * lddk : dv + ((long long)b * Sk + key) * lddv) + h * HD + j;
#pragma unroll
          for (int c = 0; c < NC; ++c) op[c * 16] = phase == 0 ? o[c][r] * scale : o[c][r];
        }
      }
    }
  }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

int r1_words(int Sk16, int HD) {
  const int a_ = Sk16 * (HD + 4), b_ = 64 * (Sk16 + 4);
  return a_ > b_ ? a_ : b_;
}

bool mfma_enabled() {
  static const bool on = [] {
    const char* e = getenv("SER_XATTN_MFMA");
    return !(e && e[0] == '0');
  }();
  return on;
}

}  // namespace

// 1 when the MFMA kernels take this problem (head_dim 32 / 64, both sides <= 256 positions, 16-byte aligned column blocks)
int ser_xattn_mfma_ok(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* c, int ldc, int Sq, int Sk,
                      int head_dim) {
  if (!mfma_enabled()) return 0;
  if (!(head_dim == 32 || head_dim == 64) || Sq > XM_MAXS || Sk > XM_MAXS || Sq < 1 || Sk < 1) return 0;
  if ((ldq | ldk | ldv | ldc) & 3) return 0;
  return aligned16(q) && aligned16(k) && aligned16(v) && aligned16(c);
}

int ser_launch_xattn_fwd_mfma(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* key_mask, int B,
                              int Sq, int Sk, int heads, int head_dim, float* P, float* ctx, int ldc, SerDropout drop, float* P_dropped,
                              hipStream_t st) {
  const int Sk16 = (Sk + 15) & ~15;
  const int r1 = r1_words(Sk16, head_dim);
  const size_t lds = (size_t)(r1 + Sk16 * (head_dim + 4)) * sizeof(float);
  const dim3 grid(ceil_div(Sq, 64), heads, B);
  if (head_dim == 32)
    hipLaunchKernelGGL((xattn_rows_mfma_kernel<32, 0>), grid, dim3(256), lds, st, q, ldq, k, ldk, v, ldv, key_mask, Sq, Sk, heads, P,
                       P_dropped, (float*)nullptr, ctx, ldc, drop, r1);
  else
    hipLaunchKernelGGL((xattn_rows_mfma_kernel<64, 0>), grid, dim3(256), lds, st, q, ldq, k, ldk, v, ldv, key_mask, Sq, Sk, heads, P,
                       P_dropped, (float*)nullptr, ctx, ldc, drop, r1);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

int ser_launch_xattn_bwd_mfma(const float* dctx, int ldc, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                              const float* P, const float* Pv, int B, int Sq, int Sk, int heads, int head_dim, float* dS, float* dq,
                              int lddq, float* dk, int lddk, float* dv, int lddv, SerDropout drop, hipStream_t st) {
  const int Sk16 = (Sk + 15) & ~15, Sq16 = (Sq + 15) & ~15;
  const int r1 = r1_words(Sk16, head_dim);
  const size_t lds_q = (size_t)(r1 + Sk16 * (head_dim + 4)) * sizeof(float);
  const size_t lds_kv = (size_t)(Sq16 * 68 + Sq16 * (head_dim + 4)) * sizeof(float);
  const dim3 gq(ceil_div(Sq, 64), heads, B), gk(ceil_div(Sk, 64), heads, B);
  if (head_dim == 32) {
    hipLaunchKernelGGL((xattn_rows_mfma_kernel<32, 1>), gq, dim3(256), lds_q, st, dctx, ldc, v, ldv, k, ldk, (const float*)nullptr, Sq, Sk,
                       heads, (float*)P, (float*)nullptr, dS, dq, lddq, drop, r1);
    hipLaunchKernelGGL(xattn_bwd_kv_mfma_kernel<32>, gk, dim3(256), lds_kv, st, dctx, ldc, q, ldq, Pv, (const float*)dS, Sq, Sk, heads, dk,
                       lddk, dv, lddv);
  } else {
    hipLaunchKernelGGL((xattn_rows_mfma_kernel<64, 1>), gq, dim3(256), lds_q, st, dctx, ldc, v, ldv, k, ldk, (const float*)nullptr, Sq, Sk,
                       heads, (float*)P, (float*)nullptr, dS, dq, lddq, drop, r1);
    hipLaunchKernelGGL(xattn_bwd_kv_mfma_kernel<64>, gk, dim3(256), lds_kv, st, dctx, ldc, q, ldq, Pv, (const float*)dS, Sq, Sk, heads, dk,
                       lddk, dv, lddv);
  }
  SER_LAUNCH_CHECK();
  return SER_OK;
}
