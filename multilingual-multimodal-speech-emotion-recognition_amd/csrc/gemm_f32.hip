// Exact-fp32 GEMM on the matrix cores for the trainable head (gfx950).
//
//   C[M,N] (+)= act(A . B + bias) + residual,   A(m,k) = a[m*sam + k*sak],  B(k,n) = b[k*sbk + n*sbn]
//
// v_mfma_f32_16x16x4_f32 takes fp32 operands and is bit-for-bit an fp32 fmaf chain, so the head
// (adapters, cross-attention projections, pooling, fusion, the 35-block classifier and all of
// their backward products) keeps fp32 parity with the reference while still running on MFMA.
// Arbitrary element strides cover the three products of a Linear layer with one kernel:
//   forward  y = x W^T      : A = x (sam=K,sak=1)   B = W^T (sbk=1,sbn=K)
//   dgrad    dx = dy W      : A = dy (sam=N,sak=1)  B = W   (sbk=K,sbn=1)
//   wgrad    dW += dy^T x   : A = dy^T (sam=1,sak=N) B = x  (sbk=K,sbn=1), accumulate
// Tiles are staged through LDS k-major (As[k][m], Bs[k][n]) so the MFMA operand reads
// (A[m = lane&15][k = lane>>4]) are conflict-free ds_read_b32; global loads are register-prefetched
// one k-tile ahead, one barrier per k-tile.
#include <vector>
#include "ser_common.h"

struct SerGemmF32Args {
  const float *a, *b;
  float* c;
  int M, N, K;
  long long sam, sak, sbk, sbn;
  int ldc;
  const float* bias;
  int act;
  const float* residual;
  int ldr;
  int accumulate;
  // split-K (wgrad over many tokens): grid.z slices of k_chunk; partial tiles go to ws[z][M][N] and the
  // row sums of A (= bias gradient when A = dy^T) to ws_rowsum[z][M]; a reduce kernel finishes
  int k_chunk;
  float* ws;
  float* ws_rowsum;
  int vec_a, vec_b;   // set by the launcher: the contiguous axis of a / b is 16-byte aligned per float4 group
  int products;       // MFMA products per multiply on the split-bf16 path: 3 (hi*hi + lo*hi + hi*lo, ~fp32) or 1 (bf16 operands)
};

namespace {

SER_DEVFN float act_f32(float v, int act) {
  switch (act) {
    case SER_ACT_GELU: return gelu_erf(v);
    case SER_ACT_RELU: return fmaxf(v, 0.f);
    case SER_ACT_TANH: return tanhf(v);
    case SER_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

// R = rows of the tile in its non-k dimension (BM for A, BN for B), KT = k extent of the tile.
// Each thread moves R*KT/256 elements as float4 groups along the operand's contiguous axis.
template <int R, int KT>
struct TileRegs {
  float4 v[R * KT / 1024];
};

// element (r, k) of the operand lives at p[r*sr + k*sk]; rows r0.. (< rmax), k from k0 (< kmax).
// Loads are UNCONDITIONAL (out-of-range groups read offset 0 and are zeroed by a select): a branch
// around a load makes hipcc wait vmcnt(0) at every join and serialises the whole prefetch.
// vec: every float4 group along the contiguous axis is 16-byte aligned and entirely in or out of range.
template <int R, int KT, bool FAST = false>
SER_DEVFN void load_tile(TileRegs<R, KT>& t, const float* __restrict__ p, long long sr, long long sk, int r0, int rmax,
                         int k0, int kmax, int tid, bool kfast, bool vec) {
  constexpr int NG = R * KT / 1024;
#pragma unroll
  for (int e = 0; e < NG; ++e) {
    const int g = tid + 256 * e;        // float4 group index; consecutive threads walk the contiguous axis
    int r, k;
    if (kfast) { r = g / (KT / 4); k = (g % (KT / 4)) * 4; } else { k = g / (R / 4); r = (g % (R / 4)) * 4; }
    const int gr = r0 + r, gk = k0 + k;
    const long long base = (long long)gr * sr + (long long)gk * sk;
    const long long step = kfast ? sk : sr;          // == 1 along the contiguous axis
    const int lim = kfast ? kmax - gk : rmax - gr;   // valid elements of this group along that axis
    const bool ok = kfast ? gr < rmax : gk < kmax;
    float4 v;
    if (FAST) {
      // no k tail (host guarantees it) and rows clamped into range: rows beyond the matrix only feed
      // accumulators that are never stored, so no select may sit between the load and the MFMAs
      const int cr = kfast ? min(gr, rmax - 1) : min(gr, rmax - 4);
      v = *(const float4*)(p + (long long)cr * sr + (long long)gk * sk);
    } else if (vec) {
      const bool in = ok && lim >= 4;
      v = *(const float4*)(p + (in ? base : 0));
      if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float x0 = p[(ok && lim > 0) ? base : 0];
      const float x1 = p[(ok && lim > 1) ? base + step : 0];
      const float x2 = p[(ok && lim > 2) ? base + 2 * step : 0];
      const float x3 = p[(ok && lim > 3) ? base + 3 * step : 0];
      v.x = (ok && lim > 0) ? x0 : 0.f;
      v.y = (ok && lim > 1) ? x1 : 0.f;
      v.z = (ok && lim > 2) ? x2 : 0.f;
      v.w = (ok && lim > 3) ? x3 : 0.f;
    }
    t.v[e] = v;
  }
}

// LDS image of a tile: k-fast operands are kept [row][k] with pitch KT+2 (fragment read = 16 rows x 2 k
// per half-wave -> banks 2*row + k: conflict-free), row-fast operands [k][row] with pitch R+16.
template <int R, int KT, int LD>
SER_DEVFN void store_tile(const TileRegs<R, KT>& t, float* s, int tid, bool kfast) {
  constexpr int NG = R * KT / 1024;
#pragma unroll
  for (int e = 0; e < NG; ++e) {
    const int g = tid + 256 * e;
    if (kfast) {
      const int r = g / (KT / 4), k = (g % (KT / 4)) * 4;
      float2* d = (float2*)(s + r * (KT + 2) + k);     // pitch KT+2 floats: 8-byte aligned, not 16
      d[0] = make_float2(t.v[e].x, t.v[e].y);
      d[1] = make_float2(t.v[e].z, t.v[e].w);
    } else {
      const int k = g / (R / 4), r = (g % (R / 4)) * 4;
      *(float4*)(s + k * LD + r) = t.v[e];
    }
  }
}

// AKF / BKF: 1 = operand is k-contiguous, 0 = row-contiguous, 2 = decide at run time; VEC likewise.
// Compile-time layouts matter: with run-time selects hipcc branches around the loads and drains vmcnt(0)
// at every join, which serialises the register prefetch.
template <int BM, int BN, int FBK, int AKF = 2, int BKF = 2, int VEC = 2>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const SerGemmF32Args g) {
  constexpr int LDA = BM == 64 ? 80 : 16, LDB = BN == 64 ? 80 : 16;   // row-fast pitch, (LD % 32) == 16: no conflict
  constexpr int LDK = FBK + 2;                                          // k-fast pitch
  constexpr int ASZ = (BM * LDK > FBK * LDA) ? BM * LDK : FBK * LDA;
  constexpr int BSZ = (BN * LDK > FBK * LDB) ? BN * LDK : FBK * LDB;
  constexpr int WAVES_M = BN == 16 ? 4 : (BM == 64 ? 2 : 1), WAVES_N = 4 / WAVES_M;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
  __shared__ __attribute__((aligned(16))) float As[2][ASZ];
  __shared__ __attribute__((aligned(16))) float Bs[2][BSZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const bool a_kfast = AKF == 2 ? g.sak == 1 : AKF == 1, b_kfast = BKF == 2 ? g.sbk == 1 : BKF == 1;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  TileRegs<BM, FBK> ra;
  TileRegs<BN, FBK> rb;
  const bool avec = VEC == 2 ? g.vec_a != 0 : VEC == 1, bvec = VEC == 2 ? g.vec_b != 0 : VEC == 1;
  const int kbeg = g.k_chunk ? blockIdx.z * g.k_chunk : 0;
  const int kend = g.k_chunk ? min(g.K, kbeg + g.k_chunk) : g.K;
  const int nk = (kend - kbeg + FBK - 1) / FBK;
  float rowsum = 0.f;
  const bool want_rowsum = g.ws_rowsum != nullptr && blockIdx.x == 0 && tid < BM;
  load_tile<BM, FBK, VEC == 1>(ra, g.a, g.sam, g.sak, m0, g.M, kbeg, kend, tid, a_kfast, avec);
  load_tile<BN, FBK, VEC == 1>(rb, g.b, g.sbn, g.sbk, n0, g.N, kbeg, kend, tid, b_kfast, bvec);
  store_tile<BM, FBK, LDA>(ra, As[0], tid, a_kfast);
  store_tile<BN, FBK, LDB>(rb, Bs[0], tid, b_kfast);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_tile<BM, FBK, VEC == 1>(ra, g.a, g.sam, g.sak, m0, g.M, kbeg + (kt + 1) * FBK, kend, tid, a_kfast, avec);
      load_tile<BN, FBK, VEC == 1>(rb, g.b, g.sbn, g.sbk, n0, g.N, kbeg + (kt + 1) * FBK, kend, tid, b_kfast, bvec);
    }
    if (want_rowsum) {
#pragma unroll
      for (int kk = 0; kk < FBK; ++kk) rowsum += a_kfast ? As[cur][tid * LDK + kk] : As[cur][kk * LDA + tid];
    }
#pragma unroll
    for (int ks = 0; ks < FBK / 4; ++ks) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WM + i * 16 + fr, k = ks * 4 + fq;
        af[i] = a_kfast ? As[cur][row * LDK + k] : As[cur][k * LDA + row];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WN + j * 16 + fr, k = ks * 4 + fq;
        bf[j] = b_kfast ? Bs[cur][row * LDK + k] : Bs[cur][k * LDB + row];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_tile<BM, FBK, LDA>(ra, As[cur ^ 1], tid, a_kfast);
      store_tile<BN, FBK, LDB>(rb, Bs[cur ^ 1], tid, b_kfast);
    }
    __syncthreads();
  }

  if (want_rowsum && m0 + tid < g.M) g.ws_rowsum[(long long)blockIdx.z * g.M + m0 + tid] = rowsum;
  if (g.ws) {   // split-K partial: raw accumulators to the workspace slice
    float* wsz = g.ws + (long long)blockIdx.z * g.M * g.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + fr;
      if (n >= g.N) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * WM + i * 16 + fq * 4 + r;
          if (m < g.M) wsz[(long long)m * g.N + n] = acc[i][j][r];
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 16 + fr;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * WM + i * 16 + fq * 4 + r;
        if (m >= g.M) continue;
        float v = act_f32(acc[i][j][r] + bv, g.act);
        if (g.residual) v += g.residual[(long long)m * g.ldr + n];
        float* cp = g.c + (long long)m * g.ldc + n;
        *cp = g.accumulate ? *cp + v : v;
      }
  }
}


// ------------------------------------------------------------------------------------------
// Skinny products for M <= 16 rows (the 35-block classifier, fusion and heads run at M = batch):
// weights are streamed straight from global memory into MFMA operand registers, 16 B per lane,
// every load of a wave issued up front; K (or N) is split over the waves of the workgroup and
// reduced through LDS.  No operand tile goes through LDS: each weight byte is used once.
// ------------------------------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void skinny_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ bias, int act,
                                                                const float* __restrict__ residual, int ldr,
                                                                float* __restrict__ y, int M, int N, int K) {
  __shared__ float red[WAVES][64][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const float* xr = x + (long long)min(i, M - 1) * K + q * 4;
  const float* wr = W + (long long)min(n0 + i, N - 1) * K + q * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nchunk = K >> 4;
  const int nit = (nchunk - w + WAVES - 1) / WAVES;   // this wave's chunks: w, w+WAVES, ...
  for (int it = 0; it < nit; it += 4) {               // 8 x 16-B loads in flight per lane before the MFMAs
    float4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                     // unconditional loads from a clamped chunk (no branch, no early wait)
      const int c = min(w + (it + u) * WAVES, nchunk - 1);
      a[u] = *(const float4*)(xr + c * 16);
      b[u] = *(const float4*)(wr + c * 16);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (it + u >= nit) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][lane][r] = acc[r];
  __syncthreads();
  if (w == 0) {
    const int n = n0 + i;
    if (n < N) {
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = q * 4 + r;
        if (m >= M) continue;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) v += red[ww][lane][r];
        v = act_f32(v + bv, act);
        if (residual) v += residual[(long long)m * ldr + n];
        y[(long long)m * N + n] = v;
      }
    }
  }
}


// skinny forward with the two chained LayerNorms of a classifier block as prologue:
//   x1 = LN(x; g1,b1), u = LN(x1; g2,b2), y = act(u W^T + bias)      (ref classifier.py:209-210 + block[0..2])
// Every workgroup recomputes the row statistics of the <= 16 rows (16 x K floats, L2-resident); workgroup 0 also
// writes x1, u and the statistics for the residual add and for backward.  Saves one launch per residual block.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void skinny_fwd_ln2_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                    const float* __restrict__ bias, int act,
                                                                    const float* __restrict__ g1, const float* __restrict__ b1,
                                                                    const float* __restrict__ g2, const float* __restrict__ b2,
                                                                    float eps, float* __restrict__ y1, float* __restrict__ y2,
                                                                    float* __restrict__ stats, float* __restrict__ y, int M,
                                                                    int N, int K) {
  __shared__ float red[WAVES][64][4];
  __shared__ float st[16][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nch4 = K >> 2;                           // float4 chunks per row (K <= 512 -> <= 2 per lane)
  for (int row = w; row < 16; row += WAVES) {        // prologue: statistics of both LayerNorms, one wave per row
    const int rr = min(row, M - 1);
    float4 v[2];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = lane + 64 * e;
      v[e] = c < nch4 ? *(const float4*)(x + (long long)rr * K + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (v[e].x + v[e].y) + (v[e].z + v[e].w);
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const float* gp = pass == 0 ? g1 : g2;
      const float* bp = pass == 0 ? b1 : b2;
      const float mean = wave_sum(s) / (float)K;
      float qq = 0.f;
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (lane + 64 * e < nch4) {
          const float a = v[e].x - mean, b = v[e].y - mean, c2 = v[e].z - mean, d = v[e].w - mean;
          qq += (a * a + b * b) + (c2 * c2 + d * d);
        }
      const float rstd = 1.0f / sqrtf(wave_sum(qq) / (float)K + eps);
      if (lane == 0) { st[row][2 * pass] = mean; st[row][2 * pass + 1] = rstd; }
      s = 0.f;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = lane + 64 * e;
        if (c < nch4) {
          const float4 gm = *(const float4*)(gp + c * 4), bt = *(const float4*)(bp + c * 4);
          float4 o;
          o.x = (v[e].x - mean) * rstd * gm.x + bt.x; o.y = (v[e].y - mean) * rstd * gm.y + bt.y;
          o.z = (v[e].z - mean) * rstd * gm.z + bt.z; o.w = (v[e].w - mean) * rstd * gm.w + bt.w;
          if (blockIdx.x == 0 && row < M) *(float4*)((pass == 0 ? y1 : y2) + (long long)row * K + c * 4) = o;
          v[e] = o;
          s += (o.x + o.y) + (o.z + o.w);
        }
      }
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const int row = threadIdx.x >> 2, which = threadIdx.x & 3;
    if (row < M) stats[which * M + row] = st[row][which];
  }
  const float m1 = st[i][0], r1 = st[i][1], m2 = st[i][2], r2 = st[i][3];
  const float* xr = x + (long long)min(i, M - 1) * K + q * 4;
  const float* wr = W + (long long)min(n0 + i, N - 1) * K + q * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nchunk = K >> 4;
  const int nit = (nchunk - w + WAVES - 1) / WAVES;
  for (int it = 0; it < nit; it += 4) {
    float4 a[4], b[4], G1[4], B1[4], G2[4], B2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = min(w + (it + u) * WAVES, nchunk - 1);
      const int ko = c * 16 + q * 4;
      a[u] = *(const float4*)(xr + c * 16);
      b[u] = *(const float4*)(wr + c * 16);
      G1[u] = *(const float4*)(g1 + ko); B1[u] = *(const float4*)(b1 + ko);
      G2[u] = *(const float4*)(g2 + ko); B2[u] = *(const float4*)(b2 + ko);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float4 t;
      t.x = (((a[u].x - m1) * r1 * G1[u].x + B1[u].x) - m2) * r2 * G2[u].x + B2[u].x;
      t.y = (((a[u].y - m1) * r1 * G1[u].y + B1[u].y) - m2) * r2 * G2[u].y + B2[u].y;
      t.z = (((a[u].z - m1) * r1 * G1[u].z + B1[u].z) - m2) * r2 * G2[u].z + B2[u].z;
      t.w = (((a[u].w - m1) * r1 * G1[u].w + B1[u].w) - m2) * r2 * G2[u].w + B2[u].w;
      if (it + u >= nit) t = make_float4(0.f, 0.f, 0.f, 0.f);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.w, b[u].w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][lane][r] = acc[r];
  __syncthreads();
  if (w == 0) {
    const int n = n0 + i;
    if (n < N) {
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = q * 4 + r;
        if (m >= M) continue;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) v += red[ww][lane][r];
        y[(long long)m * N + n] = act_f32(v + bv, act);
      }
    }
  }
}

// dx[M,Kc] = (dy[M,N] . W[N,Kc]) * relu'(mask)   (reduction over N split across the waves; mask = the
// ReLU OUTPUT of the layer below, or NULL: fuses the activation backward into the dgrad epilogue)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void skinny_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                                  const float* __restrict__ relu_mask,
                                                                  float* __restrict__ dx, int M, int N, int Kc,
                                                                  int accumulate) {
  __shared__ float red[WAVES][64][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int c0 = blockIdx.x * 64 + 4 * i;
  const bool cok = c0 < Kc;
  const int cc = cok ? c0 : 0;                       // clamped column group: loads stay unconditional
  const float* dyr = dy + (long long)min(i, M - 1) * N;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = (N + 3) >> 2;
  const int nit = (nsteps - w + WAVES - 1) / WAVES;
  for (int it = 0; it < nit; it += 8) {              // 8 weight rows (16 B per lane each) in flight per lane
    float a[8];
    float4 b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = min((w + (it + u) * WAVES) * 4 + q, N - 1);
      a[u] = dyr[n];
      b[u] = *(const float4*)(W + (long long)n * Kc + cc);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = (w + (it + u) * WAVES) * 4 + q;
      const float av = (it + u < nit && n < N) ? a[u] : 0.f;
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u].z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u].w, acc[3], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][lane][t * 4 + r] = acc[t][r];
  __syncthreads();
  if (w == 0 && cok) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = q * 4 + r;
      if (m >= M) continue;
      float v[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        v[t] = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) v[t] += red[ww][lane][t * 4 + r];
      }
      float4 o = make_float4(v[0], v[1], v[2], v[3]);
      if (relu_mask) {
        const float4 mk = *(const float4*)(relu_mask + (long long)m * Kc + c0);
        o.x = mk.x > 0.f ? o.x : 0.f; o.y = mk.y > 0.f ? o.y : 0.f; o.z = mk.z > 0.f ? o.z : 0.f; o.w = mk.w > 0.f ? o.w : 0.f;
      }
      float4* dst = (float4*)(dx + (long long)m * Kc + c0);
      if (accumulate) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
      *dst = o;
    }
  }
}


// 16-column variant of the skinny dgrad: four times as many workgroups pull the weight matrix (the M = batch
// classifier is bound by how many CUs stream weights, not by FLOPs); one 4-byte load per lane per weight row.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void skinny_dgrad16_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                                    const float* __restrict__ relu_mask,
                                                                    float* __restrict__ dx, int M, int N, int Kc,
                                                                    int accumulate) {
  __shared__ float red[WAVES][64][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int c = blockIdx.x * 16 + i;
  const bool cok = c < Kc;
  const int cc = cok ? c : 0;
  const float* dyr = dy + (long long)min(i, M - 1) * N;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nsteps = (N + 3) >> 2;
  const int nit = (nsteps - w + WAVES - 1) / WAVES;
  for (int it = 0; it < nit; it += 16) {
    float a[16], b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int n = min((w + (it + u) * WAVES) * 4 + q, N - 1);
      a[u] = dyr[n];
      b[u] = W[(long long)n * Kc + cc];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int n = (w + (it + u) * WAVES) * 4 + q;
      const float av = (it + u < nit && n < N) ? a[u] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][lane][r] = acc[r];
  __syncthreads();
  if (w == 0 && cok) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = q * 4 + r;
      if (m >= M) continue;
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < WAVES; ++ww) v += red[ww][lane][r];
      if (relu_mask) v = relu_mask[(long long)m * Kc + c] > 0.f ? v : 0.f;
      float* dst = dx + (long long)m * Kc + c;
      *dst = accumulate ? *dst + v : v;
    }
  }
}

// dW[N,Kc] (+)= dy[M<=16,N]^T . x[M,Kc] ; db[N] (+)= sum_m dy[m,n]
struct SkinnyWgradArgs {
  const float *dy, *x;
  float *dW, *db;
  int N, Kc;
};

__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const SkinnyWgradArgs p0, const SkinnyWgradArgs p1, int M,
                                                           int accumulate) {
  const SkinnyWgradArgs& P = blockIdx.z == 0 ? p0 : p1;   // up to two independent problems per launch
  const float* __restrict__ dy = P.dy;
  const float* __restrict__ x = P.x;
  float* __restrict__ dW = P.dW;
  float* __restrict__ db = P.db;
  const int N = P.N, Kc = P.Kc;
  if ((int)blockIdx.y * 64 >= N || (int)blockIdx.x * 64 >= Kc) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.y * 64 + w * 16;
  const int c0 = blockIdx.x * 64 + 4 * i;
  const bool cok = c0 < Kc, nok = n0 + i < N;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float asum = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int m = 4 * s + q;
    const bool mok = m < M;
    const float a = (mok && nok) ? dy[(long long)m * N + n0 + i] : 0.f;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mok && cok) b = *(const float4*)(x + (long long)m * Kc + c0);
    asum += a;
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.y, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.z, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.w, acc[3], 0, 0, 0);
  }
  if (cok) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + q * 4 + r;
      if (n >= N) continue;
      float4* dst = (float4*)(dW + (long long)n * Kc + c0);
      float4 o = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
      if (accumulate) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
      *dst = o;
    }
  }
  if (db && blockIdx.x == 0) {
    asum += __shfl_xor(asum, 16, 64);
    asum += __shfl_xor(asum, 32, 64);
    if (q == 0 && nok) db[n0 + i] = accumulate ? db[n0 + i] + asum : asum;
  }
}

// C[m,n] (+)= sum_z ws[z][m][n] ; rowsum_out[m] (+)= sum_z ws_rowsum[z][m]
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ ws_rowsum, int splits,
                                     long long MN, int M, float* __restrict__ c, float* __restrict__ rowsum_out,
                                     int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < MN) {
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += ws[(long long)z * MN + i];
    c[i] = accumulate ? c[i] + v : v;
  }
  if (rowsum_out && i < M) {
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += ws_rowsum[(long long)z * M + i];
    rowsum_out[i] = accumulate ? rowsum_out[i] + v : v;
  }
}

// ------------------------------------------------------------------------------------------
// fp32-in / fp32-out GEMM on the bf16 matrix cores with split operands (three products per MAC, ~2^-16 relative
// error per product): the token-level products of the head (adapters, cross-attention projections, pooling
// scorer; M = B*S rows) are bound by the fp32 MFMA rate (1/16 of bf16) in gemm_f32_kernel, so here each staged
// fp32 tile is split into hi/lo bf16 planes on its way into LDS (v_cvt_pk_bf16_f32) and multiplied with
// v_mfma_f32_16x16x32_bf16.  Same argument record, layouts (AKF/BKF), split-K and fused bias-gradient row sums as
// gemm_f32_kernel.  64x64 tile, BK = 32; LDS planes are [row][32 bf16] with the 16-byte chunk index XORed by
// 2*bit3(row) (conflict-free ds_read_b128 fragments).
// ------------------------------------------------------------------------------------------
constexpr int X3_BK = 64;
// LDS plane = 64 rows x 128 B (64 bf16); 16-byte chunk index XOR (row & 7): conflict-free ds_read_b128 fragments
SER_DEVFN int x3_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

constexpr int X3_PLANE = 64 * 128;                   // bytes
// NP = 3: A and B as hi/lo planes, three MFMAs per product.  NP = 1: hi planes only (operands rounded to bf16, fp32
// accumulation) - used for the backward products of the token-level head GEMMs in the `bf16` precision mode.
template <int AKF, int BKF, int NP = 3>
SER_DEVFN void x3_body(const SerGemmF32Args& g, const int bx, const int by, const int bz, char (*lds)[(NP == 3 ? 4 : 2) * X3_PLANE]) {
  constexpr int PLANE = X3_PLANE;
  constexpr int BOFF = (NP == 3 ? 2 : 1) * PLANE;       // first plane of B inside a stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = by * 64, n0 = bx * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int kbeg = g.k_chunk ? bz * g.k_chunk : 0;
  const int kend = g.k_chunk ? min(g.K, kbeg + g.k_chunk) : g.K;
  const int nk = (kend - kbeg + X3_BK - 1) / X3_BK;   // a k tail (only row-contiguous operands, i.e. wgrad) is zero-filled
  const bool has_tail = ((kend - kbeg) % X3_BK) != 0;
  const bool want_rowsum = g.ws_rowsum != nullptr && bx == 0 && tid < 64;
  float rowsum = 0.f;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 ra[4], rb[4];      // 8 x 16-byte loads in flight per lane: one k-tile of latency per 64 k, not per 16
  auto load = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int gi = tid + 256 * e;
      if (AKF) { const int r = min(m0 + gi / 16, g.M - 1), k = (gi % 16) * 4; ra[e] = *(const float4*)(g.a + (long long)r * g.sam + k0 + k); }
      else { const int k = min(k0 + gi / 16, kend - 1), r = min(m0 + (gi % 16) * 4, g.M - 4); ra[e] = *(const float4*)(g.a + (long long)k * g.sak + r); }
      if (BKF) { const int r = min(n0 + gi / 16, g.N - 1), k = (gi % 16) * 4; rb[e] = *(const float4*)(g.b + (long long)r * g.sbn + k0 + k); }
      else { const int k = min(k0 + gi / 16, kend - 1), r = min(n0 + (gi % 16) * 4, g.N - 4); rb[e] = *(const float4*)(g.b + (long long)k * g.sbk + r); }
    }
  };
  auto store_one = [&](char* hi, char* lo, const float4 v, int gi, bool kf) {
    uint32_t h0, l0 = 0, h1, l1 = 0;
    if (NP == 3) {
      split_bf16x2(v.x, v.y, h0, l0);
      split_bf16x2(v.z, v.w, h1, l1);
    } else {
      h0 = pack_bf16x2(v.x, v.y);
      h1 = pack_bf16x2(v.z, v.w);
    }
    if (kf) {
      const int r = gi / 16, k = (gi % 16) * 4;
      const int off = x3_off(r, k >> 3) + (k & 7) * 2;
      *(uint2*)(hi + off) = make_uint2(h0, h1);
      if (NP == 3) *(uint2*)(lo + off) = make_uint2(l0, l1);
    } else {
      const int k = gi / 16, r = (gi % 16) * 4;
      const int kb = (k & 7) * 2, ch = k >> 3;
      *(uint16_t*)(hi + x3_off(r, ch) + kb) = (uint16_t)h0;
      *(uint16_t*)(hi + x3_off(r + 1, ch) + kb) = (uint16_t)(h0 >> 16);
      *(uint16_t*)(hi + x3_off(r + 2, ch) + kb) = (uint16_t)h1;
      *(uint16_t*)(hi + x3_off(r + 3, ch) + kb) = (uint16_t)(h1 >> 16);
      if (NP == 3) {
        *(uint16_t*)(lo + x3_off(r, ch) + kb) = (uint16_t)l0;
        *(uint16_t*)(lo + x3_off(r + 1, ch) + kb) = (uint16_t)(l0 >> 16);
        *(uint16_t*)(lo + x3_off(r + 2, ch) + kb) = (uint16_t)l1;
        *(uint16_t*)(lo + x3_off(r + 3, ch) + kb) = (uint16_t)(l1 >> 16);
      }
    }
  };
  auto store = [&](int buf, int k0) {
    char* s = lds[buf];
    if (has_tail && k0 + X3_BK > kend) {      // uniform: only the last k-tile of a ragged reduction
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + (tid + 256 * e) / 16;
        if (!AKF && k >= kend) ra[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!BKF && k >= kend) rb[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      store_one(s, s + PLANE, ra[e], tid + 256 * e, AKF != 0);                                  // NP == 1: the lo pointer is unused
      store_one(s + BOFF, s + BOFF + PLANE, rb[e], tid + 256 * e, BKF != 0);
    }
  };

  load(kbeg);
  store(0, kbeg);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load(kbeg + (kt + 1) * X3_BK);
    const char* s = lds[cur];
    if (want_rowsum) {
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        const bf16x8 h = *(const bf16x8*)(s + x3_off(tid, ch));
#pragma unroll
        for (int e = 0; e < 8; ++e) rowsum += bf2f((bf16_t)h[e]);
        if (NP == 3) {
          const bf16x8 l = *(const bf16x8*)(s + PLANE + x3_off(tid, ch));
#pragma unroll
          for (int e = 0; e < 8; ++e) rowsum += bf2f((bf16_t)l[e]);
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int off = x3_off(wm * 32 + i * 16 + fr, ks * 4 + fq);
        ah[i] = *(const bf16x8*)(s + off);
        if (NP == 3) al[i] = *(const bf16x8*)(s + PLANE + off);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int off = x3_off(wn * 32 + j * 16 + fr, ks * 4 + fq);
        bh[j] = *(const bf16x8*)(s + BOFF + off);
        if (NP == 3) bl[j] = *(const bf16x8*)(s + BOFF + PLANE + off);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (NP == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) store(cur ^ 1, kbeg + (kt + 1) * X3_BK);
    __syncthreads();
  }

  if (want_rowsum && m0 + tid < g.M) g.ws_rowsum[(long long)bz * g.M + m0 + tid] = rowsum;
  float* wsz = g.ws ? g.ws + (long long)bz * g.M * g.N : nullptr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 32 + j * 16 + fr;
    if (n >= g.N) continue;
    const float bv = (g.bias && !wsz) ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + fq * 4 + r;
        if (m >= g.M) continue;
        if (wsz) { wsz[(long long)m * g.N + n] = acc[i][j][r]; continue; }
        float v = act_f32(acc[i][j][r] + bv, g.act);
        if (g.residual) v += g.residual[(long long)m * g.ldr + n];
        float* cp = g.c + (long long)m * g.ldc + n;
        *cp = g.accumulate ? *cp + v : v;
      }
  }
}


template <int AKF, int BKF, int NP = 3>
__global__ __launch_bounds__(256) void gemm_x3_kernel(const SerGemmF32Args g) {
  __shared__ __attribute__((aligned(16))) char lds[2][(NP == 3 ? 4 : 2) * X3_PLANE];   // [stage][A_hi, (A_lo,) B_hi(, B_lo)]
  x3_body<AKF, BKF, NP>(g, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// Grouped token-level weight gradients: up to 32 independent dW = dy^T x problems (split over their token
// dimension) in ONE launch.  They feed only the optimizer, so the steppers collect them during backward and issue
// them together at its end, off the dgrad critical path.
constexpr int SER_WGRAD_GROUP = 32;
struct WgradGroupProb {
  const float *dy, *x;
  float *ws, *ws_rowsum;
  int M, N, K, z0, splits;
};
struct WgradGroup {
  WgradGroupProb p[SER_WGRAD_GROUP];
  int nprob;
};
template <int NP>
__global__ __launch_bounds__(256) void gemm_x3_group_kernel(const WgradGroup G) {
  __shared__ __attribute__((aligned(16))) char lds[2][(NP == 3 ? 4 : 2) * X3_PLANE];
  int pi = 0;
  for (int i = 1; i < G.nprob; ++i)
    if ((int)blockIdx.z >= G.p[i].z0) pi = i;
  const WgradGroupProb& P = G.p[pi];
  if ((int)blockIdx.y * 64 >= P.N || (int)blockIdx.x * 64 >= P.K) return;
  SerGemmF32Args g;
  g.a = P.dy; g.b = P.x; g.c = nullptr; g.M = P.N; g.N = P.K; g.K = P.M;
  g.sam = 1; g.sak = P.N; g.sbk = P.K; g.sbn = 1; g.ldc = P.K;
  g.bias = nullptr; g.act = SER_ACT_NONE; g.residual = nullptr; g.ldr = 0; g.accumulate = 0;
  g.k_chunk = 256; g.ws = P.ws; g.ws_rowsum = P.ws_rowsum; g.vec_a = g.vec_b = 1; g.products = NP;
  x3_body<0, 0, NP>(g, blockIdx.x, blockIdx.y, (int)blockIdx.z - P.z0, lds);
}

// Independent token-level products of one dependency level in ONE launch (forward: the six first-level projections of the
// two cross-attention directions, both adapters, ...; backward: their input gradients).  The head used to run its two
// directions on two streams; the join between the two hardware queues costs 50-200 us per fork under load
// (profiles/r03_a_head_step_kernel_sequence.txt), more than the kernels themselves.  Same tile code and the same
// summation order as gemm_x3_kernel: a problem's result does not depend on what it is grouped with.
constexpr int SER_X3_MULTI = 8;
struct X3Multi {
  SerGemmF32Args p[SER_X3_MULTI];
  int start[SER_X3_MULTI + 1];     // first block of problem i in the flattened grid (tiles_n x tiles_m each)
  int nprob;
};
template <int AKF, int BKF, int NP>
__global__ __launch_bounds__(256) void gemm_x3_multi_kernel(const X3Multi G) {
  __shared__ __attribute__((aligned(16))) char lds[2][(NP == 3 ? 4 : 2) * X3_PLANE];
  int pi = 0;
  for (int i = 1; i < G.nprob; ++i)
    if ((int)blockIdx.x >= G.start[i]) pi = i;
  const SerGemmF32Args& g = G.p[pi];
  const int t = (int)blockIdx.x - G.start[pi], tn = (g.N + 63) / 64;
  x3_body<AKF, BKF, NP>(g, t % tn, t / tn, 0, lds);
}

struct ReduceGroupProb {
  const float *ws, *ws_rowsum;
  float *dW, *db;
  int splits, N;
  long long MN, off;
};
struct ReduceGroup {
  ReduceGroupProb p[SER_WGRAD_GROUP];
  int nprob;
};
__global__ void splitk_reduce_group_kernel(const ReduceGroup G, int accumulate) {
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int pi = 0;
  for (int i = 1; i < G.nprob; ++i)
    if (gi >= G.p[i].off) pi = i;
  const ReduceGroupProb& P = G.p[pi];
  const long long i = gi - P.off;
  if (i < P.MN) {
    float v = 0.f;
    for (int z = 0; z < P.splits; ++z) v += P.ws[(long long)z * P.MN + i];
    P.dW[i] = accumulate ? P.dW[i] + v : v;
  }
  if (P.db && i < P.N) {
    float v = 0.f;
    for (int z = 0; z < P.splits; ++z) v += P.ws_rowsum[(long long)z * P.N + i];
    P.db[i] = accumulate ? P.db[i] + v : v;
  }
}

}  // namespace

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static int g_f32_bk = 0;   // 0 = automatic
static int g_f32_bk_fwd() { return g_f32_bk; }
extern "C" int ser_debug_set_f32_bk(int bk) { g_f32_bk = bk; return 0; }
template <int FBK>
static void launch_6464_bk(const SerGemmF32Args& g, dim3 grid, hipStream_t st) {
  dim3 block(256);
  const int kext = g.k_chunk ? g.k_chunk : g.K;
  const bool notail = (g.K % FBK) == 0 && (kext % FBK) == 0 && g.M >= 4 && g.N >= 4;
  const bool ak = g.sak == 1, bk = g.sbk == 1, v = g.vec_a && g.vec_b && notail;
  if (v && ak && bk) hipLaunchKernelGGL((gemm_f32_kernel<64, 64, FBK, 1, 1, 1>), grid, block, 0, st, g);       // y = x W^T
  else if (v && ak && !bk) hipLaunchKernelGGL((gemm_f32_kernel<64, 64, FBK, 1, 0, 1>), grid, block, 0, st, g); // dx = dy W
  else if (v && !ak && !bk) hipLaunchKernelGGL((gemm_f32_kernel<64, 64, FBK, 0, 0, 1>), grid, block, 0, st, g); // dW = dy^T x
  else hipLaunchKernelGGL((gemm_f32_kernel<64, 64, FBK, 2, 2, 2>), grid, block, 0, st, g);
}
static int g_f32_bk_fwd();
static int g_use_x3 = 1;   // token-level head products on split-bf16 MFMA (0: exact fp32 MFMA)
// MFMA products per multiply in the BACKWARD token-level head GEMMs (dgrad, wgrad): 3 = fp32-equivalent (default, parity
// mode), 1 = bf16 operands with fp32 accumulation (the `bf16` precision mode of the system: what mixed-precision
// training does for every matmul).  Forward products always use 3: the 1e-3 logit budget is a forward bound.
static int g_head_bwd_products = 3;
extern "C" int ser_set_head_backward_products(int n) {
  SER_REQUIRE(n == 1 || n == 3, "head backward products must be 1 or 3");
  g_head_bwd_products = n;
  return SER_OK;
}
extern "C" int ser_get_head_backward_products(void) { return g_head_bwd_products; }
// MFMA products per multiply of ser_linear_fwd / ser_linear_fwd_group (token-level forward products).  3 = default (fp32-
// equivalent; the 1e-3 logit budget is a forward bound).  1 = bf16 operands, fp32 accumulation: what the reference's
// --use_amp (bf16 autocast, ref src/train.py:151) computes; the fine-tuning form of the encoders selects it around ITS
// Linear layers in the `bf16` precision mode (models/_finetune.py), the trainable head keeps 3.
static int g_linear_fwd_products = 3;
extern "C" int ser_set_linear_forward_products(int n) {
  SER_REQUIRE(n == 1 || n == 3, "linear forward products must be 1 or 3");
  g_linear_fwd_products = n;
  return SER_OK;
}
extern "C" int ser_get_linear_forward_products(void) { return g_linear_fwd_products; }
extern "C" int ser_debug_set_head_x3(int v) { g_use_x3 = v; return 0; }

static void launch_6464(const SerGemmF32Args& g, dim3 grid, hipStream_t st) {
  {
    const int kext = g.k_chunk ? g.k_chunk : g.K;
    const bool ak = g.sak == 1, bk = g.sbk == 1;
    // a ragged reduction length is supported when both operands are row-contiguous (the wgrad form)
    const bool ktail_ok = (!ak && !bk) || ((g.K % X3_BK) == 0 && (kext % X3_BK) == 0);
    const bool ok = g_use_x3 && g.vec_a && g.vec_b && ktail_ok && (!g.k_chunk || g.k_chunk % X3_BK == 0) && g.M >= 4 && g.N >= 4;
    if (ok && (ak || g.sam == 1) && (bk || g.sbn == 1)) {
      dim3 block(256);
      if (g.products == 1) {
        if (ak && bk) hipLaunchKernelGGL((gemm_x3_kernel<1, 1, 1>), grid, block, 0, st, g);
        else if (ak && !bk) hipLaunchKernelGGL((gemm_x3_kernel<1, 0, 1>), grid, block, 0, st, g);
        else if (!ak && bk) hipLaunchKernelGGL((gemm_x3_kernel<0, 1, 1>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((gemm_x3_kernel<0, 0, 1>), grid, block, 0, st, g);
      } else {
        if (ak && bk) hipLaunchKernelGGL((gemm_x3_kernel<1, 1>), grid, block, 0, st, g);
        else if (ak && !bk) hipLaunchKernelGGL((gemm_x3_kernel<1, 0>), grid, block, 0, st, g);
        else if (!ak && bk) hipLaunchKernelGGL((gemm_x3_kernel<0, 1>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((gemm_x3_kernel<0, 0>), grid, block, 0, st, g);
      }
      return;
    }
  }
  // measured on MI355X (scripts/gemm_f32_bench.py): split-K grids (many resident workgroups) are fastest at
  // BK 16, single-pass grids (<= a few hundred workgroups, latency hidden inside the workgroup) at BK 32
  const int bk = g_f32_bk_fwd() ? g_f32_bk_fwd() : (g.k_chunk ? 16 : 32);
  if (bk == 16) launch_6464_bk<16>(g, grid, st);
  else if (bk == 32) launch_6464_bk<32>(g, grid, st);
  else launch_6464_bk<64>(g, grid, st);
}

// float4 groups along an operand's contiguous axis are usable when the base and the other stride keep them
// 16-byte aligned and the extent along that axis is a multiple of 4 (a group is then all-in or all-out)
static bool vec_ok(const float* p, long long s_fast, long long s_slow, int extent) {
  return s_fast == 1 && (s_slow % 4) == 0 && (extent % 4) == 0 && (((uintptr_t)p) & 15) == 0;
}

// optional per-launch HIP-event timing (bench.py roofline leg of the fine-tuning configuration, where this kernel
// family carries the encoders' products); off by default
struct ProfRecF32 { hipEvent_t e0, e1; double flops; };
static bool g_prof_f32_on = false;
static std::vector<ProfRecF32> g_prof_f32;
extern "C" int ser_prof_gemm_f32_start(void) {
  g_prof_f32.clear();
  g_prof_f32_on = true;
  return SER_OK;
}
extern "C" int ser_prof_gemm_f32_stop(double* total_ms, double* total_flops, long long* launches) {
  g_prof_f32_on = false;
  double ms = 0.0, fl = 0.0;
  for (auto& r : g_prof_f32) {
    SER_CHECK_HIP(hipEventSynchronize(r.e1));
    float t = 0.f;
    SER_CHECK_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
    ms += t;
    fl += r.flops;
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = (long long)g_prof_f32.size();
  g_prof_f32.clear();
  return SER_OK;
}
static int launch_gemm_f32_inner(const SerGemmF32Args& gin, hipStream_t st);

int ser_launch_gemm_f32(const SerGemmF32Args& gin, hipStream_t st) {
  if (!g_prof_f32_on) return launch_gemm_f32_inner(gin, st);
  ProfRecF32 rec;
  SER_CHECK_HIP(hipEventCreate(&rec.e0));
  SER_CHECK_HIP(hipEventCreate(&rec.e1));
  rec.flops = 2.0 * gin.M * (double)gin.N * gin.K;
  SER_CHECK_HIP(hipEventRecord(rec.e0, st));
  const int rc = launch_gemm_f32_inner(gin, st);
  SER_CHECK_HIP(hipEventRecord(rec.e1, st));
  g_prof_f32.push_back(rec);
  return rc;
}

static int launch_gemm_f32_inner(const SerGemmF32Args& gin, hipStream_t st) {
  SerGemmF32Args g = gin;
  SER_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "gemm_f32: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
  SER_REQUIRE(g.a && g.b && g.c, "gemm_f32: null operand");
  g.vec_a = (g.sak == 1 ? vec_ok(g.a, g.sak, g.sam, g.K) : vec_ok(g.a, g.sam, g.sak, g.M)) ? 1 : 0;
  g.vec_b = (g.sbk == 1 ? vec_ok(g.b, g.sbk, g.sbn, g.K) : vec_ok(g.b, g.sbn, g.sbk, g.N)) ? 1 : 0;
  if (g.k_chunk) SER_REQUIRE(g.k_chunk % 4 == 0, "gemm_f32: k_chunk must be a multiple of 4");
  dim3 block(256);
  const int zs = g.k_chunk ? ceil_div(g.K, g.k_chunk) : 1;
  if (g.k_chunk) {
    dim3 grid(ceil_div(g.N, 64), ceil_div(g.M, 64), zs);
    launch_6464(g, grid, st);
  } else if (g.M <= 16) {
    dim3 grid(ceil_div(g.N, 64), ceil_div(g.M, 16));
    hipLaunchKernelGGL((gemm_f32_kernel<16, 64, 64>), grid, block, 0, st, g);
  } else if (g.N <= 16) {
    dim3 grid(ceil_div(g.N, 16), ceil_div(g.M, 64));
    hipLaunchKernelGGL((gemm_f32_kernel<64, 16, 64>), grid, block, 0, st, g);
  } else {
    dim3 grid(ceil_div(g.N, 64), ceil_div(g.M, 64));
    launch_6464(g, grid, st);
  }
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// products: bf16 MFMA products per multiply - 3 (fp32-equivalent) or 1 (bf16 operands, the --use_amp arithmetic)
extern "C" int ser_gemm_f32_np(const float* a, long long sam, long long sak, const float* b, long long sbk, long long sbn,
                               int M, int N, int K, const float* bias, int act, const float* residual, int ldr, float* c,
                               int ldc, int accumulate, int products, void* stream) {
  SER_REQUIRE(products == 1 || products == 3, "gemm_f32: products=%d (1 or 3)", products);
  SerGemmF32Args g;
  g.a = a; g.b = b; g.c = c; g.M = M; g.N = N; g.K = K;
  g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = ldc;
  g.bias = bias; g.act = act; g.residual = residual; g.ldr = ldr; g.accumulate = accumulate;
  g.k_chunk = 0; g.ws = nullptr; g.ws_rowsum = nullptr; g.vec_a = g.vec_b = 0; g.products = products;
  return ser_launch_gemm_f32(g, (hipStream_t)stream);
}
extern "C" int ser_gemm_f32(const float* a, long long sam, long long sak, const float* b, long long sbk, long long sbn,
                            int M, int N, int K, const float* bias, int act, const float* residual, int ldr, float* c,
                            int ldc, int accumulate, void* stream) {
  return ser_gemm_f32_np(a, sam, sak, b, sbk, sbn, M, N, K, bias, act, residual, ldr, c, ldc, accumulate, 3, stream);
}
static int gemm_f32_bwd(const float* a, long long sam, long long sak, const float* b, long long sbk, long long sbn, int M, int N,
                        int K, float* c, int ldc, int accumulate, void* stream) {
  SerGemmF32Args g;
  g.a = a; g.b = b; g.c = c; g.M = M; g.N = N; g.K = K;
  g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = ldc;
  g.bias = nullptr; g.act = SER_ACT_NONE; g.residual = nullptr; g.ldr = 0; g.accumulate = accumulate;
  g.k_chunk = 0; g.ws = nullptr; g.ws_rowsum = nullptr; g.vec_a = g.vec_b = 0; g.products = g_head_bwd_products;
  return ser_launch_gemm_f32(g, (hipStream_t)stream);
}



// y[M,N] = act(x[M,K] W[N,K]^T + b) + residual
extern "C" int ser_linear_fwd(const float* x, const float* W, const float* bias, int act, const float* residual, int ldr,
                              float* y, int M, int N, int K, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (M <= 16 && K % 16 == 0 && aligned16(x) && aligned16(W)) {
    SER_REQUIRE(M > 0 && N > 0, "linear_fwd: empty problem");
    if (K >= 256)
      hipLaunchKernelGGL(skinny_fwd_kernel<8>, dim3(ceil_div(N, 16)), dim3(512), 0, st, x, W, bias, act, residual, ldr, y, M, N, K);
    else
      hipLaunchKernelGGL(skinny_fwd_kernel<2>, dim3(ceil_div(N, 16)), dim3(128), 0, st, x, W, bias, act, residual, ldr, y, M, N, K);
    SER_LAUNCH_CHECK();
    return SER_OK;
  }
  SerGemmF32Args g;
  g.a = x; g.b = W; g.c = y; g.M = M; g.N = N; g.K = K;
  g.sam = K; g.sak = 1; g.sbk = 1; g.sbn = K; g.ldc = N;
  g.bias = bias; g.act = act; g.residual = residual; g.ldr = ldr; g.accumulate = 0;
  g.k_chunk = 0; g.ws = nullptr; g.ws_rowsum = nullptr; g.vec_a = g.vec_b = 0; g.products = g_linear_fwd_products;
  return ser_launch_gemm_f32(g, st);
}

// y = act(LN(LN(x; g1,b1); g2,b2) W^T + bias) with x1, u and the LayerNorm statistics written out (M <= 16, K <= 512)
extern "C" int ser_linear_fwd_ln2(const float* x, const float* W, const float* bias, int act, const float* g1,
                                  const float* b1, const float* g2, const float* b2, float eps, float* y1, float* y2,
                                  float* stats, float* y, int M, int N, int K, void* stream) {
  SER_REQUIRE(M > 0 && M <= 16 && K % 16 == 0 && K <= 512 && N > 0, "linear_fwd_ln2: needs M <= 16, K %% 16 == 0, K <= 512");
  SER_REQUIRE(aligned16(x) && aligned16(W) && aligned16(g1) && aligned16(b1) && aligned16(g2) && aligned16(b2) &&
                  aligned16(y1) && aligned16(y2), "linear_fwd_ln2: unaligned operand");
  hipLaunchKernelGGL(skinny_fwd_ln2_kernel<8>, dim3(ceil_div(N, 16)), dim3(512), 0, (hipStream_t)stream, x, W, bias, act, g1, b1,
                     g2, b2, eps, y1, y2, stats, y, M, N, K);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// dx[M,K] (+)= dy[M,N] W[N,K]
extern "C" int ser_act_bwd(const float* dy, const float* y, int act, long long n, float* dx, void* stream);
static int g_dgrad16 = 1;
extern "C" int ser_debug_set_dgrad16(int v) { g_dgrad16 = v; return 0; }

extern "C" int ser_linear_dgrad(const float* dy, const float* W, const float* relu_mask, float* dx, int M, int N, int K,
                                int accumulate, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (M <= 16 && K % 4 == 0 && aligned16(W) && aligned16(dx)) {
    SER_REQUIRE(M > 0 && N > 0 && K > 0, "linear_dgrad: empty problem");
    if (N >= 128 && g_dgrad16)
      hipLaunchKernelGGL(skinny_dgrad16_kernel<8>, dim3(ceil_div(K, 16)), dim3(512), 0, st, dy, W, relu_mask, dx, M, N, K, accumulate);
    else if (N >= 128)
      hipLaunchKernelGGL(skinny_dgrad_kernel<8>, dim3(ceil_div(K, 64)), dim3(512), 0, st, dy, W, relu_mask, dx, M, N, K, accumulate);
    else
      hipLaunchKernelGGL(skinny_dgrad_kernel<2>, dim3(ceil_div(K, 64)), dim3(128), 0, st, dy, W, relu_mask, dx, M, N, K, accumulate);
    SER_LAUNCH_CHECK();
    return SER_OK;
  }
  SER_TRY(gemm_f32_bwd(dy, N, 1, W, K, 1, M, K, N, dx, K, accumulate, stream));
  if (relu_mask) {
    SER_REQUIRE(!accumulate, "linear_dgrad: relu_mask with accumulate needs the skinny path");
    return ser_act_bwd(dx, relu_mask, SER_ACT_RELU, (long long)M * K, dx, stream);
  }
  return SER_OK;
}

// ---- grouped token-level products (one launch per dependency level) --------------------------------------------------
static bool x3_multi_ok(const SerGemmF32Args& g) {
  return g_use_x3 && g.vec_a && g.vec_b && (g.K % X3_BK) == 0 && g.M > 16 && g.N >= 4 && g.sak == 1;
}
template <int BKF>
static int launch_x3_multi(SerGemmF32Args* probs, int n, int products, hipStream_t st) {
  for (int i0 = 0; i0 < n; i0 += SER_X3_MULTI) {
    X3Multi G;
    memset(&G, 0, sizeof(G));
    G.nprob = n - i0 < SER_X3_MULTI ? n - i0 : SER_X3_MULTI;
    int blocks = 0;
    for (int i = 0; i < G.nprob; ++i) {
      G.p[i] = probs[i0 + i];
      G.start[i] = blocks;
      blocks += ceil_div(G.p[i].M, 64) * ceil_div(G.p[i].N, 64);
    }
    G.start[G.nprob] = blocks;
    if (products == 1) hipLaunchKernelGGL((gemm_x3_multi_kernel<1, BKF, 1>), dim3(blocks), dim3(256), 0, st, G);
    else hipLaunchKernelGGL((gemm_x3_multi_kernel<1, BKF, 3>), dim3(blocks), dim3(256), 0, st, G);
    SER_LAUNCH_CHECK();
  }
  return SER_OK;
}

// y_i[M,N] = act_i(x_i[M,K] W_i[N,K]^T + b_i) + residual_i for nprob independent problems.
// ptrs: 5 per problem {x, W, bias | null, residual | null, y}; dims: 5 per problem {M, N, K, act, ldr}.
// Problems the split-bf16 tile kernel cannot take (M <= 16, N < 4, K % 64 != 0, unaligned) are launched on their own.
extern "C" int ser_linear_fwd_group(const void* const* ptrs, const int* dims, int nprob, void* stream) {
  SER_REQUIRE(ptrs && dims && nprob >= 1 && nprob <= 64, "linear_fwd_group: bad arguments");
  SerGemmF32Args grp[64];
  int ng = 0;
  for (int i = 0; i < nprob; ++i) {
    const float* x = (const float*)ptrs[5 * i]; const float* W = (const float*)ptrs[5 * i + 1];
    const float* b = (const float*)ptrs[5 * i + 2]; const float* r = (const float*)ptrs[5 * i + 3]; float* y = (float*)ptrs[5 * i + 4];
    const int M = dims[5 * i], N = dims[5 * i + 1], K = dims[5 * i + 2], act = dims[5 * i + 3], ldr = dims[5 * i + 4];
    SER_REQUIRE(x && W && y && M > 0 && N > 0 && K > 0, "linear_fwd_group: problem %d is empty", i);
    SerGemmF32Args g;
    g.a = x; g.b = W; g.c = y; g.M = M; g.N = N; g.K = K; g.sam = K; g.sak = 1; g.sbk = 1; g.sbn = K; g.ldc = N;
    g.bias = b; g.act = act; g.residual = r; g.ldr = ldr; g.accumulate = 0; g.k_chunk = 0; g.ws = nullptr; g.ws_rowsum = nullptr;
    g.vec_a = vec_ok(x, 1, K, K) ? 1 : 0; g.vec_b = vec_ok(W, 1, K, K) ? 1 : 0; g.products = g_linear_fwd_products;
    if (x3_multi_ok(g)) grp[ng++] = g;
    else SER_TRY(ser_linear_fwd(x, W, b, act, r, ldr, y, M, N, K, stream));
  }
  return ng ? launch_x3_multi<1>(grp, ng, g_linear_fwd_products, (hipStream_t)stream) : SER_OK;
}

// dx_i[M,K] (+)= dy_i[M,N] W_i[N,K].  ptrs: 3 per problem {dy, W, dx}; dims: 4 per problem {M, N, K, accumulate}.
extern "C" int ser_linear_dgrad_group(const void* const* ptrs, const int* dims, int nprob, void* stream) {
  SER_REQUIRE(ptrs && dims && nprob >= 1 && nprob <= 64, "linear_dgrad_group: bad arguments");
  SerGemmF32Args grp[64];
  int ng = 0;
  for (int i = 0; i < nprob; ++i) {
    const float* dy = (const float*)ptrs[3 * i]; const float* W = (const float*)ptrs[3 * i + 1]; float* dx = (float*)ptrs[3 * i + 2];
    const int M = dims[4 * i], N = dims[4 * i + 1], K = dims[4 * i + 2], accumulate = dims[4 * i + 3];
    SER_REQUIRE(dy && W && dx && M > 0 && N > 0 && K > 0, "linear_dgrad_group: problem %d is empty", i);
    SerGemmF32Args g;   // C[M,K] = A[M,N] . B[N,K]: A = dy (k = n contiguous), B = W rows (column index contiguous)
    g.a = dy; g.b = W; g.c = dx; g.M = M; g.N = K; g.K = N; g.sam = N; g.sak = 1; g.sbk = K; g.sbn = 1; g.ldc = K;
    g.bias = nullptr; g.act = SER_ACT_NONE; g.residual = nullptr; g.ldr = 0; g.accumulate = accumulate; g.k_chunk = 0;
    g.ws = nullptr; g.ws_rowsum = nullptr;
    g.vec_a = vec_ok(dy, 1, N, N) ? 1 : 0; g.vec_b = vec_ok(W, 1, K, K) ? 1 : 0; g.products = g_head_bwd_products;
    if (x3_multi_ok(g)) grp[ng++] = g;
    else SER_TRY(ser_linear_dgrad(dy, W, nullptr, dx, M, N, K, accumulate, stream));
  }
  return ng ? launch_x3_multi<0>(grp, ng, g_head_bwd_products, (hipStream_t)stream) : SER_OK;
}

// Up to SER_WGRAD_BATCH skinny weight gradients in ONE launch (problem index = blockIdx.z): the 70 weight gradients
// of the classifier's residual stack feed nothing but the optimizer, so they are collected during backward and
// issued together at its end instead of sitting, 35 launches deep, on the dgrad critical path.
constexpr int SER_WGRAD_BATCH = 80;
struct SkinnyWgradBatch {
  SkinnyWgradArgs p[SER_WGRAD_BATCH];
};
__global__ __launch_bounds__(256) void skinny_wgrad_batch_kernel(const SkinnyWgradBatch bt, int M, int accumulate) {
  const SkinnyWgradArgs& P = bt.p[blockIdx.z];
  const float* __restrict__ dy = P.dy;
  const float* __restrict__ x = P.x;
  float* __restrict__ dW = P.dW;
  float* __restrict__ db = P.db;
  const int N = P.N, Kc = P.Kc;
  if ((int)blockIdx.y * 64 >= N || (int)blockIdx.x * 64 >= Kc) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.y * 64 + w * 16;
  const int c0 = blockIdx.x * 64 + 4 * i;
  const bool cok = c0 < Kc, nok = n0 + i < N;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float asum = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int m = 4 * s + q;
    const bool mok = m < M;
    const float a = (mok && nok) ? dy[(long long)m * N + n0 + i] : 0.f;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mok && cok) b = *(const float4*)(x + (long long)m * Kc + c0);
    asum += a;
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.y, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.z, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.w, acc[3], 0, 0, 0);
  }
  if (cok) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + q * 4 + r;
      if (n >= N) continue;
      float4* dst = (float4*)(dW + (long long)n * Kc + c0);
      float4 o = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
      if (accumulate) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
      *dst = o;
    }
  }
  if (db && blockIdx.x == 0) {
    asum += __shfl_xor(asum, 16, 64);
    asum += __shfl_xor(asum, 32, 64);
    if (q == 0 && nok) db[n0 + i] = accumulate ? db[n0 + i] + asum : asum;
  }
}

extern "C" size_t ser_linear_wgrad_group_workspace_bytes(const int* dims, int nprob) {
  size_t total = 0;
  for (int i = 0; i < nprob; ++i) {
    const int M = dims[3 * i], N = dims[3 * i + 1], K = dims[3 * i + 2];
    const size_t splits = ceil_div(M, 256);
    total += ((splits * N * K + splits * N) * sizeof(float) + 255) & ~(size_t)255;
  }
  return total + 256;
}

// ptrs[4*i..] = {dy[M,N], x[M,K], dW[N,K], db[N] or NULL}; dims[3*i..] = {M, N, K}; M > 16 rows (tokens)
extern "C" int ser_linear_wgrad_group(const void* const* ptrs, const int* dims, int nprob, int accumulate, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  SER_REQUIRE(nprob > 0 && nprob <= SER_WGRAD_GROUP, "linear_wgrad_group: nprob=%d (max %d)", nprob, SER_WGRAD_GROUP);
  SER_REQUIRE(workspace && workspace_bytes >= ser_linear_wgrad_group_workspace_bytes(dims, nprob), "linear_wgrad_group: workspace too small");
  WgradGroup G;
  ReduceGroup R;
  memset(&G, 0, sizeof(G));
  memset(&R, 0, sizeof(R));
  G.nprob = R.nprob = nprob;
  char* wp = (char*)workspace;
  int z = 0, gx = 1, gy = 1;
  long long off = 0;
  for (int i = 0; i < nprob; ++i) {
    const int M = dims[3 * i], N = dims[3 * i + 1], K = dims[3 * i + 2];
    const float* dy = (const float*)ptrs[4 * i];
    const float* x = (const float*)ptrs[4 * i + 1];
    SER_REQUIRE(M > 16 && N >= 4 && K >= 4 && N % 4 == 0 && K % 4 == 0 && aligned16(dy) && aligned16(x),
                "linear_wgrad_group: problem %d (M=%d N=%d K=%d) unsupported", i, M, N, K);
    const int splits = ceil_div(M, 256);
    WgradGroupProb& p = G.p[i];
    p.dy = dy; p.x = x; p.M = M; p.N = N; p.K = K; p.z0 = z; p.splits = splits;
    p.ws = (float*)wp;
    p.ws_rowsum = ptrs[4 * i + 3] ? p.ws + (size_t)splits * N * K : nullptr;
    wp += (((size_t)splits * N * K + (size_t)splits * N) * sizeof(float) + 255) & ~(size_t)255;
    ReduceGroupProb& r = R.p[i];
    r.ws = p.ws; r.ws_rowsum = p.ws_rowsum; r.dW = (float*)ptrs[4 * i + 2]; r.db = (float*)ptrs[4 * i + 3];
    r.splits = splits; r.N = N; r.MN = (long long)N * K; r.off = off;
    off += (r.MN + 255) / 256 * 256;
    z += splits;
    gx = gx > ceil_div(K, 64) ? gx : ceil_div(K, 64);
    gy = gy > ceil_div(N, 64) ? gy : ceil_div(N, 64);
  }
  if (g_head_bwd_products == 1) hipLaunchKernelGGL(gemm_x3_group_kernel<1>, dim3(gx, gy, z), dim3(256), 0, st, G);
  else hipLaunchKernelGGL(gemm_x3_group_kernel<3>, dim3(gx, gy, z), dim3(256), 0, st, G);
  hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3((unsigned)(off / 256)), dim3(256), 0, st, R, accumulate);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// ptrs[4*i .. 4*i+3] = {dy, x, dW, db} of problem i (host array of device pointers), dims[2*i..] = {N, K}
extern "C" int ser_linear_wgrad_batch(const void* const* ptrs, const int* dims, int nprob, int M, int accumulate,
                                      void* stream) {
  SER_REQUIRE(nprob > 0 && nprob <= SER_WGRAD_BATCH && M > 0 && M <= 16, "linear_wgrad_batch: nprob=%d (max %d), M=%d (max 16)",
              nprob, SER_WGRAD_BATCH, M);
  SkinnyWgradBatch bt;
  int gx = 1, gy = 1;
  for (int i = 0; i < nprob; ++i) {
    SkinnyWgradArgs& a = bt.p[i];
    a.dy = (const float*)ptrs[4 * i]; a.x = (const float*)ptrs[4 * i + 1];
    a.dW = (float*)ptrs[4 * i + 2]; a.db = (float*)ptrs[4 * i + 3];
    a.N = dims[2 * i]; a.Kc = dims[2 * i + 1];
    SER_REQUIRE(a.Kc % 4 == 0 && aligned16(a.x) && aligned16(a.dW), "linear_wgrad_batch: problem %d unaligned", i);
    gx = gx > ceil_div(a.Kc, 64) ? gx : ceil_div(a.Kc, 64);
    gy = gy > ceil_div(a.N, 64) ? gy : ceil_div(a.N, 64);
  }
  for (int i = nprob; i < SER_WGRAD_BATCH; ++i) bt.p[i] = bt.p[0];
  hipLaunchKernelGGL(skinny_wgrad_batch_kernel, dim3(gx, gy, nprob), dim3(256), 0, (hipStream_t)stream, bt, M, accumulate);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// two skinny weight gradients (M <= 16) in one launch: dWa (+)= dya^T xa, dWb (+)= dyb^T xb, with their bias gradients
extern "C" int ser_linear_wgrad_pair(const float* dya, const float* xa, float* dWa, float* dba, int Na, int Ka,
                                     const float* dyb, const float* xb, float* dWb, float* dbb, int Nb, int Kb, int M,
                                     int accumulate, void* stream) {
  SER_REQUIRE(M > 0 && M <= 16 && Ka % 4 == 0 && Kb % 4 == 0, "linear_wgrad_pair: needs M <= 16 and K %% 4 == 0");
  SER_REQUIRE(aligned16(xa) && aligned16(xb) && aligned16(dWa) && aligned16(dWb), "linear_wgrad_pair: unaligned operand");
  SkinnyWgradArgs a{dya, xa, dWa, dba, Na, Ka}, b{dyb, xb, dWb, dbb, Nb, Kb};
  const int gx = ceil_div(Ka > Kb ? Ka : Kb, 64), gy = ceil_div(Na > Nb ? Na : Nb, 64);
  hipLaunchKernelGGL(skinny_wgrad_kernel, dim3(gx, gy, 2), dim3(256), 0, (hipStream_t)stream, a, b, M, accumulate);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" size_t ser_linear_wgrad_workspace_bytes(int M, int N, int K) {
  if (M <= 16) return 0;
  const int splits = ceil_div(M, 256);
  return ((size_t)splits * N * K + (size_t)splits * N) * sizeof(float) + 256;
}

// dW[N,K] (+)= dy[M,N]^T x[M,K] ; db[N] (+)= colsum(dy)   (db may be NULL)
extern "C" int ser_linear_wgrad(const float* dy, const float* x, float* dW, float* db, int M, int N, int K,
                                int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  SER_REQUIRE(M > 0 && N > 0 && K > 0, "linear_wgrad: empty problem");
  if (M <= 16 && K % 4 == 0 && aligned16(x) && aligned16(dW)) {
    SkinnyWgradArgs a{dy, x, dW, db, N, K};
    hipLaunchKernelGGL(skinny_wgrad_kernel, dim3(ceil_div(K, 64), ceil_div(N, 64), 1), dim3(256), 0, st, a, a, M, accumulate);
    SER_LAUNCH_CHECK();
    return SER_OK;
  }
  // many rows (tokens): split the reduction over workgroups, partial tiles + fused bias row sums, then reduce
  SER_REQUIRE(workspace && workspace_bytes >= ser_linear_wgrad_workspace_bytes(M, N, K), "linear_wgrad: workspace too small");
  const int chunk = 256, splits = ceil_div(M, chunk);
  SerGemmF32Args g;
  g.a = dy; g.b = x; g.c = dW; g.M = N; g.N = K; g.K = M;
  g.sam = 1; g.sak = N; g.sbk = K; g.sbn = 1; g.ldc = K;
  g.bias = nullptr; g.act = SER_ACT_NONE; g.residual = nullptr; g.ldr = 0; g.accumulate = accumulate;
  g.k_chunk = chunk; g.vec_a = g.vec_b = 0; g.products = g_head_bwd_products;
  g.ws = (float*)workspace;
  g.ws_rowsum = db ? g.ws + (size_t)splits * N * K : nullptr;
  SER_TRY(ser_launch_gemm_f32(g, st));
  const long long MN = (long long)N * K;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, st, g.ws, g.ws_rowsum, splits, MN, N,
                     dW, db, accumulate);
  SER_LAUNCH_CHECK();
  return SER_OK;
}
