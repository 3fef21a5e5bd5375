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
};

namespace {

constexpr int FBK = 16;

SER_DEVFN float act_f32(float v, int act) {
  switch (act) {
    case SER_ACT_GELU: return gelu_erf(v);
    case SER_ACT_RELU: return fmaxf(v, 0.f);
    case SER_ACT_TANH: return tanhf(v);
    case SER_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

// R = rows of the tile in its non-k dimension (BM for A, BN for B); each thread moves R*16/256 elements
template <int R>
struct TileRegs {
  float v[R * FBK / 256];
};

// element (r, k) of the operand lives at p[r*sr + k*sk]; rows r0.., k from k0
template <int R>
SER_DEVFN void load_tile(TileRegs<R>& t, const float* __restrict__ p, long long sr, long long sk, int r0, int rmax,
                         int k0, int K, int tid, bool kfast) {
  constexpr int NE = R * FBK / 256;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int idx = tid * NE + e;   // consecutive elements of one thread run along the fast axis
    int r, k;
    if (kfast) { r = idx / FBK; k = idx % FBK; } else { k = idx / R; r = idx % R; }
    const int gr = r0 + r, gk = k0 + k;
    t.v[e] = (gr < rmax && gk < K) ? p[(long long)gr * sr + (long long)gk * sk] : 0.f;
  }
}

template <int R, int LD>
SER_DEVFN void store_tile(const TileRegs<R>& t, float* s, int tid, bool kfast) {
  constexpr int NE = R * FBK / 256;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int idx = tid * NE + e;
    int r, k;
    if (kfast) { r = idx / FBK; k = idx % FBK; } else { k = idx / R; r = idx % R; }
    s[k * LD + r] = t.v[e];
  }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const SerGemmF32Args g) {
  constexpr int LDA = BM == 64 ? 80 : 16, LDB = BN == 64 ? 80 : 16;   // (LD % 32) == 16: two k-rows per half-wave, no conflict
  constexpr int WAVES_M = BN == 16 ? 4 : (BM == 64 ? 2 : 1), WAVES_N = 4 / WAVES_M;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
  __shared__ float As[2][FBK * LDA];
  __shared__ float Bs[2][FBK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const bool a_kfast = g.sak == 1, b_kfast = g.sbk == 1;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  TileRegs<BM> ra;
  TileRegs<BN> rb;
  const int nk = (g.K + FBK - 1) / FBK;
  load_tile<BM>(ra, g.a, g.sam, g.sak, m0, g.M, 0, g.K, tid, a_kfast);
  load_tile<BN>(rb, g.b, g.sbn, g.sbk, n0, g.N, 0, g.K, tid, b_kfast);
  store_tile<BM, LDA>(ra, As[0], tid, a_kfast);
  store_tile<BN, LDB>(rb, Bs[0], tid, b_kfast);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_tile<BM>(ra, g.a, g.sam, g.sak, m0, g.M, (kt + 1) * FBK, g.K, tid, a_kfast);
      load_tile<BN>(rb, g.b, g.sbn, g.sbk, n0, g.N, (kt + 1) * FBK, g.K, tid, b_kfast);
    }
#pragma unroll
    for (int ks = 0; ks < FBK / 4; ++ks) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[cur][(ks * 4 + fq) * LDA + wm * WM + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[cur][(ks * 4 + fq) * LDB + wn * WN + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_tile<BM, LDA>(ra, As[cur ^ 1], tid, a_kfast);
      store_tile<BN, LDB>(rb, Bs[cur ^ 1], tid, b_kfast);
    }
    __syncthreads();
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

}  // namespace

int ser_launch_gemm_f32(const SerGemmF32Args& g, hipStream_t st) {
  SER_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "gemm_f32: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
  SER_REQUIRE(g.a && g.b && g.c, "gemm_f32: null operand");
  dim3 block(256);
  if (g.M <= 16) {
    dim3 grid(ceil_div(g.N, 64), ceil_div(g.M, 16));
    hipLaunchKernelGGL((gemm_f32_kernel<16, 64>), grid, block, 0, st, g);
  } else if (g.N <= 16) {
    dim3 grid(ceil_div(g.N, 16), ceil_div(g.M, 64));
    hipLaunchKernelGGL((gemm_f32_kernel<64, 16>), grid, block, 0, st, g);
  } else {
    dim3 grid(ceil_div(g.N, 64), ceil_div(g.M, 64));
    hipLaunchKernelGGL((gemm_f32_kernel<64, 64>), grid, block, 0, st, g);
  }
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_gemm_f32(const float* a, long long sam, long long sak, const float* b, long long sbk, long long sbn,
                            int M, int N, int K, const float* bias, int act, const float* residual, int ldr, float* c,
                            int ldc, int accumulate, void* stream) {
  SerGemmF32Args g;
  g.a = a; g.b = b; g.c = c; g.M = M; g.N = N; g.K = K;
  g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = ldc;
  g.bias = bias; g.act = act; g.residual = residual; g.ldr = ldr; g.accumulate = accumulate;
  return ser_launch_gemm_f32(g, (hipStream_t)stream);
}
