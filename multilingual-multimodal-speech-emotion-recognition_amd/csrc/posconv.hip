// Positional convolution embedding of Wav2Vec2 (hf modeling_wav2vec2.py:326-368: grouped Conv1d, kernel 128, 16 groups,
// "same" padding with the last output dropped, + bias, GELU, + residual) for the interleaved three-product mode, as ONE
// kernel with the (clip, group) slab resident in LDS.
//
// Seen as a GEMM (what encoders.hip did before, and still does for long clips): per (clip, group) M = S frames,
// N = C_g channels, K = 128 taps x C_g, with the Toeplitz operand A[t, j C_g + c] = x[t + j - 64, c].  Through the tiled
// GEMM every k-step re-stages a BM x 32 window of A although consecutive taps are the SAME rows shifted by one: 3.2 GB of
// global->LDS traffic for 0.6 MB of distinct input, 210 us, the largest single launch of the encoder pass after the
// convolution GEMMs.  Here the S + 127 rows of the group's input are split into bf16 hi / lo planes once, straight from the
// fp32 feature projection output, and stay in LDS (83 KB at S = 199); only the weights stream (12 KB per tap, double
// buffered), and a tap's A fragments are LDS reads at a row offset.  One workgroup per (clip, group): 16 x 16 = 256
// workgroups at the benchmark size, one per CU.
//
// Arithmetic: the same products in the same order per accumulator as the GEMM path (k ascending = tap-major, then the two
// 32-channel halves; a_lo w_hi, a_hi w_lo, a_hi w_hi per k-step), the same hi / lo split of the input, the same epilogue
// (erf-GELU of acc + bias, then + residual).  Measured against that path: 80 % of the outputs identical, the rest one ulp
// apart (the cause was not isolated: operands, products and their order agree by construction, and every other order of
// the three products is further away); against a float64 evaluation of the same split operands both are equally close
// (2.13e-5 on N(0,1) inputs).  tests/test_gpu_encoders.py holds both comparisons.
#include "ser_common.h"

namespace {

constexpr int PC_K = 128;          // taps
constexpr int PC_CP = 64;          // channels per group as stored (C_g <= 64, zero padded)
constexpr int PC_RB = 256;         // bytes per LDS row: [hi 0..31 | lo 0..31 | hi 32..63 | lo 32..63]
constexpr int PC_THREADS = 512;       // 8 waves: two per SIMD, one's LDS reads and barrier waits under the other's MFMAs
constexpr int PC_WAVES = PC_THREADS / 64;

SER_DEVFN int pc_off(int row, int chunk) { return row * PC_RB + ((chunk ^ ((row & 7) << 1)) << 4); }

// MAXRB: 16-frame blocks a workgroup can hold (S <= 16 MAXRB); NCB: 16-channel blocks of the group (C_g = 16 NCB)
// EPI 0 (forward): v = acc + bias; out = GELU(v) + z; `raw` (optional) keeps v for backward.
// EPI 1 (a correlation with prepared weights, no activation): out = acc + add - the input gradient of the same conv is this kernel
// over the pre-activation gradient with the taps reversed and the channel roles swapped (ser_posconv_pack, flip) and the window
// moved by one frame (shift = 1: the even kernel's padding is K/2 in front and K/2 - 1 behind), `add` = the residual branch's
// gradient.
template <int MAXRB, int NCB, int EPI>
__global__ __launch_bounds__(PC_THREADS) void posconv_direct_kernel(const float* __restrict__ z, const bf16_t* __restrict__ w,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     int S, int H, int G, float* __restrict__ raw,
                                                                     const float* __restrict__ add, int shift) {
  constexpr int RBW = (MAXRB + PC_WAVES - 1) / PC_WAVES;     // row blocks per wave
  constexpr int SLAB_ROWS = MAXRB * 16 + PC_K - 1;
  constexpr int WROWS = NCB * 16;
  __shared__ __attribute__((aligned(1024))) char lds[SLAB_ROWS * PC_RB + 2 * WROWS * PC_RB];
  char* slab = lds;
  char* wbuf = lds + SLAB_ROWS * PC_RB;
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int Cg = H / G;
  const int nrb = (S + 15) >> 4;
  const int rows = nrb * 16 + PC_K - 1;                      // slab rows any fragment can touch (zeros beyond S + K/2)
  const float* zb = z + (long long)b * S * H + g * Cg;

  // weights of tap j: rows n of this group, 256 contiguous bytes each (interleaved planes of 64 channels)
  const bf16_t* wg = w + (long long)g * Cg * (2 * PC_K * PC_CP);
  constexpr int WCH = WROWS * 16, WIT = (WCH + PC_THREADS - 1) / PC_THREADS;      // 16-byte chunks of a tap, per-thread share
  uint4 wreg[WIT];
  auto wload = [&](int j) {
#pragma unroll
    for (int e = 0; e < WIT; ++e) {
      const int idx = tid + e * PC_THREADS, n = idx >> 4, c = idx & 15;
      wreg[e] = (idx < WCH && n < Cg) ? *(const uint4*)(wg + (long long)n * (2 * PC_K * PC_CP) + j * (2 * PC_CP) + c * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int e = 0; e < WIT; ++e) {
      const int idx = tid + e * PC_THREADS, n = idx >> 4, c = idx & 15;
      if (idx < WCH) *(uint4*)(wbuf + buf * WROWS * PC_RB + pc_off(n, c)) = wreg[e];
    }
  };
  wload(0);

  // ---- the group's input, split into planes: slab row r = frame r - K/2, zero outside [0, S) and beyond C_g
  for (int item = tid; item < rows * 8; item += PC_THREADS) {
    const int r = item >> 3, h = (item >> 2) & 1, q = item & 3, t = r - PC_K / 2 + shift, c0 = h * 32 + q * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (t >= 0 && t < S && c0 < Cg) {
      const float4 a = *(const float4*)(zb + (long long)t * H + c0), c = *(const float4*)(zb + (long long)t * H + c0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    }
    uint32_t ph[4], pl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bf16_t h0, l0, h1, l1;
      split_bf16(v[2 * e], h0, l0);
      split_bf16(v[2 * e + 1], h1, l1);
      ph[e] = (uint32_t)h0 | ((uint32_t)h1 << 16);
      pl[e] = (uint32_t)l0 | ((uint32_t)l1 << 16);
    }
    *(uint4*)(slab + pc_off(r, h * 8 + q)) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    *(uint4*)(slab + pc_off(r, h * 8 + 4 + q)) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
  }
  wstore(0);
  wload(1);
  __syncthreads();

  f32x4 acc[RBW][NCB];
#pragma unroll
  for (int i = 0; i < RBW; ++i)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[i][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int j = 0; j < PC_K; ++j) {
    if (j + 1 < PC_K) wstore((j + 1) & 1);          // every wave passed the barrier of tap j - 1: that buffer is free
    if (j + 2 < PC_K) wload(j + 2);
    const char* wb = wbuf + (j & 1) * WROWS * PC_RB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 bh[NCB], bl[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        bh[cb] = *(const bf16x8*)(wb + pc_off(cb * 16 + fr, h * 8 + fq));
        bl[cb] = *(const bf16x8*)(wb + pc_off(cb * 16 + fr, h * 8 + 4 + fq));
      }
      bf16x8 ah[RBW], al[RBW];                       // every row block's fragments are requested before the first MFMA
#pragma unroll
      for (int i = 0; i < RBW; ++i) {
        const int rb = wave + PC_WAVES * i;
        const int row = (rb < nrb ? rb : 0) * 16 + fr + j;
        ah[i] = *(const bf16x8*)(slab + pc_off(row, h * 8 + fq));
        al[i] = *(const bf16x8*)(slab + pc_off(row, h * 8 + 4 + fq));
      }
#pragma unroll
      for (int i = 0; i < RBW; ++i) {
        if (wave + PC_WAVES * i < nrb) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[cb], acc[i][cb], 0, 0, 0);
            acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[cb], acc[i][cb], 0, 0, 0);
            acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[cb], acc[i][cb], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- GELU(acc + bias) + residual -> out[b, t, g C_g + n]     (acc[i][cb][r]: frame rb*16 + fq*4 + r, channel cb*16 + fr)
  float* ob = out + (long long)b * S * H + g * Cg;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int n = cb * 16 + fr;
    if (n >= Cg) continue;
    const float bv = EPI == 0 ? bias[g * Cg + n] : 0.f;
#pragma unroll
    for (int i = 0; i < RBW; ++i) {
      const int rb = wave + PC_WAVES * i;
      if (rb >= nrb) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = rb * 16 + fq * 4 + r;
        if (t >= S) continue;
        const long long o = (long long)t * H + n;
        if (EPI == 0) {
          const float v = __fadd_rn(acc[i][cb][r], bv);
          if (raw) raw[(long long)b * S * H + g * Cg + o] = v;
          ob[o] = __fadd_rn(gelu_erf(v), zb[o]);
        } else {
          ob[o] = __fadd_rn(acc[i][cb][r], add[(long long)b * S * H + g * Cg + o]);
        }
      }
    }
  }
}

}  // namespace

// 1 when the resident-slab kernel covers this geometry (interleaved weights assumed by the caller)
int ser_posconv_direct_ok(int S, int H, int G, int K) {
  const int Cg = G > 0 ? H / G : 0;
  return K == PC_K && G > 0 && H % G == 0 && (Cg == 48 || Cg == 64) && S >= 1 && S <= 22 * 16 && (Cg == 48 || S <= 14 * 16);
}

template <int EPI>
static int launch_direct(const float* z, const bf16_t* w_il, const float* bias, float* out, int B, int S, int H, int G, float* raw,
                         const float* add, int shift, hipStream_t st) {
  const int Cg = H / G;
  const dim3 grid(B * G), block(PC_THREADS);
  if (Cg == 48) {
    if (S <= 14 * 16) hipLaunchKernelGGL((posconv_direct_kernel<14, 3, EPI>), grid, block, 0, st, z, w_il, bias, out, S, H, G, raw, add, shift);
    else hipLaunchKernelGGL((posconv_direct_kernel<22, 3, EPI>), grid, block, 0, st, z, w_il, bias, out, S, H, G, raw, add, shift);
  } else {
    hipLaunchKernelGGL((posconv_direct_kernel<14, 4, EPI>), grid, block, 0, st, z, w_il, bias, out, S, H, G, raw, add, shift);
  }
  SER_LAUNCH_CHECK();
  return SER_OK;
}

int ser_launch_posconv_direct(const float* z, const bf16_t* w_il, const float* bias, float* out, int B, int S, int H, int G, int K,
                              hipStream_t st) {
  SER_REQUIRE(ser_posconv_direct_ok(S, H, G, K), "posconv (resident slab): unsupported geometry S=%d H=%d G=%d K=%d", S, H, G, K);
  return launch_direct<0>(z, w_il, bias, out, B, S, H, G, nullptr, nullptr, 0, st);
}

namespace {
// Wp [H][Cg][K] fp32 (hf layout of the weight-normed conv weight: out channel, in channel of the group, tap) -> the kernel's
// operand [G][Cg rows][K taps][128]: per (row, tap) the interleaved planes of 64 (zero-padded) channels
// [hi 0..31 | lo 0..31 | hi 32..63 | lo 32..63].  flip = 0: row = out channel, channel = in channel, tap as is (forward);
// flip = 1: row = IN channel, channel = OUT channel, taps reversed (the input-gradient correlation).
__global__ void posconv_pack_kernel(const float* __restrict__ wp, int H, int Cg, int flip, bf16_t* __restrict__ out) {
  const long long n_items = (long long)H * PC_K * PC_CP;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % PC_CP);
    const int j = (int)((i / PC_CP) % PC_K);
    const int rowg = (int)(i / ((long long)PC_CP * PC_K));      // g * Cg + row
    const int g = rowg / Cg, row = rowg % Cg;
    float v = 0.f;
    if (ch < Cg) v = flip ? wp[((long long)(g * Cg + ch) * Cg + row) * PC_K + (PC_K - 1 - j)] : wp[((long long)rowg * Cg + ch) * PC_K + j];
    bf16_t h, l;
    split_bf16(v, h, l);
    bf16_t* o = out + ((long long)rowg * PC_K + j) * (2 * PC_CP) + (ch >> 5) * 64 + (ch & 31);
    o[0] = h;
    o[32] = l;
  }
}
}  // namespace

extern "C" int ser_posconv_direct_supported(int S, int H, int G, int K) { return ser_posconv_direct_ok(S, H, G, K); }

/* packed: H * K * 128 bf16 values */
extern "C" int ser_posconv_pack(const float* wp, int H, int G, int K, int flip, uint16_t* packed, void* stream) {
  SER_REQUIRE(wp && packed && K == PC_K && G > 0 && H % G == 0 && H / G <= PC_CP, "posconv_pack: unsupported geometry H=%d G=%d K=%d", H, G, K);
  const long long n = (long long)H * PC_K * PC_CP;
  hipLaunchKernelGGL(posconv_pack_kernel, dim3((unsigned)((n + 255) / 256 > 65535 * 8 ? 65535 * 8 : (n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, wp, H, H / G, flip, (bf16_t*)packed);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

/* out = GELU(conv(z) + bias) + z on [B, S, H] with the packed weights (flip = 0); raw (optional) = conv(z) + bias */
extern "C" int ser_posconv_fwd(const float* z, const uint16_t* packed, const float* bias, int B, int S, int H, int G, int K, float* out,
                               float* raw, void* stream) {
  SER_REQUIRE(z && packed && bias && out && ser_posconv_direct_ok(S, H, G, K), "posconv_fwd: unsupported geometry S=%d H=%d G=%d K=%d", S, H, G, K);
  return launch_direct<0>(z, (const bf16_t*)packed, bias, out, B, S, H, G, raw, nullptr, 0, (hipStream_t)stream);
}

/* dz = conv^T(dpre) + add on [B, S, H] with the packed weights of flip = 1 (the input gradient of the conv itself, plus the
 * gradient `add` of whatever else consumed z) */
extern "C" int ser_posconv_dgrad(const float* dpre, const uint16_t* packed_flip, const float* add, int B, int S, int H, int G, int K,
                                 float* dz, void* stream) {
  SER_REQUIRE(dpre && packed_flip && add && dz && ser_posconv_direct_ok(S, H, G, K), "posconv_dgrad: unsupported geometry S=%d H=%d G=%d K=%d", S, H, G, K);
  return launch_direct<1>(dpre, (const bf16_t*)packed_flip, nullptr, dz, B, S, H, G, nullptr, add, 1, (hipStream_t)stream);
}
