// Internal helpers shared by the HIP translation units of libser_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/ser_hip.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define SER_DEVFN __device__ __forceinline__

// thread-local error text, returned by ser_last_error_string()
void ser_set_error(const char* fmt, ...);

#define SER_CHECK_HIP(expr)                                                         \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      ser_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return SER_E_HIP;                                                             \
    }                                                                               \
  } while (0)

#define SER_REQUIRE(cond, ...)                       \
  do {                                               \
    if (!(cond)) {                                   \
      ser_set_error(__VA_ARGS__);                    \
      return SER_E_ARG;                              \
    }                                                \
  } while (0)

#define SER_LAUNCH_CHECK()                                              \
  do {                                                                  \
    hipError_t _e = hipGetLastError();                                  \
    if (_e != hipSuccess) {                                             \
      ser_set_error("%s:%d: launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
      return SER_E_HIP;                                                 \
    }                                                                   \
  } while (0)

#define SER_TRY(expr)            \
  do {                           \
    int _rc = (expr);            \
    if (_rc != SER_OK) return _rc; \
  } while (0)

// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
SER_DEVFN bf16_t f2bf(float x) {
  uint32_t u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
SER_DEVFN float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// x = hi + lo (+ O(2^-17 |x|)); both bf16
SER_DEVFN void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

// two fp32 -> packed split planes with the hardware converter (v_cvt_pk_bf16_f32, round-to-nearest-even):
// hi = {bf16(a), bf16(b)}, lo = {bf16(a - hi_a), bf16(b - hi_b)}; low half = first element
typedef __attribute__((ext_vector_type(2))) __bf16 ser_bf16x2;
SER_DEVFN uint32_t pack_bf16x2(float a, float b) {
  ser_bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
SER_DEVFN void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(a, b);
  lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

// ---- interleaved split planes ("il" layout) -----------------------------------------------------------------------
// A split tensor x = hi + lo is stored either as two separate planes (planar) or as ONE array in which every run of
// 32 elements is followed by its 32 lo elements: [hi 0..31 | lo 0..31 | hi 32..63 | lo 32..63 | ...].  A 128-byte line of
// a K-contiguous row then holds everything the three-product MFMA step needs for 32 values of k, so the GEMM stages the
// same bytes per k-tile as the single-product bf16 kernel and issues 3 MFMAs instead of 2 on them.
// Convention (internal and at the C ABI): a plane pair with lo == hi + 32 elements IS the interleaved layout; flat
// element offsets (row * D + col with D % 32 == 0) map by il_off(), hi and lo share the mapped offset.
#define SER_IL_GROUP 32
static __host__ __device__ __forceinline__ bool ser_is_il(const void* hi, const void* lo) {
  return lo != nullptr && (const char*)lo == (const char*)hi + 2 * SER_IL_GROUP;
}
static __host__ __device__ __forceinline__ long long ser_il_off(long long off) { return off + (off & ~(long long)(SER_IL_GROUP - 1)); }

// erf-GELU, x * Phi(x), with Phi(x) = 1/2 erfc(-x / sqrt 2) from the 5-term rational/exponential form
// (Abramowitz & Stegun 7.1.26, |erf error| <= 1.5e-7): 2 transcendental + ~14 plain VALU instructions instead of the
// ~40 of libm's erff, which made the GELU epilogues VALU-bound.  Measured in fp32 over [-12, 12]:
// max |gelu_erf(x) - exact| = 4.6e-7 (1.5e-7 relative at the maximum, i.e. one fp32 ulp).
SER_DEVFN float gelu_erf(float x) {
  const float a = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float q = p * t * __builtin_amdgcn_exp2f(a * a * -1.4426950408889634f);   // erfc(a)
  const float h = fmaf(-0.5f, q, 0.5f);                                            // Phi(|x|) - 1/2
  return x * (0.5f + copysignf(h, x));
}

// Cross-lane reductions on the DPP path (a few cycles per step) instead of ds_bpermute shuffles (~100 cycles per
// step through the LDS crossbar): lanes are combined inside their quad, then across the quads of a row of 16
// (row_half_mirror, row_mirror), and the four row results are read with v_readlane.  Every step combines the same
// two partial results on every lane, so all lanes end with identical bits.  Call with all 64 lanes active.
template <int CTRL>
SER_DEVFN float dpp_move(float v, float old) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                               0xf, 0xf, false));
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

SER_DEVFN float row16_sum(float v) {      // sum over the 16 lanes of a DPP row, result on every lane of the row
  v += dpp_move<DPP_QUAD_XOR1>(v, 0.f);
  v += dpp_move<DPP_QUAD_XOR2>(v, 0.f);
  v += dpp_move<DPP_ROW_HALF_MIRROR>(v, 0.f);
  v += dpp_move<DPP_ROW_MIRROR>(v, 0.f);
  return v;
}
SER_DEVFN float row16_max(float v) {
  v = fmaxf(v, dpp_move<DPP_QUAD_XOR1>(v, v));
  v = fmaxf(v, dpp_move<DPP_QUAD_XOR2>(v, v));
  v = fmaxf(v, dpp_move<DPP_ROW_HALF_MIRROR>(v, v));
  v = fmaxf(v, dpp_move<DPP_ROW_MIRROR>(v, v));
  return v;
}
SER_DEVFN float lane_value(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
SER_DEVFN float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
SER_DEVFN float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// ---- dropout masks: a counter-based generator keyed by (state, site, element).  `state` lives in device memory (so a
// captured graph draws new masks at every replay) and is advanced by the host side once per training step; `site` names
// the dropout layer; backward regenerates the mask from the same triple instead of storing it.
SER_DEVFN unsigned ser_mix32(unsigned h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
struct SerDropout {
  const unsigned long long* state;   // null or p == 0: identity
  unsigned site;
  float p;
};
// multiplier of element idx: 1 / (1 - p) with probability 1 - p, else 0
SER_DEVFN float ser_drop_mult(unsigned long long st, unsigned site, unsigned idx, unsigned thresh, float scale) {
  unsigned h = ser_mix32(idx * 0x9E3779B1u ^ (unsigned)st);
  h = ser_mix32(h + site * 0x85EBCA77u + (unsigned)(st >> 32));
  return h >= thresh ? scale : 0.f;
}
SER_DEVFN unsigned ser_drop_thresh(float p) { return (unsigned)fminf(p * 4294967296.0f, 4294967040.0f); }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// simple bump allocator over a caller-provided workspace
struct SerArena {
  char* base;
  size_t cap, off;
  SerArena(void* p, size_t n) : base((char*)p), cap(n), off(0) {}
  void* take(size_t bytes) {
    size_t a = (off + 255) & ~(size_t)255;
    if (base != nullptr && a + bytes > cap) return nullptr;
    off = a + bytes;
    return base ? base + a : (void*)1;
  }
  template <typename T> T* get(size_t n) { return (T*)take(n * sizeof(T)); }
};

// ---- internal launchers (defined across the .hip files) -------------------------------------
struct SerGemmArgs {
  const bf16_t *a_hi, *a_lo;   // [M, lda] K-contiguous rows (lo may be null -> single bf16 product)
  const bf16_t *w_hi, *w_lo;   // [N, ldw] K-contiguous rows
  int M, N, K, lda, ldw;
  int nb1, nb2;                // two batch dims (grid.z = nb1*nb2)
  long long sa1, sa2, sw1, sw2, sc1, sc2, sr1, sr2, sbias1, sbias2;  // element strides per batch dim
  const float* bias;           // [N] or null
  int act;                     // SER_ACT_*
  const float* residual;       // [M, ldr] added after activation, or null
  int ldr;
  float* c_f32;                // [M, ldc] or null
  bf16_t *c_hi, *c_lo;         // [M, ldc] planes or null
  int ldc;
  // split-K (interleaved three-product mode): ksplit > 1 = each of ksplit slices of K writes its raw partial sums to
  // c_f32 + slice * slab_stride (bias, activation, residual and planes are left to the consumer, which adds the slabs)
  int ksplit;
  long long slab_stride;
  int cfg;                     // tile configuration id (SER_GEMM_CFG_*), 0 = let the launcher choose
  int group_m;                 // tile order: m-tiles per super-tile (0 = launcher default); pure speed, any value gives the same result
};
// tile configurations: 64..192 = rows of a 256-thread BM x 128 tile; the 512-thread tiles:
enum { SER_GEMM_CFG_NARROW = 7000 /* + rows: BM x 64 tiles */, SER_GEMM_CFG_SINGLE = 3000 /* + rows: single LDS buffer, three workgroups per CU */, SER_GEMM_CFG_WIDE = 1000, SER_GEMM_CFG_128x256 = 1128, SER_GEMM_CFG_192x256 = 1192, SER_GEMM_CFG_256x256 = 1256, SER_GEMM_CFG_256x128 = 2256,
       SER_GEMM_CFG_128x256_3 = 5128, SER_GEMM_CFG_256x128_3 = 6256, SER_GEMM_CFG_256x256_4W = 8256 };
int ser_launch_gemm_bf16(const SerGemmArgs& g, hipStream_t st);
extern "C" int ser_gemm_plan_get(long long rows_total, int N, int K, int three_products, int* cfg, int* ksplit);
int ser_launch_gemm_bf16_pair(const SerGemmArgs& small, const SerGemmArgs& big, hipStream_t st);
int ser_launch_split(const float* x, bf16_t* hi, bf16_t* lo, long long n, hipStream_t st);
int ser_launch_layernorm(const float* x, const float* x2, const float* gamma, const float* beta, float eps, int rows,
                         int D, float* y, bf16_t* yhi, bf16_t* ylo, hipStream_t st);
struct SerLnArgs {
  const float *x, *x2, *gamma, *beta;
  float eps;
  int rows, D;
  float* y;
  bf16_t *yhi, *ylo;
  // input = sum of `nsl` slabs x + s * sstride (split-K partial sums of the producing GEMM) + bias[D] + x2 (residual)
  int nsl;
  long long sstride;
  const float* bias;
};
int ser_launch_layernorm_pair(const SerLnArgs& a, const SerLnArgs& b, hipStream_t st);
int ser_launch_layernorm_ex(const SerLnArgs& a, hipStream_t st);
struct SerAttnArgs {
  const bf16_t *qkv_hi, *qkv_lo;
  const float* key_mask;
  int B, S, heads;
  bf16_t *ctx_hi, *ctx_lo;
};
int ser_launch_self_attention_pair(const SerAttnArgs& a, const SerAttnArgs& b, hipStream_t st);
size_t ser_conv0_scratch_bytes(int B, int L0, int C0);
int ser_launch_conv0(const float* wave, int B, int T, const float* w, const float* gn_g, const float* gn_b, int C0,
                     int KW, int ST, int L0, bf16_t* yhi, bf16_t* ylo, void* scratch, hipStream_t st);
int ser_launch_posconv_slab(const float* z, int B, int S, int H, int G, int K, bf16_t* hi, bf16_t* lo, hipStream_t st);
// the whole positional conv (+ bias, GELU, residual) with the (clip, group) slab resident in LDS (posconv.hip); interleaved weights
int ser_posconv_direct_ok(int S, int H, int G, int K);
int ser_launch_posconv_direct(const float* z, const bf16_t* w_il, const float* bias, float* out, int B, int S, int H, int G, int K, hipStream_t st);
int ser_launch_xlmr_embed(const int64_t* ids, int B, int S, const float* wemb, const float* pemb, const float* temb,
                          const float* gamma, const float* beta, float eps, int D, int vocab, int max_pos, int pad_id,
                          int* pos_scratch, float* y, bf16_t* yhi, bf16_t* ylo, hipStream_t st);
int ser_launch_self_attention(const bf16_t* qkv_hi, const bf16_t* qkv_lo, const float* key_mask, int B, int S,
                              int heads, bf16_t* ctx_hi, bf16_t* ctx_lo, hipStream_t st);
