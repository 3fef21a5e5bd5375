// Encoder self-attention, head_dim 64, split-bf16 operands on MFMA (gfx950).
//
// One workgroup = 4 waves = 64 query rows of one (clip, head); keys/values stream through LDS in
// chunks of 64 with an online softmax (fp32 running max / sum per row).  Scores S = Q.K^T and
// O += P.V both run on v_mfma_f32_16x16x32_bf16; with lo planes present each product is the
// three-term split product (parity mode).  K is kept row-major [key][d] and V transposed
// [d][key] in LDS, both in the 8-row x 128-B XOR-swizzled image used by the GEMM so that every
// ds_read_b128 fragment read is bank-conflict free.  P goes through a per-wave LDS tile to turn
// the MFMA C layout (key on the lane) into the A layout (key along k).
#include "ser_common.h"

namespace {

constexpr int KC = 64;   // keys per chunk
constexpr int HD = 64;   // head dim

SER_DEVFN int tile_off(int row, int chunk) {
  return (row >> 3) * 1024 + (row & 7) * 128 + ((chunk ^ (row & 7)) << 4);
}
SER_DEVFN bf16x8 frag(const char* tile, int row, int chunk) { return *(const bf16x8*)(tile + tile_off(row, chunk)); }

struct AttnProb {
  const bf16_t *qkv_hi, *qkv_lo;
  const float* key_mask;
  int S, H;
  bf16_t *ctx_hi, *ctx_lo;
};

template <bool X3>
SER_DEVFN void attn_body(const AttnProb& P, const int bx, const int head, const int b, char* lds) {
  const bf16_t* __restrict__ qkv_hi = P.qkv_hi;
  const bf16_t* __restrict__ qkv_lo = P.qkv_lo;
  const float* __restrict__ key_mask = P.key_mask;
  bf16_t* __restrict__ ctx_hi = P.ctx_hi;
  bf16_t* __restrict__ ctx_lo = P.ctx_lo;
  const int S = P.S, H = P.H;
  constexpr int NPL = X3 ? 2 : 1;
  constexpr int TILE = KC * 128;                       // 64 rows x 128 B
  char* Ks = lds;                                      // [NPL][64 key][64 d]
  char* Vt = lds + NPL * TILE;                         // [NPL][64 d][64 key]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* Ps = lds + NPL * TILE * 2 + wave * NPL * 2048; // [NPL][16 q][64 key]
  const int fr = lane & 15, fq = lane >> 4;
  const int q0 = bx * 64 + wave * 16;
  const long long ld = 3LL * H;
  const bf16_t* plane[2] = {qkv_hi, qkv_lo};

  // Q fragments straight from global: row q0+fr, d = ks*32 + fq*8 .. +8
  bf16x8 qh[2], ql[2];
  {
    int qr = q0 + fr;
    qr = qr < S ? qr : S - 1;
    const long long base = ((long long)b * S + qr) * ld + head * HD;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qh[ks] = *(const bf16x8*)(qkv_hi + base + ks * 32 + fq * 8);
      if (X3) ql[ks] = *(const bf16x8*)(qkv_lo + base + ks * 32 + fq * 8);
    }
  }

  f32x4 o[4];
  float m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[i] = f32x4{0.f, 0.f, 0.f, 0.f}; m[i] = -INFINITY; l[i] = 0.f; }

  const int nchunks = (S + KC - 1) / KC;
  for (int kc = 0; kc < nchunks; ++kc) {
    __syncthreads();
    // ---- stage K (row-major) and V (transposed) for keys kc*64 .. +63
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * 256;          // 0..511 : key = idx>>3, chunk = idx&7
        const int key = idx >> 3, ch = idx & 7;
        int kg = kc * KC + key;
        kg = kg < S ? kg : S - 1;
        const long long rb = ((long long)b * S + kg) * ld + head * HD + ch * 8;
        const bf16x8 kv = *(const bf16x8*)(plane[p] + rb + H);
        *(bf16x8*)(Ks + p * TILE + tile_off(key, ch)) = kv;
        const bf16x8 vv = *(const bf16x8*)(plane[p] + rb + 2 * H);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int d = ch * 8 + e;
          *(short*)(Vt + p * TILE + tile_off(d, key >> 3) + (key & 7) * 2) = vv[e];
        }
      }
    }
    __syncthreads();

    // ---- scores for this wave's 16 rows x 64 keys
    f32x4 sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 kh = frag(Ks, j * 16 + fr, ks * 4 + fq);
        if (X3) {
          const bf16x8 kl = frag(Ks + TILE, j * 16 + fr, ks * 4 + fq);
          sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ql[ks], kh, sc[j], 0, 0, 0);
          sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[ks], kl, sc[j], 0, 0, 0);
        }
        sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[ks], kh, sc[j], 0, 0, 0);
      }
    }
    // scale, mask (C layout: key = j*16 + fr on the lane, row = fq*4 + r in the registers)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kg = kc * KC + j * 16 + fr;
      bool ok = kg < S;
      if (ok && key_mask) ok = key_mask[(long long)b * S + kg] != 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[j][r] = ok ? sc[j][r] * 0.125f : -INFINITY;
    }
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = fmaxf(fmaxf(sc[0][r], sc[1][r]), fmaxf(sc[2][r], sc[3][r]));
      mx = row16_max(mx);
      const float mn = fmaxf(m[r], mx);
      const float mu = mn == -INFINITY ? 0.f : mn;
      alpha[r] = __expf(m[r] - mu);
      float ps = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p = __expf(sc[j][r] - mu);
        sc[j][r] = p;
        ps += p;
      }
      ps = row16_sum(ps);
      l[r] = l[r] * alpha[r] + ps;
      m[r] = mn;
    }
#pragma unroll
    for (int jd = 0; jd < 4; ++jd)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[jd][r] *= alpha[r];

    // ---- P -> per-wave LDS tile (A layout source), then O += P.V
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = fq * 4 + r, key = j * 16 + fr;
        bf16_t h, lo_;
        split_bf16(sc[j][r], h, lo_);
        const int off = tile_off(row, key >> 3) + (key & 7) * 2;
        *(bf16_t*)(Ps + off) = h;
        if (X3) *(bf16_t*)(Ps + 2048 + off) = lo_;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 ph = frag(Ps, fr, ks * 4 + fq);
      bf16x8 pl;
      if (X3) pl = frag(Ps + 2048, fr, ks * 4 + fq);
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {
        const bf16x8 vh = frag(Vt, jd * 16 + fr, ks * 4 + fq);
        if (X3) {
          const bf16x8 vl = frag(Vt + TILE, jd * 16 + fr, ks * 4 + fq);
          o[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vh, o[jd], 0, 0, 0);
          o[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vl, o[jd], 0, 0, 0);
        }
        o[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vh, o[jd], 0, 0, 0);
      }
    }
    asm volatile("" ::: "memory");
  }

  // ---- normalise and store ctx planes [B*S, H]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + fq * 4 + r;
    if (q >= S) continue;
    const float inv = l[r] > 0.f ? 1.0f / l[r] : 0.f;
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) {
      const long long off = ((long long)b * S + q) * H + head * HD + jd * 16 + fr;
      bf16_t h, lo_;
      split_bf16(o[jd][r] * inv, h, lo_);
      ctx_hi[off] = h;
      if (X3) ctx_lo[off] = lo_;
    }
  }
}

template <bool X3>
__global__ __launch_bounds__(256) void self_attention_kernel(const AttnProb P) {
  __shared__ __attribute__((aligned(1024))) char lds[(X3 ? 2 : 1) * KC * 128 * 2 + 4 * (X3 ? 2 : 1) * 2048];
  attn_body<X3>(P, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// two problems with the same (clips, heads) grid in one launch; the query tiles of problem 0 come first
template <bool X3>
__global__ __launch_bounds__(256) void self_attention_pair_kernel(const AttnProb P0, const AttnProb P1, const int qt0) {
  __shared__ __attribute__((aligned(1024))) char lds[(X3 ? 2 : 1) * KC * 128 * 2 + 4 * (X3 ? 2 : 1) * 2048];
  if ((int)blockIdx.x < qt0) attn_body<X3>(P0, blockIdx.x, blockIdx.y, blockIdx.z, lds);
  else attn_body<X3>(P1, blockIdx.x - qt0, blockIdx.y, blockIdx.z, lds);
}

}  // namespace

int ser_launch_self_attention_pair(const SerAttnArgs& a, const SerAttnArgs& b, hipStream_t st) {
  const bool x3 = a.qkv_lo && a.ctx_lo;
  if (a.B != b.B || a.heads != b.heads || x3 != (b.qkv_lo && b.ctx_lo)) {
    SER_TRY(ser_launch_self_attention(a.qkv_hi, a.qkv_lo, a.key_mask, a.B, a.S, a.heads, a.ctx_hi, a.ctx_lo, st));
    return ser_launch_self_attention(b.qkv_hi, b.qkv_lo, b.key_mask, b.B, b.S, b.heads, b.ctx_hi, b.ctx_lo, st);
  }
  SER_REQUIRE(a.B > 0 && a.S > 0 && b.S > 0 && a.heads > 0, "self_attention: empty problem");
  const int H = a.heads * HD, qt0 = ceil_div(a.S, 64);
  const AttnProb P0{a.qkv_hi, x3 ? a.qkv_lo : nullptr, a.key_mask, a.S, H, a.ctx_hi, x3 ? a.ctx_lo : nullptr};
  const AttnProb P1{b.qkv_hi, x3 ? b.qkv_lo : nullptr, b.key_mask, b.S, H, b.ctx_hi, x3 ? b.ctx_lo : nullptr};
  dim3 grid(qt0 + ceil_div(b.S, 64), a.heads, a.B), block(256);
  if (x3) hipLaunchKernelGGL(self_attention_pair_kernel<true>, grid, block, 0, st, P0, P1, qt0);
  else hipLaunchKernelGGL(self_attention_pair_kernel<false>, grid, block, 0, st, P0, P1, qt0);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

int ser_launch_self_attention(const bf16_t* qkv_hi, const bf16_t* qkv_lo, const float* key_mask, int B, int S,
                              int heads, bf16_t* ctx_hi, bf16_t* ctx_lo, hipStream_t st) {
  SER_REQUIRE(B > 0 && S > 0 && heads > 0, "self_attention: empty problem");
  SER_REQUIRE(qkv_hi && ctx_hi, "self_attention: null planes");
  const int H = heads * HD;
  dim3 grid(ceil_div(S, 64), heads, B), block(256);
  if (qkv_lo && ctx_lo) {
    const AttnProb P{qkv_hi, qkv_lo, key_mask, S, H, ctx_hi, ctx_lo};
    hipLaunchKernelGGL(self_attention_kernel<true>, grid, block, 0, st, P);
  } else {
    const AttnProb P{qkv_hi, nullptr, key_mask, S, H, ctx_hi, nullptr};
    hipLaunchKernelGGL(self_attention_kernel<false>, grid, block, 0, st, P);
  }
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_self_attention(const uint16_t* qkv_hi, const uint16_t* qkv_lo, const float* key_mask, int B, int S,
                                  int heads, uint16_t* ctx_hi, uint16_t* ctx_lo, void* stream) {
  return ser_launch_self_attention(qkv_hi, qkv_lo, key_mask, B, S, heads, ctx_hi, ctx_lo, (hipStream_t)stream);
}
