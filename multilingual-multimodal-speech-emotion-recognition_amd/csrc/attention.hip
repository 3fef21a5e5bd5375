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
#include <stdlib.h>

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
  // interleaved planes (lo == hi + 32, ser_common.h): flat element offsets map by ser_il_off; every access below is a
  // run of 8 elements inside one 32-group, so it stays one 16-byte access
  const bool il_in = X3 && ser_is_il(qkv_hi, qkv_lo), il_out = X3 && ser_is_il(ctx_hi, ctx_lo);

  // Q fragments straight from global: row q0+fr, d = ks*32 + fq*8 .. +8
  bf16x8 qh[2], ql[2];
  {
    int qr = q0 + fr;
    qr = qr < S ? qr : S - 1;
    const long long base = ((long long)b * S + qr) * ld + head * HD;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const long long e = il_in ? ser_il_off(base + ks * 32 + fq * 8) : base + ks * 32 + fq * 8;
      qh[ks] = *(const bf16x8*)(qkv_hi + e);
      if (X3) ql[ks] = *(const bf16x8*)(qkv_lo + e);
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
        const bf16x8 kv = *(const bf16x8*)(plane[p] + (il_in ? ser_il_off(rb + H) : rb + H));
        *(bf16x8*)(Ks + p * TILE + tile_off(key, ch)) = kv;
        const bf16x8 vv = *(const bf16x8*)(plane[p] + (il_in ? ser_il_off(rb + 2 * H) : rb + 2 * H));
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
      long long off = ((long long)b * S + q) * H + head * HD + jd * 16 + fr;
      if (il_out) off = ser_il_off(off);
      bf16_t h, lo_;
      split_bf16(o[jd][r] * inv, h, lo_);
      ctx_hi[off] = h;
      if (X3) ctx_lo[off] = lo_;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Short sequences (S <= 224: every clip length of the benchmark configs except the 10 s stress case): ONE workgroup of
// 8 waves per (clip, head) keeps all keys and values of the head in LDS, staged once, in the row order they have in
// global memory (no transposed copy, no per-query-block restaging).  Each wave owns 16-query blocks:
//   * scores are computed SWAPPED, S^T = K . Q^T (A = K fragment from LDS, B = Q fragment from global), so a lane holds
//     scores of ONE query (q = lane & 15) for keys 16 jb + 4 (lane >> 4) + r: the softmax of a row is lane-local apart
//     from two cross-lane steps over the four lane groups, and all NKB x 4 scores of the row stay in registers (exact
//     two-pass softmax, no online rescaling);
//   * P never goes through LDS: the contraction index of P.V may be any permutation of the keys as long as both
//     operands use it, so the lane's own eight probabilities of key blocks 2 ks and 2 ks + 1 ARE its A fragment for
//     k-step ks, and the matching V fragment is fetched by two ds_read_b64_tr_b16 (hardware transpose, gfx950) of the
//     4-key x 16-column blocks at keys 16 (2 ks [+ 1]) + 4 g from the row-major V image;
//   * the output block goes through a per-wave LDS tile and leaves as whole 128 / 256-byte rows.
// LDS rows are 128 B (one plane) or 256 B (interleaved hi / lo, the global layout); the 16-byte chunk index is XORed with
// row & 7 (resp. (row & 7) << 1), which makes the ds_read_b128 K reads and the transposed V reads conflict-free.
constexpr int SA_MAXKEYS = 224, SA_NKB = SA_MAXKEYS / 16, SA_WAVES = 8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <bool X3>
struct SaCfg {
  static constexpr int RB = X3 ? 256 : 128;                      // bytes per key row in LDS
  static constexpr int CH = RB / 16;
  static constexpr int LDS_BYTES = 2 * SA_MAXKEYS * RB + 1024 + SA_WAVES * 16 * RB;
};
template <bool X3>
SER_DEVFN int sa_off(int row, int ch) { return row * SaCfg<X3>::RB + ((ch ^ (X3 ? ((row & 7) << 1) : (row & 7))) << 4); }
// chunk of (plane p, 8-element d-chunk dc = d >> 3) inside a row
template <bool X3>
SER_DEVFN int sa_chunk(int p, int dc) { return X3 ? ((dc >> 2) << 3) + (p << 2) + (dc & 3) : dc; }

template <bool X3>
SER_DEVFN void attn_small_body(const AttnProb& P, const int head, const int b, char* lds) {
  constexpr int RB = SaCfg<X3>::RB, CH = SaCfg<X3>::CH;
  const int S = P.S, H = P.H;
  const int nkb = (S + 15) >> 4;                 // 16-key (and 16-query) blocks
  const int nks = (nkb + 1) >> 1;                // 32-key k-steps of P.V; rows up to 32 nks are staged (zeros beyond S)
  char* Ks = lds;
  char* Vs = lds + SA_MAXKEYS * RB;
  float* kbias = (float*)(lds + 2 * SA_MAXKEYS * RB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* Os = lds + 2 * SA_MAXKEYS * RB + 1024 + wave * 16 * RB;
  const int fr = lane & 15, fq = lane >> 4;
  const long long ld = 3LL * H;
  const bool il_in = X3 && ser_is_il(P.qkv_hi, P.qkv_lo), il_out = X3 && ser_is_il(P.ctx_hi, P.ctx_lo);
  (void)il_in;

  // ---- stage K and V rows of this (clip, head): row = key, chunks in global order
  const int nrows = nks * 32;
  for (int idx = tid; idx < nrows * CH; idx += SA_WAVES * 64) {
    const int row = idx / CH, c = idx % CH;
    bf16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < S) {
      const long long base = ((long long)b * S + row) * ld + head * HD;          // logical element offset of Q; K at +H, V at +2H
      if (X3) {       // interleaved: the head's 64 values are 128 contiguous physical elements [hi 32 | lo 32 | hi 32 | lo 32]
        kv = *(const bf16x8*)(P.qkv_hi + ser_il_off(base + H) + c * 8);
        vv = *(const bf16x8*)(P.qkv_hi + ser_il_off(base + 2 * H) + c * 8);
      } else {
        kv = *(const bf16x8*)(P.qkv_hi + base + H + c * 8);
        vv = *(const bf16x8*)(P.qkv_hi + base + 2 * H + c * 8);
      }
    }
    *(bf16x8*)(Ks + sa_off<X3>(row, c)) = kv;
    *(bf16x8*)(Vs + sa_off<X3>(row, c)) = vv;
  }
  for (int k = tid; k < 256; k += SA_WAVES * 64) {
    bool ok = k < S;
    if (ok && P.key_mask) ok = P.key_mask[(long long)b * S + k] != 0.f;
    kbias[k] = ok ? 0.f : -INFINITY;
  }
  __syncthreads();

  for (int qb = wave; qb < nkb; qb += SA_WAVES) {
    // Q fragments (B operand of the swapped product): query qb*16 + fr, d = ks*32 + fq*8 .. +8
    bf16x8 qh[2], ql[2];
    {
      int qr = qb * 16 + fr;
      qr = qr < S ? qr : S - 1;
      const long long base = ((long long)b * S + qr) * ld + head * HD;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (X3) {
          const long long e = ser_il_off(base + ks * 32 + fq * 8);
          qh[ks] = *(const bf16x8*)(P.qkv_hi + e);
          ql[ks] = *(const bf16x8*)(P.qkv_hi + e + SER_IL_GROUP);
        } else {
          qh[ks] = *(const bf16x8*)(P.qkv_hi + base + ks * 32 + fq * 8);
        }
      }
    }
    // ---- scores^T: sc[jb][r] = <K[16 jb + 4 fq + r], Q[qb*16 + fr]>
    f32x4 sc[SA_NKB];
#pragma unroll
    for (int jb = 0; jb < SA_NKB; ++jb) {
      sc[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (jb < nkb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 kh = *(const bf16x8*)(Ks + sa_off<X3>(jb * 16 + fr, sa_chunk<X3>(0, ks * 4 + fq)));
          if (X3) {
            const bf16x8 kl = *(const bf16x8*)(Ks + sa_off<X3>(jb * 16 + fr, sa_chunk<X3>(1, ks * 4 + fq)));
            sc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[ks], sc[jb], 0, 0, 0);
            sc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[ks], sc[jb], 0, 0, 0);
          }
          sc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[ks], sc[jb], 0, 0, 0);
        }
      }
    }
    // ---- softmax of row q = fr over all keys: lane-local over (jb, r), then over the four lane groups
    float mx = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < SA_NKB; ++jb)
      if (jb < nkb) {
        const float4 kb4 = *(const float4*)(kbias + jb * 16 + fq * 4);
        sc[jb][0] = sc[jb][0] * 0.125f + kb4.x;
        sc[jb][1] = sc[jb][1] * 0.125f + kb4.y;
        sc[jb][2] = sc[jb][2] * 0.125f + kb4.z;
        sc[jb][3] = sc[jb][3] * 0.125f + kb4.w;
        mx = fmaxf(mx, fmaxf(fmaxf(sc[jb][0], sc[jb][1]), fmaxf(sc[jb][2], sc[jb][3])));
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mu = mx == -INFINITY ? 0.f : mx;
    float sum = 0.f;
#pragma unroll
    for (int jb = 0; jb < SA_NKB; ++jb)
      if (jb < nkb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __expf(sc[jb][r] - mu);
          sc[jb][r] = pv;
          sum += pv;
        }
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;

    // ---- O = P . V with the key permutation described above
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tq = fr >> 2, tp = fr & 3;                       // transposed read: this lane addresses block row tq, columns 4 tp .. +3
#pragma unroll
    for (int ks = 0; ks < SA_NKB / 2; ++ks) {
      if (ks < nks) {
        // A fragment: own probabilities of key blocks 2 ks and 2 ks + 1 (block 2 ks + 1 may lie beyond nkb: zeros)
        uint32_t ph[4], pl[4];
        const f32x4 p0 = sc[2 * ks], p1 = sc[2 * ks + 1];       // sc[jb >= nkb] is 0
        if (X3) {
          split_bf16x2(p0[0] * inv, p0[1] * inv, ph[0], pl[0]);
          split_bf16x2(p0[2] * inv, p0[3] * inv, ph[1], pl[1]);
          split_bf16x2(p1[0] * inv, p1[1] * inv, ph[2], pl[2]);
          split_bf16x2(p1[2] * inv, p1[3] * inv, ph[3], pl[3]);
        } else {
          ph[0] = pack_bf16x2(p0[0] * inv, p0[1] * inv);
          ph[1] = pack_bf16x2(p0[2] * inv, p0[3] * inv);
          ph[2] = pack_bf16x2(p1[0] * inv, p1[1] * inv);
          ph[3] = pack_bf16x2(p1[2] * inv, p1[3] * inv);
        }
        const bf16x8 pah = __builtin_bit_cast(bf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3]));
        bf16x8 pal;
        if (X3) pal = __builtin_bit_cast(bf16x8, make_uint4(pl[0], pl[1], pl[2], pl[3]));
        const int r0 = ks * 32 + fq * 4 + tq, r1 = r0 + 16;    // V rows (keys) this lane addresses in the two blocks
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const int dc = db * 2 + (tp >> 1), hb = (tp & 1) * 8;
          const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Vs + sa_off<X3>(r0, sa_chunk<X3>(0, dc)) + hb));
          const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Vs + sa_off<X3>(r1, sa_chunk<X3>(0, dc)) + hb));
          const bf16x8 vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          if (X3) {
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Vs + sa_off<X3>(r0, sa_chunk<X3>(1, dc)) + hb));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Vs + sa_off<X3>(r1, sa_chunk<X3>(1, dc)) + hb));
            const bf16x8 vl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pal, vh, o[db], 0, 0, 0);
            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pah, vl, o[db], 0, 0, 0);
          }
          o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pah, vh, o[db], 0, 0, 0);
        }
      }
    }
    // ---- output block [16 q][64 d] -> per-wave LDS tile in the global chunk order -> whole-row stores
    // (o[db][r]: row q = fq*4 + r, column d = db*16 + fr)
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = fq * 4 + r, d = db * 16 + fr;
        bf16_t h, lo_;
        split_bf16(o[db][r], h, lo_);
        *(bf16_t*)(Os + row * RB + sa_chunk<X3>(0, d >> 3) * 16 + (d & 7) * 2) = h;
        if (X3) *(bf16_t*)(Os + row * RB + sa_chunk<X3>(1, d >> 3) * 16 + (d & 7) * 2) = lo_;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 16 * CH / 64; ++it) {
      const int idx = lane + 64 * it, row = idx / CH, c = idx % CH;
      const int q = qb * 16 + row;
      if (q < S) {
        const uint4 v = *(const uint4*)(Os + row * RB + c * 16);
        const long long base = ((long long)b * S + q) * H + head * HD;
        if (X3) {
          if (il_out) {
            *(uint4*)(P.ctx_hi + ser_il_off(base) + c * 8) = v;
          } else {      // planar output planes: chunk c = (dc >> 2) * 8 + plane * 4 + (dc & 3)
            const int plane = (c >> 2) & 1, dc = ((c >> 3) << 2) + (c & 3);
            *(uint4*)((plane ? P.ctx_lo : P.ctx_hi) + base + dc * 8) = v;
          }
        } else {
          *(uint4*)(P.ctx_hi + base + c * 8) = v;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();      // the tile is reused by this wave's next query block
  }
}

// ---- the same kernel with HALF the LDS footprint (two workgroups per CU, or one beside a GEMM workgroup of the other
// stream): K and V take turns in one region.  Phase A: every wave computes the scores and the softmax of ALL its query
// blocks (at most two) while K is resident - the probabilities stay in registers (2 x 56 VGPRs); barrier; phase B: V, which
// every thread fetched into registers at the very start, replaces K in LDS; barrier; phase C: P.V and the output, staged
// per 32-column half through a 2 KB tile per wave.  Products and their order per accumulator are those of
// attn_small_body: results are bit-identical (tests/test_gpu_ops.py).
template <bool X3>
struct Sa2Cfg {
  static constexpr int RB = SaCfg<X3>::RB;
  static constexpr int OT = 16 * (RB / 2);                        // per-wave output tile: 16 rows x one 32-column half
  static constexpr int LDS_BYTES = SA_MAXKEYS * RB + 1024 + SA_WAVES * OT;
};

template <bool X3>
SER_DEVFN void attn_small2_body(const AttnProb& P, const int head, const int b, char* lds) {
  constexpr int RB = SaCfg<X3>::RB, CH = SaCfg<X3>::CH, QPW = (SA_NKB + SA_WAVES - 1) / SA_WAVES;   // query blocks per wave
  const int S = P.S, H = P.H;
  const int nkb = (S + 15) >> 4;
  const int nks = (nkb + 1) >> 1;
  char* KV = lds;
  float* kbias = (float*)(lds + SA_MAXKEYS * RB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* Os = lds + SA_MAXKEYS * RB + 1024 + wave * Sa2Cfg<X3>::OT;
  const int fr = lane & 15, fq = lane >> 4;
  const long long ld = 3LL * H;
  const bool il_out = X3 && ser_is_il(P.ctx_hi, P.ctx_lo);
  const int nrows = nks * 32;
  constexpr int IT = (SA_MAXKEYS * CH + SA_WAVES * 64 - 1) / (SA_WAVES * 64);

  // ---- K -> LDS, V -> registers (stored after phase A)
  bf16x8 vreg[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int idx = tid + it * SA_WAVES * 64, row = idx / CH, c = idx % CH;
    bf16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0};
    vreg[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (idx < nrows * CH && row < S) {
      const long long base = ((long long)b * S + row) * ld + head * HD;
      if (X3) {
        kv = *(const bf16x8*)(P.qkv_hi + ser_il_off(base + H) + c * 8);
        vreg[it] = *(const bf16x8*)(P.qkv_hi + ser_il_off(base + 2 * H) + c * 8);
      } else {
        kv = *(const bf16x8*)(P.qkv_hi + base + H + c * 8);
        vreg[it] = *(const bf16x8*)(P.qkv_hi + base + 2 * H + c * 8);
      }
    }
    if (idx < nrows * CH) *(bf16x8*)(KV + sa_off<X3>(row, c)) = kv;
  }
  for (int k = tid; k < 256; k += SA_WAVES * 64) {
    bool ok = k < S;
    if (ok && P.key_mask) ok = P.key_mask[(long long)b * S + k] != 0.f;
    kbias[k] = ok ? 0.f : -INFINITY;
  }
  __syncthreads();

  // ---- phase A: scores^T and softmax of my query blocks (sc[i][jb][r] = <K[16 jb + 4 fq + r], Q[qb*16 + fr]>)
  f32x4 sc[QPW][SA_NKB];
  float inv[QPW];
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int qb = wave + SA_WAVES * i;
    inv[i] = 0.f;
#pragma unroll
    for (int jb = 0; jb < SA_NKB; ++jb) sc[i][jb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (qb < nkb) {
      bf16x8 qh[2], ql[2];
      {
        int qr = qb * 16 + fr;
        qr = qr < S ? qr : S - 1;
        const long long base = ((long long)b * S + qr) * ld + head * HD;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (X3) {
            const long long e = ser_il_off(base + ks * 32 + fq * 8);
            qh[ks] = *(const bf16x8*)(P.qkv_hi + e);
            ql[ks] = *(const bf16x8*)(P.qkv_hi + e + SER_IL_GROUP);
          } else {
            qh[ks] = *(const bf16x8*)(P.qkv_hi + base + ks * 32 + fq * 8);
          }
        }
      }
#pragma unroll
      for (int jb = 0; jb < SA_NKB; ++jb) {
        if (jb < nkb) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 kh = *(const bf16x8*)(KV + sa_off<X3>(jb * 16 + fr, sa_chunk<X3>(0, ks * 4 + fq)));
            if (X3) {
              const bf16x8 kl = *(const bf16x8*)(KV + sa_off<X3>(jb * 16 + fr, sa_chunk<X3>(1, ks * 4 + fq)));
              sc[i][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[ks], sc[i][jb], 0, 0, 0);
              sc[i][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[ks], sc[i][jb], 0, 0, 0);
            }
            sc[i][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[ks], sc[i][jb], 0, 0, 0);
          }
        }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int jb = 0; jb < SA_NKB; ++jb)
        if (jb < nkb) {
          const float4 kb4 = *(const float4*)(kbias + jb * 16 + fq * 4);
          sc[i][jb][0] = sc[i][jb][0] * 0.125f + kb4.x;
          sc[i][jb][1] = sc[i][jb][1] * 0.125f + kb4.y;
          sc[i][jb][2] = sc[i][jb][2] * 0.125f + kb4.z;
          sc[i][jb][3] = sc[i][jb][3] * 0.125f + kb4.w;
          mx = fmaxf(mx, fmaxf(fmaxf(sc[i][jb][0], sc[i][jb][1]), fmaxf(sc[i][jb][2], sc[i][jb][3])));
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mu = mx == -INFINITY ? 0.f : mx;
      float sum = 0.f;
#pragma unroll
      for (int jb = 0; jb < SA_NKB; ++jb)
        if (jb < nkb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __expf(sc[i][jb][r] - mu);
            sc[i][jb][r] = pv;
            sum += pv;
          }
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      inv[i] = sum > 0.f ? 1.0f / sum : 0.f;
    }
  }
  __syncthreads();                 // every wave is done with K

  // ---- phase B: V replaces K
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int idx = tid + it * SA_WAVES * 64, row = idx / CH, c = idx % CH;
    if (idx < nrows * CH) *(bf16x8*)(KV + sa_off<X3>(row, c)) = vreg[it];
  }
  __syncthreads();

  // ---- phase C: O = P . V (key permutation as in attn_small_body), output per 32-column half
  const int tq = fr >> 2, tp = fr & 3;
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int qb = wave + SA_WAVES * i;
    if (qb >= nkb) continue;
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < SA_NKB / 2; ++ks) {
      if (ks < nks) {
        uint32_t ph[4], pl[4];
        const f32x4 p0 = sc[i][2 * ks], p1 = sc[i][2 * ks + 1];
        if (X3) {
          split_bf16x2(p0[0] * inv[i], p0[1] * inv[i], ph[0], pl[0]);
          split_bf16x2(p0[2] * inv[i], p0[3] * inv[i], ph[1], pl[1]);
          split_bf16x2(p1[0] * inv[i], p1[1] * inv[i], ph[2], pl[2]);
          split_bf16x2(p1[2] * inv[i], p1[3] * inv[i], ph[3], pl[3]);
        } else {
          ph[0] = pack_bf16x2(p0[0] * inv[i], p0[1] * inv[i]);
          ph[1] = pack_bf16x2(p0[2] * inv[i], p0[3] * inv[i]);
          ph[2] = pack_bf16x2(p1[0] * inv[i], p1[1] * inv[i]);
          ph[3] = pack_bf16x2(p1[2] * inv[i], p1[3] * inv[i]);
        }
        const bf16x8 pah = __builtin_bit_cast(bf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3]));
        bf16x8 pal;
        if (X3) pal = __builtin_bit_cast(bf16x8, make_uint4(pl[0], pl[1], pl[2], pl[3]));
        const int r0 = ks * 32 + fq * 4 + tq, r1 = r0 + 16;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const int dc = db * 2 + (tp >> 1), hb = (tp & 1) * 8;
          const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(KV + sa_off<X3>(r0, sa_chunk<X3>(0, dc)) + hb));
          const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(KV + sa_off<X3>(r1, sa_chunk<X3>(0, dc)) + hb));
          const bf16x8 vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          if (X3) {
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(KV + sa_off<X3>(r0, sa_chunk<X3>(1, dc)) + hb));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(KV + sa_off<X3>(r1, sa_chunk<X3>(1, dc)) + hb));
            const bf16x8 vl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pal, vh, o[db], 0, 0, 0);
            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pah, vl, o[db], 0, 0, 0);
          }
          o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pah, vh, o[db], 0, 0, 0);
        }
      }
    }
    // output block [16 q][64 d], one 32-column half at a time: tile row = [hi 32 | lo 32] (X3) or [32 values]
    constexpr int HB = RB / 2;                                   // bytes of a half row in the tile and in the global row
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {
        const int db = hf * 2 + dd;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = fq * 4 + r, dloc = dd * 16 + fr;           // column inside the half
          bf16_t h, lo_;
          split_bf16(o[db][r], h, lo_);
          *(bf16_t*)(Os + row * HB + dloc * 2) = h;
          if (X3) *(bf16_t*)(Os + row * HB + 64 + dloc * 2) = lo_;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      constexpr int CPR = HB / 16;                               // 16-byte chunks per half row
#pragma unroll
      for (int it = 0; it < 16 * CPR / 64; ++it) {
        const int idx = lane + 64 * it, row = idx / CPR, c = idx % CPR;
        const int q = qb * 16 + row;
        if (q < S) {
          const uint4 v = *(const uint4*)(Os + row * HB + c * 16);
          const long long base = ((long long)b * S + q) * H + head * HD;
          if (X3) {
            if (il_out) {
              *(uint4*)(P.ctx_hi + ser_il_off(base) + hf * 64 + c * 8) = v;
            } else {                                             // planar planes: chunks 0-3 = hi, 4-7 = lo of columns 32 hf ..
              *(uint4*)(((c >> 2) ? P.ctx_lo : P.ctx_hi) + base + hf * 32 + (c & 3) * 8) = v;
            }
          } else {
            *(uint4*)(P.ctx_hi + base + hf * 32 + c * 8) = v;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();                           // the tile is reused by the next half / query block
    }
  }
}

template <bool X3>
__global__ __launch_bounds__(SA_WAVES * 64) void self_attention_small2_kernel(const AttnProb P0, const AttnProb P1, const int n0,
                                                                                const int heads) {
  __shared__ __attribute__((aligned(1024))) char lds[Sa2Cfg<X3>::LDS_BYTES];
  const int blk = blockIdx.x;
  if (blk < n0) attn_small2_body<X3>(P0, blk % heads, blk / heads, lds);
  else attn_small2_body<X3>(P1, (blk - n0) % heads, (blk - n0) / heads, lds);
}

// one or two problems (same clip count and head count) in one launch; the (clip, head) blocks of problem 0 come first
template <bool X3>
__global__ __launch_bounds__(SA_WAVES * 64) void self_attention_small_kernel(const AttnProb P0, const AttnProb P1, const int n0,
                                                                               const int heads) {
  __shared__ __attribute__((aligned(1024))) char lds[SaCfg<X3>::LDS_BYTES];
  const int blk = blockIdx.x;
  if (blk < n0) attn_small_body<X3>(P0, blk % heads, blk / heads, lds);
  else attn_small_body<X3>(P1, (blk - n0) % heads, (blk - n0) / heads, lds);
}

// the resident-K/V kernel takes one-plane problems and three-product problems whose INPUT planes are interleaved
static bool small_ok(const bf16_t* qkv_hi, const bf16_t* qkv_lo, const bf16_t* ctx_lo, int S) {
  if (S > SA_MAXKEYS) return false;
  const bool x3 = qkv_lo && ctx_lo;
  return !x3 || ser_is_il(qkv_hi, qkv_lo);
}
static int g_attn_force_generic = 0;      // tests: run the chunked kernel on shapes the resident kernel would take
static int attn_variant_default() {
  const char* e = getenv("SER_ATTN_VARIANT");
  return e && e[0] == '1' ? 1 : 2;
}
static int g_attn_small_variant = attn_variant_default();      // 2: K and V share one LDS region (two workgroups per CU); 1: both resident
extern "C" int ser_debug_set_attention_small_variant(int v) { g_attn_small_variant = v; return 0; }
extern "C" int ser_debug_set_attention_generic(int on) { g_attn_force_generic = on; return 0; }

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
  if (!g_attn_force_generic && small_ok(a.qkv_hi, a.qkv_lo, a.ctx_lo, a.S) && small_ok(b.qkv_hi, b.qkv_lo, b.ctx_lo, b.S)) {
    const int Hh = a.heads * HD, n0 = a.B * a.heads;
    const AttnProb Q0{a.qkv_hi, x3 ? a.qkv_lo : nullptr, a.key_mask, a.S, Hh, a.ctx_hi, x3 ? a.ctx_lo : nullptr};
    const AttnProb Q1{b.qkv_hi, x3 ? b.qkv_lo : nullptr, b.key_mask, b.S, Hh, b.ctx_hi, x3 ? b.ctx_lo : nullptr};
    if (g_attn_small_variant == 2) {
      if (x3) hipLaunchKernelGGL(self_attention_small2_kernel<true>, dim3(2 * n0), dim3(SA_WAVES * 64), 0, st, Q0, Q1, n0, a.heads);
      else hipLaunchKernelGGL(self_attention_small2_kernel<false>, dim3(2 * n0), dim3(SA_WAVES * 64), 0, st, Q0, Q1, n0, a.heads);
    } else if (x3) hipLaunchKernelGGL(self_attention_small_kernel<true>, dim3(2 * n0), dim3(SA_WAVES * 64), 0, st, Q0, Q1, n0, a.heads);
    else hipLaunchKernelGGL(self_attention_small_kernel<false>, dim3(2 * n0), dim3(SA_WAVES * 64), 0, st, Q0, Q1, n0, a.heads);
    SER_LAUNCH_CHECK();
    return SER_OK;
  }
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
  if (!g_attn_force_generic && small_ok(qkv_hi, qkv_lo, ctx_lo, S)) {
    const bool x3 = qkv_lo && ctx_lo;
    const AttnProb Q{qkv_hi, x3 ? qkv_lo : nullptr, key_mask, S, H, ctx_hi, x3 ? ctx_lo : nullptr};
    if (g_attn_small_variant == 2) {
      if (x3) hipLaunchKernelGGL(self_attention_small2_kernel<true>, dim3(B * heads), dim3(SA_WAVES * 64), 0, st, Q, Q, B * heads, heads);
      else hipLaunchKernelGGL(self_attention_small2_kernel<false>, dim3(B * heads), dim3(SA_WAVES * 64), 0, st, Q, Q, B * heads, heads);
    } else if (x3) hipLaunchKernelGGL(self_attention_small_kernel<true>, dim3(B * heads), dim3(SA_WAVES * 64), 0, st, Q, Q, B * heads, heads);
    else hipLaunchKernelGGL(self_attention_small_kernel<false>, dim3(B * heads), dim3(SA_WAVES * 64), 0, st, Q, Q, B * heads, heads);
    SER_LAUNCH_CHECK();
    return SER_OK;
  }
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
