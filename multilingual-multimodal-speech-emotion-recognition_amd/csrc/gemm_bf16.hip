// Split-bf16 NT GEMM for the frozen encoders (gfx950 / CDNA4).
//
//   C[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual        fp32 accumulate on MFMA
//
// Operands are bf16 planes with K contiguous.  In parity mode (both hi and lo planes present)
// every product is  a_hi*w_hi + a_lo*w_hi + a_hi*w_lo  (three v_mfma_f32_16x16x32_bf16), which
// carries ~2^-16 relative error per product instead of bf16's 2^-8; in fast mode only the hi
// planes are read.  Three operand modes:
//   M_BF16  hi planes only, one product, a k-tile = 64 values of k (128 B per row)
//   M_X3P   planar hi and lo planes (the positional conv's Toeplitz view needs it), both staged: two LDS images per operand
//   M_X3I   interleaved hi/lo (ser_common.h): a 128-B line = 32 values of k with both planes, so a k-tile stages exactly
//           the bytes of a bf16 k-tile and feeds 3 MFMAs per accumulator instead of 2 — same LDS footprint, same tile
//           heights, two workgroups per CU; the summation order equals M_X3P's (bit-identical results).
//
// Structure: BMxBN output tile per 256-thread workgroup (4 waves as 2x2; BM = 64...192, BN = 128 or 64), BK = 64.
// Global -> LDS goes through global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip): one wave
// instruction moves 8 rows x 128 B.  The LDS image is lane-linear, so the bank-conflict swizzle
// (16-byte chunk index XOR row&7) is applied to the per-lane SOURCE address and again on the
// ds_read_b128 side (cdna_hip_programming.md rule 21).  NS LDS buffers (2, or 3 where measured to pay): the loads of
// k-tiles t+1 .. t+NS-1 are in flight while tile t is multiplied; one counted s_waitcnt vmcnt + s_barrier per k-tile.
// Inside a k-tile the fragment reads of k-step 1 are issued under the MFMAs of k-step 0 (pinned with sched_barrier).
// Epilogue: accumulators -> LDS half tile -> 16-byte row-coalesced stores with bias / GELU / residual / bf16 plane split.
#include <type_traits>
#include <vector>
#include "ser_common.h"

namespace {

// Phase probe (scripts/gemm_phase_probe.py builds side libraries with -DSER_GEMM_DIAG=n; the product build has 0):
// bit 0 drops the epilogue, bit 1 the fragment reads + MFMAs, bit 2 the global->LDS staging, bit 3 the epilogue's
// global stores (its LDS pass and arithmetic stay).  Timing only — the
// results of a probe build are meaningless.
#ifndef SER_GEMM_DIAG
#define SER_GEMM_DIAG 0
#endif
constexpr int BK = 64;           // bf16 elements per k-tile = 128 B per row
constexpr int ROW_BYTES = 128;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Stage R rows x 64 bf16 of a K-contiguous matrix into a swizzled LDS tile.
// rows beyond `rmax` are clamped (their products are discarded by the epilogue).
// The per-lane part of the source address is a 32-bit byte offset computed ONCE per tile (stage_offsets); per k-tile
// only the wave-uniform base moves.  That keeps the address VGPRs of in-flight LDS-DMA instructions untouched, so
// the compiler has no reason to drain the DMA queue (vmcnt(0)) before issuing the next stage.
template <int R, int NW = 4>
SER_DEVFN void stage_offsets(unsigned (&off)[R / (8 * NW)], long long ld, int row0, int rmax, int wave, int lane) {
  const int r = lane >> 3;
  const int c = (lane & 7) ^ r;   // source chunk for LDS chunk position lane&7 of row r
#pragma unroll
  for (int i = 0; i < R / (8 * NW); ++i) {
    const int p = wave + NW * i;
    int row = row0 + p * 8 + r;
    row = row < rmax ? row : rmax;
    off[i] = (unsigned)(((long long)row * ld + c * 8) * 2);
  }
}
template <int R, int NW = 4>
SER_DEVFN void stage_tile(const bf16_t* __restrict__ base_k, const unsigned (&off)[R / (8 * NW)], char* lds_tile, int wave) {
#pragma unroll
  for (int i = 0; i < R / (8 * NW); ++i) {      // constant trip count: no scalar branches between the LDS-DMA issues
    const int p = wave + NW * i;
    const char* src = (const char*)base_k + off[i];
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + p * 1024), 16, 0, 0);
  }
}

// 16-byte fragment of row `row` (tile-local), 16-B chunk `chunk` (0..7)
SER_DEVFN bf16x8 lds_frag(const char* tile, int row, int chunk) {
  const int off = (row >> 3) * 1024 + (row & 7) * ROW_BYTES + ((chunk ^ (row & 7)) << 4);
  return *(const bf16x8*)(tile + off);
}

SER_DEVFN float apply_act(float v, int act) {
  if (act == SER_ACT_GELU) return gelu_erf(v);
  if (act == SER_ACT_RELU) return fmaxf(v, 0.0f);
  return v;
}

template <int ACT>
SER_DEVFN float act_const(float v) {
  return ACT == SER_ACT_GELU ? gelu_erf(v) : ACT == SER_ACT_RELU ? fmaxf(v, 0.0f) : v;
}

enum { M_BF16 = 0, M_X3P = 1, M_X3I = 2 };

// Waves form a WR x WC grid over the BM x BN tile (wave tile (BM/WR) x (BN/WC)).  2 x 2 (256 threads) is the classic
// form, two workgroups per CU; 2 x 4 (512 threads, BN = 256, one workgroup per CU, two waves per SIMD) halves the bytes
// staged per MFMA: on this chip a CU takes in ~45-58 GB/s through global->LDS whatever the L2 hit rate
// (scripts/gemm_il_probe.py), so the tile's area / perimeter ratio, not the MFMA rate, bounds a 128-wide tile.
template <int P, int NP, class F>
SER_DEVFN void static_for(F&& f) {
  if constexpr (P < NP) {
    f(std::integral_constant<int, P>{});
    static_for<P + 1, NP>(f);
  }
}

template <int BM, int BN, int MODE, int NS = 2, int WR = 2, int WC = 2>
struct GemmCfg {
  static constexpr int NW = WR * WC, NT = 64 * NW;
  static constexpr int NPL = MODE == M_X3P ? 2 : 1;          // LDS images per operand
  static constexpr int A_TILE = BM * ROW_BYTES, W_TILE = BN * ROW_BYTES;
  static constexpr int STAGE = (A_TILE + W_TILE) * NPL;
  // rows per epilogue pass: a wave-row's rows in one pass when its fp32 image fits the staging LDS, else half of them
  // (a pass never straddles two wave-rows)
  static constexpr int EPR = (BM / WR) * (BN + 4) * 4 <= (NS < 2 ? 2 : NS) * STAGE || NW == 4 ? BM / WR : BM / WR / 2;
  static_assert((BM / WR) % EPR == 0 && EPR % 16 == 0, "epilogue pass rows");
  static constexpr int EPI_BYTES = EPR * (BN + 4) * 4;        // fp32 rows staged for the coalesced epilogue
  static constexpr int LDS_RAW = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
  static constexpr int GLDS = (BM / (8 * NW) + BN / (8 * NW)) * NPL;       // LDS-DMA instructions per stage per wave
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && (BM / WR) % 16 == 0 && (BN / WC) % 16 == 0, "tile / wave grid");
  // leave >= 24 KB of every CU's 160 KB LDS unclaimed: the head kernels of the previous batch run beside these
  // GEMMs on another stream, and a small workgroup that cannot get LDS waits for a whole GEMM workgroup to retire
  static constexpr int LDS_BYTES = LDS_RAW <= 32 * 1024 ? 34 * 1024 : LDS_RAW;
  // register budget: where LDS lets two workgroups share a CU, VGPRs + AGPRs must stay within 256 per lane
  // single-buffer form: three workgroups per CU (four for the 64-row tile) — set by registers (<= 168 / 128 per lane)
  static constexpr int WG_PER_CU = NS == 1 ? (BM <= 64 ? 4 : 3) : (2 * LDS_BYTES <= 160 * 1024 ? 2 : 1);
  static_assert(NS != 1 || WG_PER_CU * LDS_BYTES <= 160 * 1024, "single-buffer form: LDS");
  static constexpr int WAVES_PER_SIMD = WG_PER_CU * NW / 4;   // __launch_bounds__ second argument
};

// one output tile; `bid` = tile index inside the (clip, group) batch entry `bz`
// wait until at most N of this wave's LDS-DMA instructions are still in flight, then a barrier that orders LDS only
template <int N>
SER_DEVFN void wait_dma_barrier() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int BM, int BN, int MODE, int NS = 2, int WR = 2, int WC = 2>
SER_DEVFN void gemm_tile(const SerGemmArgs& g, const int bid_in, const int bz, char* lds, const int ksl = 0) {
  constexpr bool X3 = MODE == M_X3P, IL = MODE == M_X3I;
  using Cfg = GemmCfg<BM, BN, MODE, NS, WR, WC>;
  constexpr int NW = Cfg::NW, NT = Cfg::NT;
  constexpr int NPL = Cfg::NPL;
  constexpr int A_TILE = Cfg::A_TILE, W_TILE = Cfg::W_TILE;
  constexpr int STAGE = Cfg::STAGE;
  constexpr int GLDS = Cfg::GLDS;
  constexpr int WM = BM / WR, WN = BN / WC, TM = WM / 16, TN = WN / 16;
  constexpr int PM = IL ? 2 : 1;                    // physical elements per logical element along K (interleaved: hi + lo)
  constexpr int KT = IL ? 32 : 64;                  // values of k per k-tile (a k-tile is always 128 B per row)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WC, wn = wave % WC;

  // blockIdx -> tile: (1) XCD-contiguous chunks (blocks b, b+8, ... share an XCD's L2, so give each XCD a
  // contiguous run of tiles), (2) inside a run, GROUP_M x tiles_n super-tiles so the ~32 workgroups resident on
  // one XCD touch few distinct A and W panels at a time.  Pure speed: any placement gives the same result.
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  int tile_m, tile_n;
  {
    const int nt = tiles_m * tiles_n, bid = bid_in;
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7, local = bid >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    const int GROUP_M = g.group_m > 0 ? g.group_m : 4;
    const int per_group = GROUP_M * tiles_n;
    const int first_m = (t / per_group) * GROUP_M;
    const int gsz = min(GROUP_M, tiles_m - first_m);
    tile_m = first_m + (t % per_group) % gsz;
    tile_n = (t % per_group) / gsz;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int b1 = bz / g.nb2, b2 = bz % g.nb2;

  const long long aoff = (b1 * g.sa1 + b2 * g.sa2) * PM, woff = (b1 * g.sw1 + b2 * g.sw2) * PM;
  const bf16_t* a_hi = g.a_hi + aoff;
  const bf16_t* w_hi = g.w_hi + woff;
  const bf16_t* a_lo = X3 ? g.a_lo + aoff : nullptr;
  const bf16_t* w_lo = X3 ? g.w_lo + woff : nullptr;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // split-K: slice ksl of g.ksplit takes k-tiles [k0, k0 + nk) and writes its partial sums to slab ksl of c_f32
  int nk = g.K / KT, k0 = 0;
  if (g.ksplit > 1) {
    const int per = (nk + g.ksplit - 1) / g.ksplit;
    k0 = ksl * per;
    nk = min(per, nk - k0);
  }
  unsigned offa[BM / (8 * NW)], offw[BN / (8 * NW)];
  stage_offsets<BM, NW>(offa, (long long)g.lda * PM, m0, g.M - 1, wave, lane);
  stage_offsets<BN, NW>(offw, (long long)g.ldw * PM, n0, g.N - 1, wave, lane);
  auto stage = [&](int kt, int buf) {
    if (SER_GEMM_DIAG & 4) return;
    char* s = lds + buf * STAGE;
    stage_tile<BM, NW>(a_hi + (k0 + kt) * BK, offa, s, wave);
    stage_tile<BN, NW>(w_hi + (k0 + kt) * BK, offw, s + A_TILE * NPL, wave);
    if (X3) {
      stage_tile<BM, NW>(a_lo + (k0 + kt) * BK, offa, s + A_TILE, wave);
      stage_tile<BN, NW>(w_lo + (k0 + kt) * BK, offw, s + A_TILE * NPL + W_TILE, wave);
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  if constexpr (NS == 1) {
    // ONE LDS buffer: the registers are the second buffer.  Per k-tile: wait for the tile, barrier, read EVERY fragment
    // of the tile into registers, barrier, refill the same buffer with the next k-tile, then the MFMAs run under the
    // DMA.  Half the LDS of the double-buffered form, so three workgroups share a CU and three k-tiles per CU are in
    // flight on the global->LDS path, whose round trip (not the MFMA rate) bounds the two-buffer loop.
    static_assert(IL, "single-buffer form: interleaved three-product mode only");
    if (nk > 0) stage(0, 0);
    const char* sa = lds + wm * WM * ROW_BYTES;
    const char* sw = lds + A_TILE + wn * WN * ROW_BYTES;
    for (int kt = 0; kt < nk; ++kt) {
      wait_dma_barrier<0>();
      bf16x8 bh[TN], bl[TN], ah[TM], al[TM];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = lds_frag(sw, j * 16 + fr, fq);
        bl[j] = lds_frag(sw, j * 16 + fr, 4 + fq);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = lds_frag(sa, i * 16 + fr, fq);
        al[i] = lds_frag(sa, i * 16 + fr, 4 + fq);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) stage(kt + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  } else {
  // NS LDS buffers, NS-1 k-tiles in flight.  Tile kt is retired by a COUNTED wait (the younger tiles stay in flight
  // across the barrier), the barrier makes every wave's part of it visible, and the buffer read in the previous
  // iteration is refilled right after the barrier.  With NS = 2 this is the classic double buffer (vmcnt(0)).
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nk) stage(s0, s0);
  int rbuf = 0, wbuf = NS - 1;
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;                     // tiles issued after kt (capped at NS-2 by construction)
    if (NS >= 4 && ahead >= 2) wait_dma_barrier<(NS >= 4 ? 2 : 0) * GLDS>();
    else if (NS >= 3 && ahead >= 1) wait_dma_barrier<(NS >= 3 ? 1 : 0) * GLDS>();
    else wait_dma_barrier<0>();
    const char* s = lds + rbuf * STAGE;
    rbuf = rbuf + 1 == NS ? 0 : rbuf + 1;
    const char* sa = s + wm * WM * ROW_BYTES;
    const char* sw = s + A_TILE * NPL + wn * WN * ROW_BYTES;
    if constexpr (IL) {
      // interleaved planes: chunks 0-3 of a row are the hi values of this k-tile's 32 k, chunks 4-7 their lo values.
      // One MFMA k-step per k-tile, three products per accumulator; the A fragments of row block i+1 are requested
      // before the MFMAs of row block i (pinned), so their LDS latency hides under 3*TN MFMAs.
      bf16x8 bh[TN], bl[TN], ah[TM], al[TM];
      constexpr bool RUN = !(SER_GEMM_DIAG & 2);
      if (RUN) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = lds_frag(sw, j * 16 + fr, fq);
          bl[j] = lds_frag(sw, j * 16 + fr, 4 + fq);
        }
        ah[0] = lds_frag(sa, fr, fq);
        al[0] = lds_frag(sa, fr, 4 + fq);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the whole next stage goes out in one burst, as early as possible: the loop is bound by the global->LDS round
      // trip (1.5-2 us under load), and issuing the pieces one group per MFMA row block instead measured 1-20 % slower
      if (kt + NS - 1 < nk) stage(kt + NS - 1, wbuf);
      __builtin_amdgcn_sched_barrier(0);
      if (RUN) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (i + 1 < TM) {
            ah[i + 1] = lds_frag(sa, (i + 1) * 16 + fr, fq);
            al[i + 1] = lds_frag(sa, (i + 1) * 16 + fr, 4 + fq);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      wbuf = wbuf + 1 == NS ? 0 : wbuf + 1;
    } else {
    // Fragment reads are software-pipelined by hand: the ds_read_b128s of k-step ks+1 are issued after the first MFMA
    // rows of k-step ks and pinned there (sched_barrier: nothing is scheduled across it), so they land under the
    // remaining rows (the summation order per accumulator is unchanged).  Left alone, the compiler keeps one
    // A fragment live at a time — read, s_waitcnt lgkmcnt(0), four MFMAs, read, ... — which exposes the full LDS latency
    // every 64 MFMA cycles (measured: ~2700 cycles per k-tile of a 192x128 tile against 768 cycles of MFMA issue).
    bf16x8 ah[2][TM], bh[2][TN], al[2][X3 ? TM : 1], bl[2][X3 ? TN : 1];
    auto read_frags = [&](int ks, int set) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[set][j] = lds_frag(sw, j * 16 + fr, ks * 4 + fq);
        if (X3) bl[set][j] = lds_frag(sw + W_TILE, j * 16 + fr, ks * 4 + fq);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[set][i] = lds_frag(sa, i * 16 + fr, ks * 4 + fq);
        if (X3) al[set][i] = lds_frag(sa + A_TILE, i * 16 + fr, ks * 4 + fq);
      }
    };
    constexpr int KS = (SER_GEMM_DIAG & 2) ? 0 : 2;
    if (KS) read_frags(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // the refill of the buffer read in the previous iteration is issued while the first fragments are on their way
    if (kt + NS - 1 < nk) stage(kt + NS - 1, wbuf);
    wbuf = wbuf + 1 == NS ? 0 : wbuf + 1;
    __builtin_amdgcn_sched_barrier(0);
    auto mfma_rows = [&](int cur, int i_lo, int i_hi) {
#pragma unroll
      for (int i = i_lo; i < i_hi; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (X3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[cur][i], bh[cur][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cur][i], bl[cur][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cur][i], bh[cur][j], acc[i][j], 0, 0, 0);
        }
    };
    constexpr int HEAD = TM >= 3 ? TM / 3 : 1;     // MFMA rows issued before the next k-step's reads go out
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cur = ks & 1;
      mfma_rows(cur, 0, HEAD);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < KS) read_frags(ks + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_rows(cur, HEAD, TM);
    }
    }
  }
  }   // NS > 1

  // epilogue: accumulators -> LDS (row-major fp32 rows) -> coalesced 16-byte global stores, in BM / EPR passes of EPR rows
  // (the classic 2 x 2 form: the rows of wave-row 0, then those of wave-row 1; the staging area is at most half a tile, so
  // tall tiles keep 2 workgroups per CU).  C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg.
  constexpr int LDT = BN + 4;                       // floats; +4 keeps the two half-waves on different banks
  constexpr int EPR = Cfg::EPR, IPP = EPR / 16, NPASS = BM / EPR;
  float* tile = (float*)lds;
  if (SER_GEMM_DIAG & 1) {                          // probe: keep the accumulators alive, store (almost) nothing
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sacc == 123456.789f && g.c_f32) g.c_f32[0] = sacc;
    return;
  }

  const bool partial = g.ksplit > 1;                // split-K slice: raw partial sums to its fp32 slab, nothing else
  const long long coff = b1 * g.sc1 + b2 * g.sc2 + (partial ? (long long)ksl * g.slab_stride : 0);
  const float* bias = (g.bias && !partial) ? g.bias + b1 * g.sbias1 + b2 * g.sbias2 : nullptr;
  const float* res = (g.residual && !partial) ? g.residual + b1 * g.sr1 + b2 * g.sr2 : nullptr;
  const int act = partial ? SER_ACT_NONE : g.act;
  const bool c_il = ser_is_il(g.c_hi, g.c_lo);      // interleaved output planes: one array, offsets mapped by ser_il_off
  constexpr int TPR = BN / 8;                        // threads per row, 8 columns each
  constexpr int RPI = NT / TPR;                      // rows per iteration
  const int tc = (tid % TPR) * 8, tr = tid / TPR;
  const int n = n0 + tc;
  const bool vec = (n + 7 < g.N) && ((g.ldc & 7) == 0) && ((coff & 7) == 0);
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = (bias && n + e < g.N) ? bias[n + e] : 0.f;
  static_for<0, NPASS>([&](auto pass_tag) {
  constexpr int pass = decltype(pass_tag)::value;
  constexpr int owner = (pass * EPR) / WM;           // wave-row whose accumulators hold these rows
  constexpr int i0 = ((pass * EPR) % WM) / 16;       // compile-time accumulator indices: acc stays in registers
  __syncthreads();                                   // k-loop reads (first pass) / previous pass's reads are done
  if (wm == owner) {
#pragma unroll
    for (int ii = 0; ii < IPP; ++ii)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(ii * 16 + fq * 4 + r) * LDT + wn * WN + j * 16 + fr] = acc[i0 + ii][j][r];
  }
  __syncthreads();
#pragma unroll 2
  for (int rr = tr; rr < EPR; rr += RPI) {
    const int m = m0 + pass * EPR + rr;
    if (m >= g.M || n >= g.N) continue;
    float v[8];
    const float4 t0 = *(const float4*)(tile + rr * LDT + tc), t1 = *(const float4*)(tile + rr * LDT + tc + 4);
    v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e] + bv[e], act);
    const long long o = coff + (long long)m * g.ldc + n;
    const long long op = c_il ? ser_il_off(o) : o;   // plane offset (8 consecutive columns never straddle a 32-group)
    if (SER_GEMM_DIAG & 8) {
      if (v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] == 123456.789f && g.c_hi) g.c_hi[o] = 1;
      continue;
    }
    if (vec) {
      if (res) {
        const float* rp = res + (long long)m * g.ldr + n;
        if ((((uintptr_t)rp) & 15) == 0) {
          const float4 r0 = *(const float4*)rp, r1 = *(const float4*)(rp + 4);
          v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rp[e];
        }
      }
      if (g.c_f32) {
        *(float4*)(g.c_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(g.c_f32 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
      if (g.c_hi && !partial) {
        uint32_t ph[4], pl[4];
        if (g.c_lo) {
#pragma unroll
          for (int e = 0; e < 4; ++e) split_bf16x2(v[2 * e], v[2 * e + 1], ph[e], pl[e]);
          *(uint4*)(g.c_lo + op) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) ph[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        }
        *(uint4*)(g.c_hi + op) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (n + e >= g.N) break;
        float x = v[e];
        if (res) x += res[(long long)m * g.ldr + n + e];
        if (g.c_f32) g.c_f32[o + e] = x;
        if (g.c_hi && !partial) {
          bf16_t h, l;
          split_bf16(x, h, l);
          const long long oe = c_il ? ser_il_off(o + e) : o + e;
          g.c_hi[oe] = h;
          if (g.c_lo) g.c_lo[oe] = l;
        }
      }
    }
  }
  });   // pass
}

// Persistent launch: the grid never exceeds what is resident at once (gridDim.x <= 2 workgroups per CU), each
// workgroup walks tiles `blockIdx.x, blockIdx.x + gridDim.x, ...` of the flattened (batch entry, tile) space.  No tile
// ever waits in the dispatcher's queue, so the small head kernels of the other stream are placed as soon as they
// arrive instead of behind this kernel's not-yet-dispatched workgroups.
template <int BM, int BN, int MODE, int NS = 2, int WR = 2, int WC = 2>
__global__ __launch_bounds__((GemmCfg<BM, BN, MODE, NS, WR, WC>::NT), (GemmCfg<BM, BN, MODE, NS, WR, WC>::WAVES_PER_SIMD))
void gemm_bf16_nt_kernel(const SerGemmArgs g, const int tiles, const int total) {
  __shared__ __attribute__((aligned(1024))) char lds[GemmCfg<BM, BN, MODE, NS, WR, WC>::LDS_BYTES];
  // flattened work space: (split-K slice, batch entry, tile); the slices of a tile are adjacent block ids
  const int ks = g.ksplit > 1 ? g.ksplit : 1;
  for (int w = blockIdx.x; w < total * ks; w += gridDim.x) {
    const int t = w / ks;
    gemm_tile<BM, BN, MODE, NS, WR, WC>(g, t % tiles, t / tiles, lds, w % ks);
    __syncthreads();     // the epilogue's LDS tile is dead before the next tile's first stage lands
  }
}

// Two independent problems in one launch (the layer-l GEMMs of Wav2Vec2 and of XLM-R have no dependence on each
// other): the tiles of the small problem come first in the grid, so they start at once and ride along with the
// large one instead of queueing, launch after launch, on a second stream behind it.
template <int BM, int BN, int MODE, int NS = 2, int WR = 2, int WC = 2>
__global__ __launch_bounds__((GemmCfg<BM, BN, MODE, NS, WR, WC>::NT), (GemmCfg<BM, BN, MODE, NS, WR, WC>::WAVES_PER_SIMD))
void gemm_bf16_pair_kernel(const SerGemmArgs g0, const SerGemmArgs g1, const int total0, const int tiles0, const int tiles1, const int total_all) {
  __shared__ __attribute__((aligned(1024))) char lds[GemmCfg<BM, BN, MODE, NS, WR, WC>::LDS_BYTES];
  // total0 / tiles0 / tiles1 count (tile, split-K slice) pairs when the problems are split (both with the same factor)
  const int ks = g0.ksplit > 1 ? g0.ksplit : 1;
  // grid-stride walk of the flattened (problem, batch entry, tile, slice) space: the grid may be smaller than the work
  // (occupancy headroom, see launch_kernel); with one workgroup per item the loop runs once
  for (int wi = blockIdx.x; wi < total_all; wi += gridDim.x) {
    // one call site: the problem is chosen by (uniform) address, not by duplicating the tile code in two branches
    const bool first = wi < total0;
    const SerGemmArgs* g = first ? &g0 : &g1;
    const int w = first ? wi : wi - total0, tiles = first ? tiles0 : tiles1;
    const int t = w / ks;
    gemm_tile<BM, BN, MODE, NS, WR, WC>(*g, t % tiles, t / tiles, lds, w % ks);
    __syncthreads();     // the epilogue's LDS tile is dead before the next tile's first stage lands
  }
}

// ---- optional per-launch timing with HIP events (bench.py roofline leg; off by default) ----------
struct ProfRec {
  hipEvent_t e0, e1;
  double flops;
};
static int g_gemm_persist_cap = 0;   // 0 = one workgroup per tile; N = persistent grid of at most N workgroups
extern "C" int ser_debug_set_gemm_persist(int cap) { g_gemm_persist_cap = cap; return 0; }
// Occupancy headroom.  An encoder GEMM normally claims every workgroup slot of the chip (2-4 per CU); the head of the
// previous batches runs beside it on other queues as ~150 small dependent kernels, and each of those then waits for a GEMM
// workgroup to retire before it can be placed (measured: ~11 us per kernel boundary instead of 1.7, head step 2.75 -> 5.0 ms
// - the head, not the encoders, became the critical path once one encoder pass covered several batches).  With
// g_gemm_occupancy_pct = P < 100 a GEMM launches at most P % of the slots its tile configuration could hold resident and
// walks its tiles grid-stride; the remaining slots stay free for the other queues for the whole launch.
static int g_gemm_occupancy_pct = 100;
extern "C" int ser_set_gemm_occupancy_pct(int pct) { g_gemm_occupancy_pct = pct < 10 ? 10 : (pct > 100 ? 100 : pct); return 0; }
extern "C" int ser_get_gemm_occupancy_pct(void) { return g_gemm_occupancy_pct; }
static int resident_cap(int wg_per_cu) {
  if (g_gemm_occupancy_pct >= 100) return 0;                         // no cap
  int cap = 256 * wg_per_cu * g_gemm_occupancy_pct / 100;
  cap -= cap % 8;                                                    // blocks b, b + 8, ... share an XCD: keeps tile t on XCD t % 8
  return cap < 8 ? 8 : cap;
}
// experiment knob: extra dynamic LDS per workgroup, i.e. fewer resident GEMM workgroups per CU
static int g_gemm_lds_pad = 0;
extern "C" int ser_debug_set_gemm_lds_pad(int bytes) { g_gemm_lds_pad = bytes; return 0; }
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

// LDS stages for the bf16 (single-product) kernels, per tile shape: experiment knob, see ser_launch_gemm_bf16
static int g_gemm_stages[7] = {2, 3, 2, 2, 2, 2, 2};   // [0] 128x128  [1] 64x128  [2] 64x64  [3] 128x64  [4] 96x128  [5] 160x128  [6] 192x128
// measured at BASELINE config 2 (ms/step): {2,2,2,2} 3.906, {2,3,2,2} 3.870, {2,4,2,2} 4.04, {3,3,2,2} 4.45 (96 KB of LDS
// per 128x128 workgroup leaves one per CU).  scripts/gemm_phase_probe.py: with two buffers a k-tile costs the DMA round
// trip (~0.75 us) whatever the MFMA phase takes, so the tall one-round tiles are latency-bound, not LDS- or MFMA-bound.
extern "C" int ser_debug_set_gemm_stages(int s128, int s64x128, int s64, int s128x64) {
  g_gemm_stages[0] = s128; g_gemm_stages[1] = s64x128; g_gemm_stages[2] = s64; g_gemm_stages[3] = s128x64;
  return 0;
}
extern "C" int ser_debug_set_gemm_stages_tall(int s96, int s160, int s192) {
  g_gemm_stages[4] = s96; g_gemm_stages[5] = s160; g_gemm_stages[6] = s192;
  return 0;
}
template <int BM, int BN>
static int stages_for() {
  const int idx = BM == 128 ? (BN == 128 ? 0 : 3) : BM == 64 ? (BN == 128 ? 1 : 2) : BM == 96 ? 4 : BM == 160 ? 5 : 6;
  const int v = g_gemm_stages[idx];
  return v < 2 ? 2 : (v > 4 ? 4 : v);
}

// operand mode of a problem (host side): no lo planes = one product; lo == hi + 32 elements on BOTH operands =
// interleaved three-product; otherwise planar three-product
static int gemm_mode(const SerGemmArgs& g) {
  if (!g.a_lo || !g.w_lo) return M_BF16;
  return (ser_is_il(g.a_hi, g.a_lo) && ser_is_il(g.w_hi, g.w_lo)) ? M_X3I : M_X3P;
}

struct ProfScope {       // HIP events around one launch when bench.py's roofline leg is recording
  ProfRec rec;
  bool on;
  hipStream_t st;
  int begin(double flops, hipStream_t s) {
    on = g_prof_on; st = s;
    if (!on) return SER_OK;
    SER_CHECK_HIP(hipEventCreate(&rec.e0));
    SER_CHECK_HIP(hipEventCreate(&rec.e1));
    rec.flops = flops;
    SER_CHECK_HIP(hipEventRecord(rec.e0, st));
    return SER_OK;
  }
  int end() {
    if (!on) return SER_OK;
    SER_CHECK_HIP(hipEventRecord(rec.e1, st));
    g_prof.push_back(rec);
    return SER_OK;
  }
};
static double gemm_flops(const SerGemmArgs& g) { return 2.0 * g.M * (double)g.N * g.K * g.nb1 * g.nb2; }   // algorithmic

// one problem, or two (small first) in the paired kernel; WR x WC waves
template <int BM, int BN, int MODE, int NS, int WR, int WC>
static int launch_kernel(const SerGemmArgs* small, const SerGemmArgs& big, hipStream_t st) {
  using Cfg = GemmCfg<BM, BN, MODE, NS, WR, WC>;
  const int ks = big.ksplit > 1 ? big.ksplit : 1;
  const int tiles1 = ceil_div(big.M, BM) * ceil_div(big.N, BN), total1 = tiles1 * big.nb1 * big.nb2;
  ProfScope ps;
  if (small) {
    const int tiles0 = ceil_div(small->M, BM) * ceil_div(small->N, BN), total0 = tiles0 * small->nb1 * small->nb2;
    SER_TRY(ps.begin(gemm_flops(*small) + gemm_flops(big), st));
    const int all = (total0 + total1) * ks, rc = resident_cap(Cfg::WG_PER_CU);
    hipLaunchKernelGGL((gemm_bf16_pair_kernel<BM, BN, MODE, NS, WR, WC>), dim3(rc > 0 && rc < all ? rc : all), dim3(Cfg::NT), g_gemm_lds_pad, st,
                       *small, big, total0 * ks, tiles0, tiles1, all);
  } else {
    const int rc = resident_cap(Cfg::WG_PER_CU);
    const int cap = g_gemm_persist_cap > 0 ? g_gemm_persist_cap : (rc > 0 ? rc : total1 * ks);
    SER_TRY(ps.begin(gemm_flops(big), st));
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM, BN, MODE, NS, WR, WC>), dim3(total1 * ks < cap ? total1 * ks : cap), dim3(Cfg::NT),
                       g_gemm_lds_pad, st, big, tiles1, total1);
  }
  SER_TRY(ps.end());
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// classic 2 x 2 wave grid, LDS stages from the per-shape table
template <int BM, int BN, int MODE>
static int launch_stages(const SerGemmArgs* small, const SerGemmArgs& big, hipStream_t st) {
  if constexpr (MODE == M_X3P) {
    return launch_kernel<BM, BN, MODE, 2, 2, 2>(small, big, st);
  } else {
    const int ns = stages_for<BM, BN>();
    if (ns == 3) return launch_kernel<BM, BN, MODE, 3, 2, 2>(small, big, st);
    if (ns == 4 && BM * BN <= 64 * 128) return launch_kernel<BM, BN, MODE, (BM * BN <= 64 * 128 ? 4 : 2), 2, 2>(small, big, st);
    return launch_kernel<BM, BN, MODE, 2, 2, 2>(small, big, st);
  }
}

template <int BM, int BN>
static int launch_any(const SerGemmArgs* small, const SerGemmArgs& big, hipStream_t st) {
  const int mode = gemm_mode(big);
  if (mode == M_X3P) {
    constexpr int XBM = (BM == 64 || BM == 128) ? BM : 128;     // the planar form (two LDS images per operand) keeps the two classic heights
    SER_REQUIRE(XBM == BM, "gemm_bf16: tile height %d is not built for the planar 3-product mode", BM);
    return launch_stages<XBM, BN, M_X3P>(small, big, st);
  }
  if (mode == M_X3I) return launch_stages<BM, BN, M_X3I>(small, big, st);
  return launch_stages<BM, BN, M_BF16>(small, big, st);
}
template <int BM, int BN>
int launch_cfg(const SerGemmArgs& g, hipStream_t st) { return launch_any<BM, BN>(nullptr, g, st); }
template <int BM, int BN>
int launch_pair_cfg(const SerGemmArgs& small, const SerGemmArgs& big, hipStream_t st) { return launch_any<BM, BN>(&small, big, st); }

// 512-thread tiles (interleaved three-product mode): 2 x 4 waves over BM x 256, 4 x 2 over 256 x 128; two LDS buffers
static int launch_wide(int cfg, const SerGemmArgs* small, const SerGemmArgs& big, hipStream_t st) {
  SER_REQUIRE(gemm_mode(big) == M_X3I, "gemm_bf16: the 512-thread tiles are built for the interleaved three-product mode");
  switch (cfg) {
    case SER_GEMM_CFG_128x256: return launch_kernel<128, 256, M_X3I, 2, 2, 4>(small, big, st);
    case SER_GEMM_CFG_192x256: return launch_kernel<192, 256, M_X3I, 2, 2, 4>(small, big, st);
    case SER_GEMM_CFG_256x256: return launch_kernel<256, 256, M_X3I, 2, 2, 4>(small, big, st);
    case SER_GEMM_CFG_256x128: return launch_kernel<256, 128, M_X3I, 2, 4, 2>(small, big, st);
    case SER_GEMM_CFG_128x256_3: return launch_kernel<128, 256, M_X3I, 3, 2, 4>(small, big, st);    // three LDS buffers: two k-tiles in flight
    case SER_GEMM_CFG_256x128_3: return launch_kernel<256, 128, M_X3I, 3, 4, 2>(small, big, st);
    case SER_GEMM_CFG_256x256_4W: return launch_kernel<256, 256, M_X3I, 2, 2, 2>(small, big, st);   // four waves, 128 x 128 each, 512 registers per lane
    default: break;
  }
  ser_set_error("gemm_bf16: unknown tile configuration %d", cfg);
  return SER_E_ARG;
}

static int gemm_check(const SerGemmArgs& g) {
  SER_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "gemm_bf16: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
  SER_REQUIRE(g.a_hi && g.w_hi, "gemm_bf16: null operand");
  SER_REQUIRE(g.nb1 >= 1 && g.nb2 >= 1, "gemm_bf16: bad batch");
  if (g.a_lo && g.w_lo) SER_REQUIRE(ser_is_il(g.a_hi, g.a_lo) == ser_is_il(g.w_hi, g.w_lo), "gemm_bf16: A and W must use the same plane layout (planar or interleaved)");
  if (gemm_mode(g) == M_X3I) {
    const int G = SER_IL_GROUP;
    SER_REQUIRE(g.K % G == 0 && g.lda % G == 0 && g.ldw % G == 0, "gemm_bf16 (interleaved): K=%d lda=%d ldw=%d must be multiples of %d", g.K, g.lda, g.ldw, G);
    SER_REQUIRE(g.sa1 % G == 0 && g.sa2 % G == 0 && g.sw1 % G == 0 && g.sw2 % G == 0, "gemm_bf16 (interleaved): batch strides must be multiples of %d", G);
  } else {
    SER_REQUIRE(g.K % BK == 0, "gemm_bf16: K=%d must be a multiple of %d", g.K, BK);
    SER_REQUIRE(g.lda % 8 == 0 && g.ldw % 8 == 0, "gemm_bf16: lda=%d ldw=%d must be multiples of 8", g.lda, g.ldw);
  }
  if (g.ksplit > 1) {
    SER_REQUIRE(gemm_mode(g) == M_X3I && g.c_f32 && g.slab_stride > 0, "gemm_bf16: split-K needs interleaved operands and an fp32 slab area");
    SER_REQUIRE(g.ksplit <= 8 && (g.K / 32) >= g.ksplit, "gemm_bf16: bad split-K factor %d for K=%d", g.ksplit, g.K);
  }
  if (g.c_hi && ser_is_il(g.c_hi, g.c_lo))
    SER_REQUIRE(g.ldc % SER_IL_GROUP == 0 && g.sc1 % SER_IL_GROUP == 0 && g.sc2 % SER_IL_GROUP == 0, "gemm_bf16: interleaved output needs ldc=%d and batch strides in multiples of %d", g.ldc, SER_IL_GROUP);
  return SER_OK;
}

}  // namespace

// Tile choice for BN = 128 (one-product and interleaved three-product modes share LDS footprint and tile set).  At these
// sizes (a few hundred tiles on 256 CUs) the time of a GEMM is
// rounds x (time of one tile), rounds = ceil(tiles / resident workgroups): 522 tiles of 128 rows need two rounds of
// 512 slots, 432 tiles of 160 rows need one.  Candidates: 64, 96, 128, 160, 192 rows; the cost of a tile grows with
// its rows plus a constant (W panel, prologue, epilogue).  g_gemm_force_bm overrides (experiments / tests).
static int g_gemm_force_bm = 0;
extern "C" int ser_debug_set_gemm_bm(int bm) { g_gemm_force_bm = bm; return 0; }
static int g_gemm_group_m = 0;       // 0 = rule below; experiments: m-tiles per super-tile of the tile order
extern "C" int ser_debug_set_gemm_group_m(int v) { g_gemm_group_m = v; return 0; }

template <int BM>
static int slots_per_cu() {
  const int lds = GemmCfg<BM, 128, M_BF16, (BM == 64 ? 3 : 2)>::LDS_BYTES;
  const int by_lds = (160 * 1024) / lds;
  return by_lds < 1 ? 1 : (by_lds > 4 ? 4 : by_lds);
}
// measured choices (filled by the engines' one-time tuning pass) take precedence over the model.  A plan = tile
// configuration (SER_GEMM_CFG_*: a tile height 64..192 of the 2 x 2 form, or a 512-thread tile) + split-K factor.
struct TileHint { long long rows; int N, K, mode, cfg, ksplit; };
static std::vector<TileHint> g_tile_hints;
extern "C" int ser_gemm_plan_set(long long rows_total, int N, int K, int three_products, int cfg, int ksplit) {
  const int mode = three_products ? M_X3I : M_BF16;
  for (auto& h : g_tile_hints)
    if (h.rows == rows_total && h.N == N && h.K == K && h.mode == mode) { h.cfg = cfg; h.ksplit = ksplit; return SER_OK; }
  g_tile_hints.push_back(TileHint{rows_total, N, K, mode, cfg, ksplit < 1 ? 1 : ksplit});
  return SER_OK;
}
extern "C" int ser_gemm_plan_get(long long rows_total, int N, int K, int three_products, int* cfg, int* ksplit) {
  const int mode = three_products ? M_X3I : M_BF16;
  if (cfg) *cfg = 0;
  if (ksplit) *ksplit = 1;
  for (const auto& h : g_tile_hints)
    if (h.rows == rows_total && h.N == N && h.K == K && h.mode == mode) {
      if (cfg) *cfg = h.cfg;
      if (ksplit) *ksplit = h.ksplit < 1 ? 1 : h.ksplit;
      return SER_OK;
    }
  return SER_OK;
}
extern "C" int ser_gemm_tile_hint_mode(long long rows_total, int N, int K, int three_products, int bm) {
  return ser_gemm_plan_set(rows_total, N, K, three_products, bm, 1);
}
extern "C" int ser_gemm_tile_hint(long long rows_total, int N, int K, int bm) { return ser_gemm_plan_set(rows_total, N, K, 0, bm, 1); }

static int pick_bm(long long rows_a, long long rows_b, int N, int K, long long nb_b, int mode) {
  if (g_gemm_force_bm) return g_gemm_force_bm;
  for (const auto& h : g_tile_hints)
    if (h.rows == rows_a + rows_b * nb_b && h.N == N && h.K == K && h.mode == mode && h.cfg) return h.cfg;
  // Interleaved three-product mode with (nearly) a full round of 128 x 128 tiles or more (a grouped encoder pass: 7 392+ rows):
  // the single-LDS-buffer 128 x 128 form, three workgroups per CU.  Stand-alone it is the fastest or within a few percent of
  // it on every shape of the pass (scripts/gemm_cfg_probe.py, profiles/r03_*), and beside the head it is the better citizen
  // (34 KB of LDS per workgroup, 58 KB of every CU left to the other queues): 0.129 of peak in situ against 0.119-0.122 for
  // plans picked by stand-alone timings.  Deterministic: every rank and every run launches the same kernels.
  if (mode == M_X3I) {
    const long long t128 = ((rows_a + 127) / 128 + ((rows_b + 127) / 128) * nb_b) * ((N + 127) / 128);
    if (t128 >= 640) return SER_GEMM_CFG_SINGLE + 128;
  }
  const int cands[5] = {64, 96, 128, 160, 192};
  const int slots[5] = {slots_per_cu<64>(), slots_per_cu<96>(), slots_per_cu<128>(), slots_per_cu<160>(), slots_per_cu<192>()};
  int best = 128;
  double best_cost = 1e30;
  for (int k = 0; k < 5; ++k) {
    const int bm = cands[k];
    const long long tiles = ((rows_a + bm - 1) / bm + ((rows_b + bm - 1) / bm) * nb_b) * ((N + 127) / 128);
    const long long cap = 256LL * slots[k];
    const long long rounds = (tiles + cap - 1) / cap;
    const long long per_round = (tiles + rounds - 1) / rounds;
    const long long per_cu = (per_round + 255) / 256;                     // resident tiles on the busiest CU
    // fitted to scripts/gemm_bm_sweep.py: a tile costs (rows + ~100), a second tile on the same CU adds ~10 %,
    // and a trailing partial round costs about 70 % of a full one
    const double cost = (1.0 + 0.7 * (double)(rounds - 1)) * (0.9 + 0.1 * (double)per_cu) * (bm + 100.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = bm; }
  }
  return best;
}

template <int BM>
static int launch_bm(const SerGemmArgs* small, const SerGemmArgs& big, hipStream_t st) {
  return small ? launch_pair_cfg<BM, 128>(*small, big, st) : launch_cfg<BM, 128>(big, st);
}
// single-LDS-buffer form (fragments of a whole k-tile in registers, three workgroups per CU): 256-thread BM x 128 tiles
static int launch_single(int cfg, const SerGemmArgs* small, const SerGemmArgs& big, hipStream_t st) {
  SER_REQUIRE(gemm_mode(big) == M_X3I, "gemm_bf16: the single-buffer tiles are built for the interleaved three-product mode");
  switch (cfg - SER_GEMM_CFG_SINGLE) {
    case 64: return launch_kernel<64, 128, M_X3I, 1, 2, 2>(small, big, st);
    case 96: return launch_kernel<96, 128, M_X3I, 1, 2, 2>(small, big, st);
    case 128: return launch_kernel<128, 128, M_X3I, 1, 2, 2>(small, big, st);
    default: break;      // taller tiles do not fit 168 registers per lane with every fragment of a k-tile resident
  }
  ser_set_error("gemm_bf16: unknown tile configuration %d", cfg);
  return SER_E_ARG;
}

// 64-column tiles (interleaved three-product mode): twice the workgroups of the 128-column form for the N = 768 shapes,
// whose 128-column tilings leave a quarter of the CUs with one workgroup or none
static int launch_narrow(int cfg, const SerGemmArgs* small, const SerGemmArgs& big, hipStream_t st) {
  SER_REQUIRE(gemm_mode(big) == M_X3I, "gemm_bf16: the 64-column tiles of the plan table are built for the interleaved three-product mode");
  switch (cfg - SER_GEMM_CFG_NARROW) {
    case 64: return launch_kernel<64, 64, M_X3I, 3, 2, 2>(small, big, st);
    case 96: return launch_kernel<96, 64, M_X3I, 3, 2, 2>(small, big, st);
    case 128: return launch_kernel<128, 64, M_X3I, 2, 2, 2>(small, big, st);
    case 192: return launch_kernel<192, 64, M_X3I, 2, 2, 2>(small, big, st);
    default: break;
  }
  ser_set_error("gemm_bf16: unknown tile configuration %d", cfg);
  return SER_E_ARG;
}

static int launch_bn128(const SerGemmArgs* small_in, const SerGemmArgs& big_in, hipStream_t st) {
  SerGemmArgs big = big_in, small_copy;
  const SerGemmArgs* small = small_in;
  if (g_gemm_group_m > 0) {
    big.group_m = g_gemm_group_m;
    if (small_in) { small_copy = *small_in; small_copy.group_m = g_gemm_group_m; small = &small_copy; }
  }
  const long long rows_a = small ? (long long)small->M * small->nb1 * small->nb2 : 0;
  const int cfg = big.cfg ? big.cfg : pick_bm(rows_a, big.M, big.N, big.K, (long long)big.nb1 * big.nb2, gemm_mode(big));
  if (cfg >= SER_GEMM_CFG_NARROW && cfg < SER_GEMM_CFG_NARROW + 1000) return launch_narrow(cfg, small, big, st);
  if (cfg >= SER_GEMM_CFG_SINGLE && cfg < SER_GEMM_CFG_SINGLE + 1000) return launch_single(cfg, small, big, st);
  if (cfg >= SER_GEMM_CFG_WIDE) return launch_wide(cfg, small, big, st);
  switch (cfg) {
    case 64: return launch_bm<64>(small, big, st);
    case 96: return launch_bm<96>(small, big, st);
    case 160: return launch_bm<160>(small, big, st);
    case 192: return launch_bm<192>(small, big, st);
    default: return launch_bm<128>(small, big, st);
  }
}

// `big` decides the tile shape; both problems must be in the same precision mode.
int ser_launch_gemm_bf16_pair(const SerGemmArgs& small, const SerGemmArgs& big, hipStream_t st) {
  SER_TRY(gemm_check(small));
  SER_TRY(gemm_check(big));
  SER_REQUIRE(gemm_mode(small) == gemm_mode(big), "gemm_bf16 pair: mixed precision modes");
  SER_REQUIRE((small.ksplit > 1 ? small.ksplit : 1) == (big.ksplit > 1 ? big.ksplit : 1), "gemm_bf16 pair: the two problems must use the same split-K factor");
  const long long nb = (long long)big.nb1 * big.nb2;
  const long long t128 = (long long)ceil_div(big.M, 128) * ceil_div(big.N, 128) * nb;
  if (big.N <= 64 || small.N <= 64) {          // keep the narrow-N shape out of the pair path
    SER_TRY(ser_launch_gemm_bf16(small, st));
    return ser_launch_gemm_bf16(big, st);
  }
  const bool planar = gemm_mode(big) == M_X3P;
  if (!planar && big.M > 64 && big.N >= 128) return launch_bn128(&small, big, st);
  if (t128 >= 384 && big.M > 64) return launch_pair_cfg<128, 128>(small, big, st);
  const long long t64 = (long long)ceil_div(big.M, 64) * ceil_div(big.N, 128) * nb;
  if (t64 >= 256 || big.M <= 64) return launch_pair_cfg<64, 128>(small, big, st);
  return launch_pair_cfg<64, 64>(small, big, st);
}

int ser_launch_gemm_bf16(const SerGemmArgs& g, hipStream_t st) {
  SER_TRY(gemm_check(g));
  // tile choice: fill >= 256 CUs when the problem allows it
  const long long nb = (long long)g.nb1 * g.nb2;
  const long long t128 = (long long)ceil_div(g.M, 128) * ceil_div(g.N, 128) * nb;
  if (g.N <= 64) return launch_cfg<128, 64>(g, st);
  const bool planar = gemm_mode(g) == M_X3P;
  if (!planar && g.M > 64 && g.N >= 128 && t128 * 4 >= 256) return launch_bn128(nullptr, g, st);
  if (t128 >= 384 || g.M <= 64) {
    if (g.M <= 64) return launch_cfg<64, 128>(g, st);
    return launch_cfg<128, 128>(g, st);
  }
  const long long t64 = (long long)ceil_div(g.M, 64) * ceil_div(g.N, 128) * nb;
  if (t64 >= 256) return launch_cfg<64, 128>(g, st);
  return launch_cfg<64, 64>(g, st);
}

extern "C" int ser_gemm_bf16_nt(const uint16_t* a_hi, const uint16_t* a_lo, int lda, const uint16_t* w_hi,
                                const uint16_t* w_lo, int ldw, int M, int N, int K, const float* bias, int act,
                                const float* residual, int ldr, float* c_f32, uint16_t* c_hi, uint16_t* c_lo,
                                int ldc, void* stream) {
  SerGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_hi = a_hi; g.a_lo = a_lo; g.w_hi = w_hi; g.w_lo = w_lo;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw;
  g.nb1 = 1; g.nb2 = 1;
  g.bias = bias; g.act = act; g.residual = residual; g.ldr = ldr;
  g.c_f32 = c_f32; g.c_hi = c_hi; g.c_lo = c_lo; g.ldc = ldc;
  if ((a_lo == nullptr) != (w_lo == nullptr)) { g.a_lo = nullptr; g.w_lo = nullptr; }
  return ser_launch_gemm_bf16(g, (hipStream_t)stream);
}

// debug / test entry: two plain bf16 NT GEMMs (fp32 outputs) through the paired launch
extern "C" int ser_debug_gemm_pair(const uint16_t* a0, const uint16_t* w0, int M0, int N0, int K0, float* c0,
                                   const uint16_t* a1, const uint16_t* w1, int M1, int N1, int K1, float* c1, void* stream) {
  SerGemmArgs g[2];
  memset(g, 0, sizeof(g));
  g[0].a_hi = a0; g[0].w_hi = w0; g[0].M = M0; g[0].N = N0; g[0].K = K0; g[0].lda = K0; g[0].ldw = K0;
  g[0].nb1 = g[0].nb2 = 1; g[0].c_f32 = c0; g[0].ldc = N0;
  g[1].a_hi = a1; g[1].w_hi = w1; g[1].M = M1; g[1].N = N1; g[1].K = K1; g[1].lda = K1; g[1].ldw = K1;
  g[1].nb1 = g[1].nb2 = 1; g[1].c_f32 = c1; g[1].ldc = N1;
  return ser_launch_gemm_bf16_pair(g[0], g[1], (hipStream_t)stream);
}

// debug / probe entry: batched NT GEMM with explicit batch strides (in logical elements; 0 = every batch entry reads the
// same operand, i.e. an L2-resident working set of any tile count), interleaved three-product planes when il != 0
extern "C" int ser_debug_gemm_batched(const uint16_t* a, const uint16_t* w, int M, int N, int K, int nb, long long sa,
                                      long long sw, uint16_t* c, long long sc, int il, void* stream) {
  SerGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_hi = a; g.w_hi = w;
  if (il) { g.a_lo = a + SER_IL_GROUP; g.w_lo = w + SER_IL_GROUP; }
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K;
  g.nb1 = nb; g.nb2 = 1; g.sa1 = sa; g.sw1 = sw; g.sc1 = sc;
  g.c_hi = c; g.c_lo = il ? c + SER_IL_GROUP : nullptr; g.ldc = N;
  return ser_launch_gemm_bf16(g, (hipStream_t)stream);
}

// NT product in the interleaved three-product mode with the K range cut into `ksplit` slices (<= 8): slabs = [ksplit][M][N] raw
// partial sums, to be added by the caller in slice order (ser_colsum over [ksplit, M * N]).  For products whose K is long and whose
// output is small - the weight gradients of the fine-tuning encoders (K = tokens of the batch, or conv frames): a handful of output
// tiles would otherwise walk the whole K range on a handful of CUs.
extern "C" int ser_gemm_bf16_nt_splitk(const uint16_t* a_hi, const uint16_t* a_lo, int lda, const uint16_t* w_hi, const uint16_t* w_lo,
                                       int ldw, int M, int N, int K, int ksplit, float* slabs, void* stream) {
  SerGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_hi = a_hi; g.a_lo = a_lo; g.w_hi = w_hi; g.w_lo = w_lo;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw;
  g.nb1 = 1; g.nb2 = 1;
  g.c_f32 = slabs; g.ldc = N;
  g.ksplit = ksplit; g.slab_stride = (long long)M * N;
  return ser_launch_gemm_bf16(g, (hipStream_t)stream);
}

// debug / probe entry: interleaved three-product GEMM with an explicit tile configuration and split-K factor;
// c_f32 = [ksplit][M][N] slabs of raw partial sums when ksplit > 1, else the [M][N] result
extern "C" int ser_debug_gemm_il_cfg(const uint16_t* a, const uint16_t* w, int M, int N, int K, int cfg, int ksplit, float* c_f32,
                                     void* stream) {
  SerGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_hi = a; g.w_hi = w; g.a_lo = a + SER_IL_GROUP; g.w_lo = w + SER_IL_GROUP;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K;
  g.nb1 = 1; g.nb2 = 1;
  g.c_f32 = c_f32; g.ldc = N;
  g.cfg = cfg; g.ksplit = ksplit; g.slab_stride = (long long)M * N;
  return ser_launch_gemm_bf16(g, (hipStream_t)stream);
}

// Per-launch HIP-event timing of the encoder GEMM kernel.  start: begin recording; stop: synchronise
// on the recorded events and return total kernel milliseconds, algorithmic FLOPs and launch count.
extern "C" int ser_prof_gemm_start(void) {
  g_prof.clear();
  g_prof_on = true;
  return SER_OK;
}
extern "C" int ser_prof_gemm_stop(double* total_ms, double* total_flops, long long* launches) {
  g_prof_on = false;
  double ms = 0.0, fl = 0.0;
  for (auto& r : g_prof) {
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
  if (launches) *launches = (long long)g_prof.size();
  g_prof.clear();
  return SER_OK;
}
