// Persistent multi-workgroup kernels: a handful of resident workgroups walk a chain of small dependent
// steps and hand activations to each other through device memory, instead of one launch per step.
//
// Exchange protocol (no cache-wide fences): data that crosses workgroups is written with agent-scope
// relaxed atomic stores (write-through, `sc1`) and read with agent-scope relaxed atomic loads, so it
// never lives in a non-coherent per-XCD L2 line; arrival is one epoch word per workgroup, written after
// `s_waitcnt vmcnt(0)` + workgroup barrier, and polled by one wave of every workgroup.
// Every wait is bounded: after SPIN_LIMIT polls a workgroup raises `abort` and all loops fall through,
// so the grid always drains.
#include "ser_common.h"

namespace {

constexpr unsigned SPIN_LIMIT = 1u << 22;

struct GridBar {
  unsigned* flags;   // [G] epoch reached by each workgroup (zeroed before the launch)
  unsigned* abort;   // [1]
  int G;
};

SER_DEVFN void st_coh(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SER_DEVFN float ld_coh(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SER_DEVFN float2 ld_coh2(const float* p) {
  const unsigned long long u =
      __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
}

// all workgroups reach epoch `e`.  Returns false when the wait was abandoned.
SER_DEVFN bool grid_arrive_wait(const GridBar& gb, unsigned e) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  __shared__ int ok_sh;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (lane == 0) __hip_atomic_store(gb.flags + blockIdx.x, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    bool ok = true;
    for (;;) {
      bool mine = true;
      for (int i = lane; i < gb.G; i += 64)
        mine = mine && (__hip_atomic_load(gb.flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= e);
      if (__all(mine)) break;
      if (++spins > SPIN_LIMIT || __hip_atomic_load(gb.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        if (lane == 0) __hip_atomic_store(gb.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) ok_sh = ok ? 1 : 0;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return ok_sh != 0;
}

// the conservative variant: plain accesses + agent-scope fences (cache write-back / invalidate)
SER_DEVFN bool grid_arrive_wait_fenced(const GridBar& gb, unsigned e) {
  __syncthreads();
  __shared__ int ok_sh2;
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_store(gb.flags + blockIdx.x, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    bool ok = true;
    for (int i = 0; i < gb.G && ok; ++i)
      while (__hip_atomic_load(gb.flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < e) {
        if (++spins > SPIN_LIMIT) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    __threadfence();
    ok_sh2 = ok ? 1 : 0;
  }
  __syncthreads();
  return ok_sh2 != 0;
}

// Probe: `rounds` exchanges of a [16][16 G] fp32 panel; workgroup b writes columns [16 b, 16 b + 16) and then
// every workgroup reads the whole panel and checks it.  mode 0 = fenced, 1 = coherent accesses.
__global__ __launch_bounds__(512) void barrier_probe_kernel(GridBar gb, float* data /*[2][16][16G]*/, int rounds, int mode,
                                                            unsigned* errors) {
  const int G = gb.G, N = 16 * G, tid = threadIdx.x;
  unsigned bad = 0;
  for (int r = 1; r <= rounds; ++r) {
    float* buf = data + (size_t)(r & 1) * 16 * N;
    if (tid < 256) {
      const int m = tid >> 4, c = tid & 15;
      const float v = (float)(r * 64 + (int)blockIdx.x);
      if (mode) st_coh(buf + m * N + blockIdx.x * 16 + c, v);
      else buf[m * N + blockIdx.x * 16 + c] = v;
    }
    const bool ok = mode ? grid_arrive_wait(gb, (unsigned)r) : grid_arrive_wait_fenced(gb, (unsigned)r);
    if (!ok) break;
    for (int i = tid * 2; i < 16 * N; i += 1024) {
      float2 v;
      if (mode) v = ld_coh2(buf + i);
      else v = *(const float2*)(buf + i);
      const int col = i % N;
      const float want0 = (float)(r * 64 + col / 16), want1 = (float)(r * 64 + (col + 1) / 16);
      bad += (v.x != want0) + (v.y != want1);
    }
  }
  if (bad) atomicAdd(errors, bad);
}


// ------------------------------------------------------------------------------------------
// The residual stack of the deep classifier (ref classifier.py:77-89 DeepResidualBlock, :209-216 the loop):
//   x1 = LN(h; g1,b1)   u = LN(x1; g2,b2)   a = relu(u W1^T + c1)   h' = x1 + a W2^T + c2
// at M <= 16 rows.  One launch walks all L blocks: workgroup b owns output columns [16 b, 16 b + 16) of every
// Linear (its weight slice is fetched one phase ahead), and the M x D activations travel between workgroups as
// (value, tag) pairs: every element is ONE 8-byte coherent store whose upper half carries the number of the
// phase that produced it, and a consumer simply re-reads the elements it needs until their tags match.  There is
// no flag, no fence and no barrier on the path: a hand-off costs one store and one load latency.  The exchange
// area is a two-slot ring (a slot is rewritten two phases later, which the data dependencies already order);
// tags are unique per launch (launch counter in the scratch area), so nothing is reset between launches.
// Dense copies of everything backward needs are written on the side with ordinary stores.
// Arithmetic and summation order are those of skinny_fwd_ln2_kernel / skinny_fwd_kernel /
// skinny_dgrad16_kernel / ln2_bwd_kernel.
// ------------------------------------------------------------------------------------------
struct StackBlockPtrs {
  const float *g1, *b1, *g2, *b2, *W1, *c1, *W2, *c2;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long ull;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int SW = 8;            // waves per workgroup
constexpr int PPAD = 4;          // LDS panel row padding (floats): rows land on different banks
constexpr unsigned LL_SPIN_LIMIT = 1u << 20;

// optional phase timestamps of workgroup 0 (scripts/stack_timeline.py): [L][8] wall-clock ticks (10 ns)
__device__ unsigned long long* g_stack_dbg = nullptr;
#define STACK_MARK(k)                                                                       \
  do {                                                                                      \
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[blk * 8 + (k)] = wall_clock64();   \
  } while (0)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and
// store of the wave (s_waitcnt vmcnt(0)), i.e. the weight prefetch and the dense side copies would sit on the
// critical path of every phase; nothing in these kernels communicates through global memory inside a workgroup.
SER_DEVFN void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct StackScratch {            // device scratch of one direction: 256-byte header + ring[2][16][D] of (value, tag)
  unsigned* launch;              // [0] launches so far
  unsigned* abort;               // [1] sticky: a wait was abandoned
  ull* ring;
};
SER_DEVFN StackScratch stack_scratch(void* p) {
  return StackScratch{(unsigned*)p, (unsigned*)p + 1, (ull*)((char*)p + 256)};
}

SER_DEVFN void st_ll(ull* p, float v, unsigned tag) {
  __hip_atomic_store(p, ((ull)tag << 32) | (ull)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Copy the first M rows of a ring slot into the LDS panel [16][D + PPAD], waiting for tag `want` on every element.
// 512 threads; returns false (for the whole workgroup) when the wait was abandoned.
SER_DEVFN bool ll_fetch_panel(const ull* slot, float* panel, int M, int D, unsigned want, unsigned* abort_flag) {
  __shared__ int ll_ok;
  if (threadIdx.x == 0) ll_ok = 1;
  lds_barrier();
  bool ok = true;
  for (int col = threadIdx.x; col < D && ok; col += SW * 64) {
    unsigned spins = 0;
    for (;;) {
      ull u[16];                                         // all (<= 16) rows of this column in one round trip
#pragma unroll
      for (int k = 0; k < 16; ++k)
        u[k] = __hip_atomic_load(slot + min(k, M - 1) * D + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool good = true;
#pragma unroll
      for (int k = 0; k < 16; ++k) good = good && ((unsigned)(u[k] >> 32) == want);
      if (good) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (k < M) panel[k * (D + PPAD) + col] = __uint_as_float((unsigned)u[k]);
        break;
      }
      if (++spins > LL_SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (!ok) ll_ok = 0;
  lds_barrier();
  return ll_ok != 0;
}

__global__ __launch_bounds__(SW * 64) void stack_fwd_kernel(const StackBlockPtrs* __restrict__ tab, const float* __restrict__ x0,
                                                            float* __restrict__ Hs, float* __restrict__ X1,
                                                            float* __restrict__ U, float* __restrict__ A,
                                                            float* __restrict__ ST, int L, int M, int D, float eps,
                                                            void* scratch, SerDropout drop) {
  extern __shared__ float lds_dyn[];
  __shared__ float red[SW][64][4];
  __shared__ float st[16][4];
  const StackScratch sc = stack_scratch(scratch);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16, n = n0 + i;
  const int nch4 = D >> 2, nchunk = D >> 4, PD = D + PPAD;
  const float invD = 1.0f / (float)D;            // exact for the power-of-two widths in use
  float* panel = lds_dyn;                             // [16][PD]  block input h, then a
  float* upan = lds_dyn + 16 * PD;                    // [16][PD]  u = LN(LN(h))
  const int nit = (nchunk - w + SW - 1) / SW;          // <= 4 (D <= 512)
  const long long MD = (long long)M * D;
  const int ri = min(i, M - 1);
  const unsigned tagbase = (*sc.launch + 1u) << 12;
  const long long slot_elems = (long long)16 * D;
  unsigned long long* dbg = g_stack_dbg;
  // the two nn.Dropout layers of every block (ref classifier.py:83,85): site ids drop.site + 2 blk (+1)
  const bool dropping = drop.state != nullptr && drop.p > 0.f;
  const unsigned long long dstate = dropping ? *drop.state : 0ull;
  const unsigned dthresh = ser_drop_thresh(drop.p);
  const float dscale = 1.0f / (1.0f - drop.p);
  float4 b[4], bn[4];
  {
    const float* wr = tab[0].W1 + (long long)n * D + q * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) b[u] = *(const float4*)(wr + min(w + u * SW, nchunk - 1) * 16);
  }
  for (int blk = 0; blk < L; ++blk) {
    const StackBlockPtrs P = tab[blk];
    const int pA = 2 * blk, pB = 2 * blk + 1;          // phase numbers; phase p writes slot p & 1 with tag tagbase + p + 1
    // weight slice of the second Linear: a whole phase ahead of its use
    {
      const float* wr = P.W2 + (long long)n * D + q * 4;
#pragma unroll
      for (int u = 0; u < 4; ++u) bn[u] = *(const float4*)(wr + min(w + u * SW, nchunk - 1) * 16);
    }
    // small per-block parameters: requested before the hand-off wait, used after it
    float4 pg[2][2], pb[2][2];                          // [LayerNorm][chunk of this lane]
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = min(lane + 64 * e, nch4 - 1);
      pg[0][e] = *(const float4*)(P.g1 + c * 4); pb[0][e] = *(const float4*)(P.b1 + c * 4);
      pg[1][e] = *(const float4*)(P.g2 + c * 4); pb[1][e] = *(const float4*)(P.b2 + c * 4);
    }
    const float c1n = P.c1[n], g1n = P.g1[n], b1n = P.b1[n], c2n = P.c2[n];
    STACK_MARK(0);
    // ---- block input -> LDS panel
    if (blk == 0) {
      for (int idx = threadIdx.x; idx < M * nch4; idx += SW * 64) {
        const float4 v = ((const float4*)x0)[idx];
        *(float4*)(panel + (idx / nch4) * PD + (idx % nch4) * 4) = v;
      }
      lds_barrier();
    } else {
      if (!ll_fetch_panel(sc.ring + ((pA - 1) & 1) * slot_elems, panel, M, D, tagbase + pA, sc.abort)) return;
    }
    STACK_MARK(1);
    // ---- statistics of both LayerNorms, one wave per row (rows w, w + 8)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = w + SW * k;
      const int rr = min(row, M - 1);
      float4 v[2];
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = lane + 64 * e;
        v[e] = c < nch4 ? *(const float4*)(panel + rr * PD + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[e].x + v[e].y) + (v[e].z + v[e].w);
      }
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const float mean = wave_sum(s) * invD;
        float qq = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e)
          if (lane + 64 * e < nch4) {
            const float a_ = v[e].x - mean, bb = v[e].y - mean, c2 = v[e].z - mean, d = v[e].w - mean;
            qq += (a_ * a_ + bb * bb) + (c2 * c2 + d * d);
          }
        const float rstd = rsqrtf(wave_sum(qq) * invD + eps);      // v_rsq_f32 (1 ulp) instead of IEEE sqrt + divide
        if (lane == 0) { st[row][2 * pass] = mean; st[row][2 * pass + 1] = rstd; }
        s = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = lane + 64 * e;
          if (c < nch4) {
            const float4 gm = pg[pass][e], bt = pb[pass][e];
            float4 o;
            o.x = (v[e].x - mean) * rstd * gm.x + bt.x; o.y = (v[e].y - mean) * rstd * gm.y + bt.y;
            o.z = (v[e].z - mean) * rstd * gm.z + bt.z; o.w = (v[e].w - mean) * rstd * gm.w + bt.w;
            if ((c >> 2) == (int)blockIdx.x && row < M)     // this workgroup's 16 columns of the dense copy
              *(float4*)((pass == 0 ? X1 : U) + blk * MD + (long long)row * D + c * 4) = o;
            if (pass == 1) *(float4*)(upan + row * PD + c * 4) = o;
            v[e] = o;
            s += (o.x + o.y) + (o.z + o.w);
          }
        }
      }
    }
    lds_barrier();
    if (blockIdx.x == 0 && threadIdx.x < 64) {
      const int row = threadIdx.x >> 2, which = threadIdx.x & 3;
      if (row < M) ST[(long long)blk * 4 * M + which * M + row] = st[row][which];
    }
    STACK_MARK(2);
    // ---- a = relu(u W1^T + c1)
    {
      const float* ur = upan + i * PD + q * 4;          // rows >= M hold copies of row M-1 (never stored)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float4 t = *(const float4*)(ur + min(w + u * SW, nchunk - 1) * 16);
        if (u >= nit) t = make_float4(0.f, 0.f, 0.f, 0.f);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, b[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.y, b[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.z, b[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.w, b[u].w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][lane][r] = acc[r];
    }
    lds_barrier();
    STACK_MARK(3);
    float x1v[4] = {0.f, 0.f, 0.f, 0.f};                 // residual operand of this thread's outputs (wave 0)
    if (w == 0) {
      const float bv = c1n;
      ull* slot = sc.ring + (pA & 1) * slot_elems;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = q * 4 + r;
        if (m >= M) continue;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < SW; ++ww) v += red[ww][lane][r];
        v = fmaxf(v + bv, 0.f);
        if (dropping) v *= ser_drop_mult(dstate, drop.site + 2u * blk, (unsigned)(m * D + n), dthresh, dscale);
        st_ll(slot + m * D + n, v, tagbase + pA + 1);
        A[blk * MD + (long long)m * D + n] = v;
        x1v[r] = (panel[m * PD + n] - st[m][0]) * st[m][1] * g1n + b1n;
      }
    }
    // next block's first weight slice (used two phases from now)
    if (blk + 1 < L) {
      const float* wr = tab[blk + 1].W1 + (long long)n * D + q * 4;
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = *(const float4*)(wr + min(w + u * SW, nchunk - 1) * 16);
    }
    STACK_MARK(4);
    // ---- h' = x1 + a W2^T + c2
    if (!ll_fetch_panel(sc.ring + (pA & 1) * slot_elems, panel, M, D, tagbase + pA + 1, sc.abort)) return;
    STACK_MARK(5);
    {
      const float* ar = panel + ri * PD + q * 4;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float4 a = *(const float4*)(ar + min(w + u * SW, nchunk - 1) * 16);
        if (u >= nit) a = make_float4(0.f, 0.f, 0.f, 0.f);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bn[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bn[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bn[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bn[u].w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][lane][r] = acc[r];
    }
    lds_barrier();
    STACK_MARK(6);
    if (w == 0) {
      const float bv = c2n;
      ull* slot = sc.ring + (pB & 1) * slot_elems;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = q * 4 + r;
        if (m >= M) continue;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < SW; ++ww) v += red[ww][lane][r];
        v = v + bv;
        if (dropping) v *= ser_drop_mult(dstate, drop.site + 2u * blk + 1u, (unsigned)(m * D + n), dthresh, dscale);
        v += x1v[r];
        if (blk + 1 < L) st_ll(slot + m * D + n, v, tagbase + pB + 1);
        Hs[blk * MD + (long long)m * D + n] = v;
      }
    }
    STACK_MARK(7);
    lds_barrier();                                   // `red` and the panel are rewritten by the next block
  }
  // every workgroup has read the launch counter before any can get here
  if (blockIdx.x == 0 && threadIdx.x == 0) *sc.launch = *sc.launch + 1u;
}

// Backward of the same stack.  Per block (last first), with dh' the gradient at the block output:
//   da = (dh' W2) * relu'(a)      du = da W1      dx1 = dh' + LN2'(du)      dh = LN1'(dx1)
// Workgroup b owns columns [16 b, 16 b + 16) of da and du (the reduction over the rows of W is split over the waves,
// as in skinny_dgrad16_kernel); dh' lives in an LDS panel that every workgroup recomputes from du (row-wise
// LayerNorm backward, one wave per row, as ln2_bwd_kernel).  Two tagged hand-offs per block (da, du).  dh', da, du
// and dx1 of every block are kept for the batched weight / LayerNorm-parameter gradients issued after this kernel.
__global__ __launch_bounds__(SW * 64) void stack_bwd_kernel(const StackBlockPtrs* __restrict__ tab, const float* __restrict__ x0,
                                                            const float* __restrict__ Hs, const float* __restrict__ X1,
                                                            const float* __restrict__ A, const float* __restrict__ ST,
                                                            float* __restrict__ DH, float* __restrict__ DA,
                                                            float* __restrict__ DU, float* __restrict__ DX1, int L, int M,
                                                            int D, void* scratch, SerDropout drop, float* __restrict__ DT) {
  extern __shared__ float lds_dyn[];
  const int PD = D + PPAD;
  const float invD = 1.0f / (float)D;
  float* panel = lds_dyn;                             // [16][PD]  gradient at the current block's output
  float* xpan = lds_dyn + 16 * PD;                    // [16][PD]  da, then du, as they arrive
  float* x1pan = lds_dyn + 2 * 16 * PD;               // [16][D]   x1 of the current block   } filled by LDS-DMA at the top of
  float* hpan = x1pan + 16 * D;                       // [16][D]   input of the current block } the block: no registers held
  float* gpan = hpan + 16 * D;                        // [2][D]    gamma of LN1, LN2
  __shared__ float red[SW][64][4];
  const StackScratch sc = stack_scratch(scratch);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int c = blockIdx.x * 16 + i;                  // output column of this lane
  const int nch4 = D >> 2;
  const int nsteps = D >> 2;                          // reduction steps of 4 rows of W
  const int nit = (nsteps - w + SW - 1) / SW;         // <= 16
  const long long MD = (long long)M * D;
  const int ri = min(i, M - 1);
  const unsigned tagbase = (*sc.launch + 1u) << 12;
  const long long slot_elems = (long long)16 * D;
  unsigned long long* dbg = g_stack_dbg ? g_stack_dbg + 8 * 64 : nullptr;     // second half of the timeline buffer
  // dropout: dh' reaches W2 through the second dropout of the block (mask regenerated), da carries the first one's
  // scale; DT[i] = dropped dh' is kept for the batched weight gradients (dW2 = DT^T a)
  const bool dropping = drop.state != nullptr && drop.p > 0.f;
  const unsigned long long dstate = dropping ? *drop.state : 0ull;
  const unsigned dthresh = ser_drop_thresh(drop.p);
  const float dscale = 1.0f / (1.0f - drop.p);
  for (int idx = threadIdx.x; idx < M * nch4; idx += SW * 64) {
    const float4 v = ((const float4*)(DH + (long long)L * MD))[idx];
    *(float4*)(panel + (idx / nch4) * PD + (idx % nch4) * 4) = v;
    if (dropping && blockIdx.x == 0) {
      const unsigned e0 = (unsigned)idx * 4u, sid = drop.site + 2u * (L - 1) + 1u;
      ((float4*)(DT + (long long)L * MD))[idx] =
          make_float4(v.x * ser_drop_mult(dstate, sid, e0, dthresh, dscale), v.y * ser_drop_mult(dstate, sid, e0 + 1, dthresh, dscale),
                      v.z * ser_drop_mult(dstate, sid, e0 + 2, dthresh, dscale), v.w * ser_drop_mult(dstate, sid, e0 + 3, dthresh, dscale));
    }
  }
  float bw[16], bw2[16];
  {
    const float* Wp = tab[L - 1].W2;
#pragma unroll
    for (int u = 0; u < 16; ++u) bw[u] = Wp[(long long)min((w + u * SW) * 4 + q, D - 1) * D + c];
  }
  lds_barrier();
  int phase = 0;
  for (int blk = L - 1; blk >= 0; --blk) {
    const StackBlockPtrs P = tab[blk];
    const int p1 = phase, p2 = phase + 1;
    phase += 2;
    // weights of the second product and the local LayerNorm operands: a phase ahead of their use
#pragma unroll
    for (int u = 0; u < 16; ++u) bw2[u] = P.W1[(long long)min((w + u * SW) * 4 + q, D - 1) * D + c];
    const float* hin = blk == 0 ? x0 : Hs + (long long)(blk - 1) * MD;
    const float* stp = ST + (long long)blk * 4 * M;
    float sm[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = min(w + SW * k, M - 1);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) sm[k][t4] = stp[t4 * M + row];
    }
    // saved activations of this block -> LDS, asynchronously (needed after the second hand-off; the LDS-DMA queue is
    // in order, so they have landed when the tagged loads issued later return, and the barrier inside ll_fetch_panel
    // publishes them to every wave).  The previous block's LayerNorm backward finished with a workgroup barrier.
    for (int idx = threadIdx.x; idx < 16 * nch4; idx += SW * 64) {      // uniform trip count: 16 nch4 is a multiple of 64
      const int row = min(idx / nch4, M - 1), ch = idx % nch4;
      const int slot64 = (idx & ~63) * 16;                               // wave-uniform LDS base, lane-linear 16-byte slots
      __builtin_amdgcn_global_load_lds((gptr_t)(X1 + blk * MD + (long long)row * D + ch * 4), (lptr_t)((char*)x1pan + slot64), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(hin + (long long)row * D + ch * 4), (lptr_t)((char*)hpan + slot64), 16, 0, 0);
    }
    for (int idx = threadIdx.x; idx < 2 * nch4; idx += SW * 64) {        // the two gamma vectors, same route
      const float* src = idx < nch4 ? P.g1 + idx * 4 : P.g2 + (idx - nch4) * 4;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)((char*)gpan + (idx & ~63) * 16), 16, 0, 0);
    }
    float amask[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) amask[r] = A[blk * MD + (long long)min(q * 4 + r, M - 1) * D + c];
    STACK_MARK(0);
    // ---- da = (dh' W2) * relu'(a)
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int nn = (w + u * SW) * 4 + q;
        float av = (u < nit && nn < D) ? panel[ri * PD + min(nn, D - 1)] : 0.f;
        if (dropping) av *= ser_drop_mult(dstate, drop.site + 2u * blk + 1u, (unsigned)(ri * D + min(nn, D - 1)), dthresh, dscale);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[u], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][lane][r] = acc[r];
    }
    lds_barrier();
    STACK_MARK(1);
    if (w == 0) {
      ull* slot = sc.ring + (p1 & 1) * slot_elems;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = q * 4 + r;
        if (m >= M) continue;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < SW; ++ww) v += red[ww][lane][r];
        v = amask[r] > 0.f ? (dropping ? v * dscale : v) : 0.f;      // a > 0 <=> ReLU active and kept by the first dropout
        st_ll(slot + m * D + c, v, tagbase + p1 + 1);
        DA[blk * MD + (long long)m * D + c] = v;
      }
    }
    if (blk > 0) {
      const float* Wp = tab[blk - 1].W2;
#pragma unroll
      for (int u = 0; u < 16; ++u) bw[u] = Wp[(long long)min((w + u * SW) * 4 + q, D - 1) * D + c];
    }
    STACK_MARK(2);
    // ---- du = da W1
    if (!ll_fetch_panel(sc.ring + (p1 & 1) * slot_elems, xpan, M, D, tagbase + p1 + 1, sc.abort)) return;
    STACK_MARK(3);
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int nn = (w + u * SW) * 4 + q;
        const float a1 = (u < nit && nn < D) ? xpan[ri * PD + min(nn, D - 1)] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bw2[u], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][lane][r] = acc[r];
    }
    lds_barrier();
    STACK_MARK(4);
    if (w == 0) {
      ull* slot = sc.ring + (p2 & 1) * slot_elems;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = q * 4 + r;
        if (m >= M) continue;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < SW; ++ww) v += red[ww][lane][r];
        st_ll(slot + m * D + c, v, tagbase + p2 + 1);
        DU[blk * MD + (long long)m * D + c] = v;
      }
    }
    // ---- panel <- LN1'(panel + LN2'(du)), rows w, w + 8
    STACK_MARK(5);
    if (!ll_fetch_panel(sc.ring + (p2 & 1) * slot_elems, xpan, M, D, tagbase + p2 + 1, sc.abort)) return;
    STACK_MARK(6);
    {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int row = w + SW * k;
        if (row >= M) continue;
        const float m1 = sm[k][0], r1 = sm[k][1], m2 = sm[k][2], r2 = sm[k][3];
        float4 xh[2], dgv[2], d1[2];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ch = lane + 64 * e;
          if (ch < nch4) {
            const float4 zz = *(const float4*)(x1pan + row * D + ch * 4);
            const float4 d = *(const float4*)(xpan + row * PD + ch * 4);
            const float4 gm = *(const float4*)(gpan + D + ch * 4);
            xh[e] = make_float4((zz.x - m2) * r2, (zz.y - m2) * r2, (zz.z - m2) * r2, (zz.w - m2) * r2);
            dgv[e] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
            s1 += (dgv[e].x + dgv[e].y) + (dgv[e].z + dgv[e].w);
            s2 += (dgv[e].x * xh[e].x + dgv[e].y * xh[e].y) + (dgv[e].z * xh[e].z + dgv[e].w * xh[e].w);
          }
        }
        float a1 = wave_sum(s1) * invD, a2 = wave_sum(s2) * invD;
        s1 = 0.f; s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ch = lane + 64 * e;
          if (ch < nch4) {
            d1[e].x = r2 * (dgv[e].x - a1 - xh[e].x * a2);
            d1[e].y = r2 * (dgv[e].y - a1 - xh[e].y * a2);
            d1[e].z = r2 * (dgv[e].z - a1 - xh[e].z * a2);
            d1[e].w = r2 * (dgv[e].w - a1 - xh[e].w * a2);
            const float4 ee = *(const float4*)(panel + row * PD + ch * 4);
            d1[e].x += ee.x; d1[e].y += ee.y; d1[e].z += ee.z; d1[e].w += ee.w;
            if ((ch >> 2) == (int)blockIdx.x) *(float4*)(DX1 + blk * MD + (long long)row * D + ch * 4) = d1[e];
            const float4 zz = *(const float4*)(hpan + row * D + ch * 4);
            const float4 gm = *(const float4*)(gpan + ch * 4);
            xh[e] = make_float4((zz.x - m1) * r1, (zz.y - m1) * r1, (zz.z - m1) * r1, (zz.w - m1) * r1);
            dgv[e] = make_float4(d1[e].x * gm.x, d1[e].y * gm.y, d1[e].z * gm.z, d1[e].w * gm.w);
            s1 += (dgv[e].x + dgv[e].y) + (dgv[e].z + dgv[e].w);
            s2 += (dgv[e].x * xh[e].x + dgv[e].y * xh[e].y) + (dgv[e].z * xh[e].z + dgv[e].w * xh[e].w);
          }
        }
        a1 = wave_sum(s1) * invD; a2 = wave_sum(s2) * invD;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ch = lane + 64 * e;
          if (ch < nch4) {
            float4 o;
            o.x = r1 * (dgv[e].x - a1 - xh[e].x * a2);
            o.y = r1 * (dgv[e].y - a1 - xh[e].y * a2);
            o.z = r1 * (dgv[e].z - a1 - xh[e].z * a2);
            o.w = r1 * (dgv[e].w - a1 - xh[e].w * a2);
            *(float4*)(panel + row * PD + ch * 4) = o;
            if ((ch >> 2) == (int)blockIdx.x) {
              *(float4*)(DH + blk * MD + (long long)row * D + ch * 4) = o;
              if (dropping && blk > 0) {                 // DH[blk] is the gradient at the output of block blk - 1
                const unsigned e0 = (unsigned)(row * D + ch * 4), sid = drop.site + 2u * (blk - 1) + 1u;
                *(float4*)(DT + blk * MD + (long long)row * D + ch * 4) =
                    make_float4(o.x * ser_drop_mult(dstate, sid, e0, dthresh, dscale), o.y * ser_drop_mult(dstate, sid, e0 + 1, dthresh, dscale),
                                o.z * ser_drop_mult(dstate, sid, e0 + 2, dthresh, dscale), o.w * ser_drop_mult(dstate, sid, e0 + 3, dthresh, dscale));
              }
            }
          }
        }
      }
    }
    STACK_MARK(7);
    lds_barrier();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *sc.launch = *sc.launch + 1u;
}

// LayerNorm-parameter gradients of all blocks in one launch (off the dgrad chain):
//   dg2 = sum_m du * xhat2   db2 = sum_m du   dg1 = sum_m dx1 * xhat1   db1 = sum_m dx1
// grid (D / 64, L); rows are added in the order of ln2_bwd_kernel (wave w owned rows w, w + 8; waves in order).
struct StackGradPtrs {
  float *dg1, *db1, *dg2, *db2;
};
__global__ __launch_bounds__(64) void stack_ln_param_kernel(const StackGradPtrs* __restrict__ gtab, const float* __restrict__ x0,
                                                            const float* __restrict__ Hs, const float* __restrict__ X1,
                                                            const float* __restrict__ ST, const float* __restrict__ DU,
                                                            const float* __restrict__ DX1, int L, int M, int D,
                                                            int accumulate) {
  const int col = blockIdx.x * 64 + threadIdx.x, blk = blockIdx.y;
  if (col >= D) return;
  const long long MD = (long long)M * D;
  const float* hin = blk == 0 ? x0 : Hs + (long long)(blk - 1) * MD;
  const float* stp = ST + (long long)blk * 4 * M;
  float tg1 = 0.f, tb1 = 0.f, tg2 = 0.f, tb2 = 0.f;
  for (int ww = 0; ww < SW; ++ww) {
    float ag1 = 0.f, ab1 = 0.f, ag2 = 0.f, ab2 = 0.f;
    for (int row = ww; row < M; row += SW) {
      const float m1 = stp[row], r1 = stp[M + row], m2 = stp[2 * M + row], r2 = stp[3 * M + row];
      const float d = DU[blk * MD + (long long)row * D + col];
      const float xh2 = (X1[blk * MD + (long long)row * D + col] - m2) * r2;
      ag2 += d * xh2;
      ab2 += d;
      const float d1 = DX1[blk * MD + (long long)row * D + col];
      const float xh1 = (hin[(long long)row * D + col] - m1) * r1;
      ag1 += d1 * xh1;
      ab1 += d1;
    }
    if (ww == 0) { tg1 = ag1; tb1 = ab1; tg2 = ag2; tb2 = ab2; }
    else { tg1 += ag1; tb1 += ab1; tg2 += ag2; tb2 += ab2; }
  }
  const StackGradPtrs G = gtab[blk];
  if (accumulate) { tg1 += G.dg1[col]; tb1 += G.db1[col]; tg2 += G.dg2[col]; tb2 += G.db2[col]; }
  G.dg1[col] = tg1; G.db1[col] = tb1; G.dg2[col] = tg2; G.db2[col] = tb2;
}

}  // namespace

extern "C" int ser_debug_barrier_probe(int G, int rounds, int mode, void* flags, void* data, void* errors, void* stream) {
  SER_REQUIRE(G >= 1 && G <= 64 && rounds >= 1, "barrier probe: bad arguments");
  GridBar gb{(unsigned*)flags, (unsigned*)flags + 64, G};
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(flags, 0, 65 * sizeof(unsigned), st) != hipSuccess) return SER_E_HIP;
  hipLaunchKernelGGL(barrier_probe_kernel, dim3(G), dim3(512), 0, st, gb, (float*)data, rounds, mode, (unsigned*)errors);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// Clock probe: one wave spins on dependent VALU work for `iters` iterations and reports the shader clock it ran at
// (delta s_memtime = shader cycles, delta s_memrealtime = 100 MHz ticks; MI355X_MICROARCH.md, DVFS give-back item 6).
// Launched beside other work it shows what clock the rest of the chip leaves to a latency-bound kernel.
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, int iters) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) x = fmaf(x, 1.0000001f, 1e-7f);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (x == 123456.789f) out[0] = 0;
}
}  // namespace
extern "C" int ser_debug_clock_probe(void* out, int blocks, int iters, void* stream) {
  SER_REQUIRE(out && blocks >= 1 && iters >= 1, "clock probe: bad arguments");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, iters);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

static int stack_check(int L, int M, int D) {
  SER_REQUIRE(L >= 1 && M >= 1 && M <= 16 && D >= 16 && D <= 512 && D % 16 == 0,
              "classifier stack: L=%d M=%d D=%d unsupported (M <= 16, D a multiple of 16 up to 512)", L, M, D);
  return SER_OK;
}

extern "C" int ser_stack_supported(int L, int M, int D) {
  return (L >= 1 && M >= 1 && M <= 16 && D >= 16 && D <= 512 && D % 16 == 0) ? 1 : 0;
}

extern "C" int ser_debug_stack_timeline(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stack_dbg), &p, sizeof(p)) == hipSuccess ? SER_OK : SER_E_HIP;
}

extern "C" size_t ser_stack_scratch_bytes(int D) { return 256 + (size_t)2 * 16 * D * sizeof(unsigned long long); }

extern "C" int ser_stack_fwd(const void* ptr_table, const float* x0, float* Hs, float* X1, float* U, float* A, float* ST,
                             int L, int M, int D, float eps, void* scratch, const void* drop_state, unsigned drop_site,
                             float drop_p, void* stream) {
  SER_TRY(stack_check(L, M, D));
  SER_REQUIRE(scratch != nullptr && ((uintptr_t)scratch & 255) == 0, "classifier stack: scratch must be 256-byte aligned");
  hipLaunchKernelGGL(stack_fwd_kernel, dim3(D / 16), dim3(SW * 64), (size_t)2 * 16 * (D + PPAD) * sizeof(float),
                     (hipStream_t)stream, (const StackBlockPtrs*)ptr_table, x0, Hs, X1, U, A, ST, L, M, D, eps, scratch,
                     SerDropout{(const unsigned long long*)drop_state, drop_site, drop_p});
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_stack_bwd(const void* ptr_table, const float* x0, const float* Hs, const float* X1, const float* A,
                             const float* ST, float* DH, float* DA, float* DU, float* DX1, int L, int M, int D, void* scratch,
                             const void* drop_state, unsigned drop_site, float drop_p, float* DT, void* stream) {
  SER_TRY(stack_check(L, M, D));
  SER_REQUIRE(scratch != nullptr && ((uintptr_t)scratch & 255) == 0, "classifier stack: scratch must be 256-byte aligned");
  hipLaunchKernelGGL(stack_bwd_kernel, dim3(D / 16), dim3(SW * 64), (size_t)(2 * 16 * (D + PPAD) + 2 * 16 * D + 2 * D) * sizeof(float),
                     (hipStream_t)stream, (const StackBlockPtrs*)ptr_table, x0, Hs, X1, A, ST, DH, DA, DU, DX1, L, M, D,
                     scratch, SerDropout{(const unsigned long long*)drop_state, drop_site, drop_p}, DT);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_stack_ln_param_bwd(const void* grad_table, const float* x0, const float* Hs, const float* X1,
                                      const float* ST, const float* DU, const float* DX1, int L, int M, int D, int accumulate,
                                      void* stream) {
  SER_TRY(stack_check(L, M, D));
  hipLaunchKernelGGL(stack_ln_param_kernel, dim3((D + 63) / 64, L), dim3(64), 0, (hipStream_t)stream,
                     (const StackGradPtrs*)grad_table, x0, Hs, X1, ST, DU, DX1, L, M, D, accumulate);
  SER_LAUNCH_CHECK();
  return SER_OK;
}
