// HBM-bound kernels of the frozen encoders: bf16 splitting, LayerNorm, raw-waveform conv0 +
// GroupNorm + GELU, positional-conv slab relayout, XLM-R embedding gather + LayerNorm.
#include <stdarg.h>
#include "ser_common.h"

static thread_local char g_err[512] = "";
void ser_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ser_last_error_string(void) { return g_err; }
extern "C" int ser_abi_version(void) { return 1; }

// A HIP stream restricted to a subset of the compute units (bit i of mask word i/32 = CU i in the runtime's
// enumeration).  Used to keep a few CUs free for the latency-bound head kernels while the encoder GEMMs of the next
// batch run on the rest.  The caller owns the stream (ser_stream_destroy).
extern "C" int ser_stream_create_cu_masked(const uint32_t* mask, int words, void** stream_out) {
  SER_REQUIRE(mask && words > 0 && stream_out, "stream_create_cu_masked: bad arguments");
  hipStream_t st = nullptr;
  SER_CHECK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
  *stream_out = (void*)st;
  return SER_OK;
}
extern "C" int ser_stream_destroy(void* stream) {
  if (stream) SER_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
  return SER_OK;
}

// ------------------------------------------------------------------------------------------
// fp32 -> split bf16 planes
// ------------------------------------------------------------------------------------------
__global__ void split_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                             long long n) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  const bool il = ser_is_il(hi, lo);       // interleaved planes (lo == hi + 32): offsets map by ser_il_off
  for (; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = *(const float4*)(x + i);
      bf16_t h[4], l[4];
      split_bf16(v.x, h[0], l[0]); split_bf16(v.y, h[1], l[1]);
      split_bf16(v.z, h[2], l[2]); split_bf16(v.w, h[3], l[3]);
      const long long o = il ? ser_il_off(i) : i;
      *(uint2*)(hi + o) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
      if (lo) *(uint2*)(lo + o) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
    } else {
      for (long long k = i; k < n; ++k) {
        bf16_t h, l;
        split_bf16(x[k], h, l);
        const long long o = il ? ser_il_off(k) : k;
        hi[o] = h;
        if (lo) lo[o] = l;
      }
    }
  }
}

int ser_launch_split(const float* x, bf16_t* hi, bf16_t* lo, long long n, hipStream_t st) {
  if (n <= 0) return SER_OK;
  SER_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)hi % 8 == 0) && (!lo || (uintptr_t)lo % 8 == 0),
              "split_bf16: pointers must be 16/8-byte aligned");
  SER_REQUIRE(!ser_is_il(hi, lo) || n % SER_IL_GROUP == 0, "split_bf16: the interleaved layout needs n %% %d == 0", SER_IL_GROUP);
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, hi, lo, n);
  SER_LAUNCH_CHECK();
  return SER_OK;
}
extern "C" int ser_split_bf16(const float* x, uint16_t* hi, uint16_t* lo, long long n, void* stream) {
  return ser_launch_split(x, hi, lo, n, (hipStream_t)stream);
}

// x [R, C] fp32 (row stride ldx) -> split planes of x^T: C rows of Rp >= R values (zero beyond R), Rp % 32 == 0.
// lo == hi + 32 elements: interleaved layout (row c occupies 2 Rp bf16, every 32 hi values followed by their 32 lo values);
// lo == nullptr: the hi plane alone, rows of Rp values.  A 32 x 32 tile goes through LDS, so reads run along C and writes
// along R.  Feeds the K-contiguous NT GEMM with the transposed operands of a Linear layer's backward products
// (dx = dy W: W^T; dW = dy^T x: dy^T and x^T), so that the fine-tuning path runs on the encoder tile kernels.
__global__ __launch_bounds__(256) void split_t_kernel(const float* __restrict__ x, int R, int C, long long ldx, bf16_t* __restrict__ hi,
                                                      bf16_t* __restrict__ lo, int Rp) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? x[(long long)r * ldx + c] : 0.f;
  }
  __syncthreads();
  const bool il = lo != nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i;
    if (c >= C) continue;
    bf16_t h, l;
    split_bf16(tile[tx][ty + 8 * i], h, l);
    if (il) {
      const long long o = (long long)c * 2 * Rp + 2 * r0 + tx;      // il_off(r0 + tx) with r0 % 32 == 0
      hi[o] = h;
      hi[o + SER_IL_GROUP] = l;
    } else {
      hi[(long long)c * Rp + r0 + tx] = h;
    }
  }
}
// Both operand forms of one fp32 matrix in ONE pass over it: the planes of x [R, C] (rows of C values, C % 32 == 0) and the
// planes of x^T [C, Rp] as above.  s_lo / t_lo == nullptr: that form's hi plane alone.  A Linear layer needs x and W straight in
// its forward product and transposed in its backward products (and dy both ways in backward): one launch instead of two each.
__global__ __launch_bounds__(256) void split_both_kernel(const float* __restrict__ x, int R, int C, long long ldx, bf16_t* __restrict__ s_hi,
                                                         bf16_t* __restrict__ s_lo, bf16_t* __restrict__ t_hi, bf16_t* __restrict__ t_lo,
                                                         int Rp, float* __restrict__ colpart) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const bool s_il = s_lo != nullptr, t_il = t_lo != nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    const bool in = r < R && c < C;
    const float v = in ? x[(long long)r * ldx + c] : 0.f;
    tile[ty + 8 * i][tx] = v;
    if (in) {
      bf16_t h, l;
      split_bf16(v, h, l);
      if (s_il) {
        const long long o = 2 * ((long long)r * C + c0) + tx;          // il_off(r C + c0 + tx), c0 % 32 == 0 and C % 32 == 0
        s_hi[o] = h;
        s_hi[o + SER_IL_GROUP] = l;
      } else {
        s_hi[(long long)r * C + c] = h;
      }
    }
  }
  __syncthreads();
  // column sums of this block's 32 rows (rows >= R are zeros), rows added in increasing order: colpart[blockIdx.x][c] - the
  // first stage of a bias gradient (x = dy of a Linear layer) at no extra pass over x
  if (colpart != nullptr && ty == 0 && c0 + tx < C) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) a += tile[r][tx];
    colpart[(long long)blockIdx.x * C + c0 + tx] = a;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i;
    if (c >= C) continue;
    bf16_t h, l;
    split_bf16(tile[tx][ty + 8 * i], h, l);
    if (t_il) {
      const long long o = (long long)c * 2 * Rp + 2 * r0 + tx;
      t_hi[o] = h;
      t_hi[o + SER_IL_GROUP] = l;
    } else {
      t_hi[(long long)c * Rp + r0 + tx] = h;
    }
  }
}
// The same for MANY matrices in one launch (the weights of every Linear layer of an encoder, once per step): blockIdx.z picks a
// descriptor of 12 64-bit words {x, s_hi, s_lo, t_hi, t_lo, R, C, Rp, t_roff, ldx, cover, bias_src | bias_dst packed below}.
// A matrix may be a row block of a vertically fused operand (q | k | v): its straight planes start at its first row (the pointer)
// and its transposed planes at row offset t_roff of the fused operand's Rp-long rows.  cover = rows to walk (R, or the padded Rp
// of a stand-alone matrix: the padding is zero-filled).  An optional bias copy (bias_n floats) rides on block (0, 0).
struct SplitDesc {
  const float* x; bf16_t* s_hi; bf16_t* s_lo; bf16_t* t_hi; bf16_t* t_lo;
  long long R, C, Rp, t_roff, ldx, cover;
  const float* bias_src; float* bias_dst; long long bias_n;
  long long blk0, cblocks;                             // first workgroup of this matrix in the launch, its 32-column blocks
};
__global__ __launch_bounds__(256) void split_both_multi_kernel(const SplitDesc* __restrict__ tab, int nprob) {
  // the launch has exactly the workgroups the matrices need (sum of cover / 32 x C / 32): find this one's matrix by bisection
  // over the descriptors' first-workgroup numbers
  const long long bid = blockIdx.x;
  int lo = 0, hi = nprob - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
  }
  const SplitDesc d = tab[lo];
  const long long local = bid - d.blk0;
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = (int)(local / d.cblocks) * 32, c0 = (int)(local % d.cblocks) * 32;
  if (local == 0 && d.bias_src != nullptr)
    for (int i = threadIdx.x; i < d.bias_n; i += 256) d.bias_dst[i] = d.bias_src[i];
  const int R = (int)d.R, C = (int)d.C;
  const bool s_il = d.s_lo != nullptr, t_il = d.t_lo != nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    const bool in = r < R && c < C;
    const float v = in ? d.x[(long long)r * d.ldx + c] : 0.f;
    tile[ty + 8 * i][tx] = v;
    if (in) {
      bf16_t h, l;
      split_bf16(v, h, l);
      if (s_il) {
        const long long o = 2 * ((long long)r * C + c0) + tx;
        d.s_hi[o] = h;
        d.s_hi[o + SER_IL_GROUP] = l;
      } else {
        d.s_hi[(long long)r * C + c] = h;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i;
    if (c >= C) continue;
    bf16_t h, l;
    split_bf16(tile[tx][ty + 8 * i], h, l);
    if (t_il) {
      const long long o = (long long)c * 2 * d.Rp + 2 * (d.t_roff + r0) + tx;
      d.t_hi[o] = h;
      d.t_hi[o + SER_IL_GROUP] = l;
    } else {
      d.t_hi[(long long)c * d.Rp + d.t_roff + r0 + tx] = h;
    }
  }
}
// table: nprob descriptors of 16 64-bit words each in DEVICE memory (layout of SplitDesc), blk0 ascending from 0; total_blocks =
// sum over the matrices of (cover / 32) x (C / 32).  Every C and t_roff must be a multiple of 32, every Rp and cover a multiple of 32.
extern "C" int ser_split_bf16_both_multi(const void* table, int nprob, long long total_blocks, void* stream) {
  SER_REQUIRE(table && nprob > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "split_bf16_both_multi: bad arguments");
  static_assert(sizeof(SplitDesc) == 16 * 8, "descriptor layout");
  hipLaunchKernelGGL(split_both_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const SplitDesc*)table, nprob);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// out[i] = slabs[0][i] + slabs[1][i] + ... (slice order) + bias[i % N]: the consumer of a split-K product that carries a bias
__global__ void sum_slabs_bias_kernel(const float* __restrict__ slabs, int ks, long long n, int N, const float* __restrict__ bias,
                                      float* __restrict__ out) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 a = *(const float4*)(slabs + i * 4);
    for (int s = 1; s < ks; ++s) {
      const float4 v = *(const float4*)(slabs + (long long)s * n + i * 4);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (bias) {
      const float4 bv = *(const float4*)(bias + (i * 4) % N);
      a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
    }
    *(float4*)(out + i * 4) = a;
  }
}
extern "C" int ser_sum_slabs_bias(const float* slabs, int ks, long long n, int N, const float* bias, float* out, void* stream) {
  SER_REQUIRE(slabs && out && ks >= 1 && n > 0 && n % 4 == 0 && N > 0 && N % 4 == 0 && n % N == 0, "sum_slabs_bias: bad arguments");
  const long long blocks = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(sum_slabs_bias_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, slabs, ks, n, N,
                     bias, out);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_split_bf16_both_colsum(const float* x, int R, int C, long long ldx, uint16_t* s_hi, uint16_t* s_lo, uint16_t* t_hi,
                                          uint16_t* t_lo, int Rp, float* colpart, void* stream) {
  SER_REQUIRE(x && s_hi && t_hi && R > 0 && C > 0 && C % SER_IL_GROUP == 0 && Rp >= R && Rp % SER_IL_GROUP == 0,
              "split_bf16_both: bad arguments (R=%d C=%d Rp=%d)", R, C, Rp);
  SER_REQUIRE((s_lo == nullptr || ser_is_il(s_hi, s_lo)) && (t_lo == nullptr || ser_is_il(t_hi, t_lo)),
              "split_bf16_both: both planes are written only in the interleaved layout (lo == hi + 32)");
  hipLaunchKernelGGL(split_both_kernel, dim3(Rp / 32, C / 32), dim3(256), 0, (hipStream_t)stream, x, R, C, ldx, (bf16_t*)s_hi,
                     (bf16_t*)s_lo, (bf16_t*)t_hi, (bf16_t*)t_lo, Rp, colpart);
  SER_LAUNCH_CHECK();
  return SER_OK;
}
extern "C" int ser_split_bf16_both(const float* x, int R, int C, long long ldx, uint16_t* s_hi, uint16_t* s_lo, uint16_t* t_hi,
                                   uint16_t* t_lo, int Rp, void* stream) {
  SER_REQUIRE(x && s_hi && t_hi && R > 0 && C > 0 && C % SER_IL_GROUP == 0 && Rp >= R && Rp % SER_IL_GROUP == 0,
              "split_bf16_both: bad arguments (R=%d C=%d Rp=%d)", R, C, Rp);
  SER_REQUIRE((s_lo == nullptr || ser_is_il(s_hi, s_lo)) && (t_lo == nullptr || ser_is_il(t_hi, t_lo)),
              "split_bf16_both: both planes are written only in the interleaved layout (lo == hi + 32)");
  hipLaunchKernelGGL(split_both_kernel, dim3(Rp / 32, C / 32), dim3(256), 0, (hipStream_t)stream, x, R, C, ldx, (bf16_t*)s_hi,
                     (bf16_t*)s_lo, (bf16_t*)t_hi, (bf16_t*)t_lo, Rp, (float*)nullptr);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_split_bf16_t(const float* x, int R, int C, long long ldx, uint16_t* hi, uint16_t* lo, int Rp, void* stream) {
  SER_REQUIRE(x && hi && R > 0 && C > 0 && Rp >= R && Rp % SER_IL_GROUP == 0, "split_bf16_t: bad arguments (R=%d C=%d Rp=%d)", R, C, Rp);
  SER_REQUIRE(lo == nullptr || ser_is_il(hi, lo), "split_bf16_t: both planes are written only in the interleaved layout (lo == hi + 32)");
  hipLaunchKernelGGL(split_t_kernel, dim3(Rp / 32, ceil_div(C, 32)), dim3(256), 0, (hipStream_t)stream, x, R, C, ldx, hi, lo, Rp);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// ------------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wave per row, row kept in registers (D <= 1024, D % 4 == 0)
// ------------------------------------------------------------------------------------------
struct LnProb {
  const float *x, *x2, *gamma, *beta;
  float eps;
  int rows, D;
  float* y;
  bf16_t *yhi, *ylo;
  int nsl;                 // input = x + slabs 1 .. nsl-1 (split-K partial sums, `sstride` elements apart) + bias + x2
  long long sstride;
  const float* bias;
};

template <int NV>  // float4 chunks per lane
SER_DEVFN void ln_row(const LnProb& P, const int row, const int lane) {
  const float* __restrict__ x = P.x;
  const float* __restrict__ x2 = P.x2;
  const float* __restrict__ gamma = P.gamma;
  const float* __restrict__ beta = P.beta;
  float* __restrict__ y = P.y;
  bf16_t* __restrict__ yhi = P.yhi;
  bf16_t* __restrict__ ylo = P.ylo;
  const float eps = P.eps;
  const int D = P.D;
  const int nchunk = D >> 2;
  const float* xr = x + (long long)row * D;
  const float* x2r = x2 ? x2 + (long long)row * D : nullptr;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      v[i] = *(const float4*)(xr + c * 4);
      for (int sl = 1; sl < P.nsl; ++sl) {      // split-K slabs of the producing GEMM, summed in slab order
        const float4 w = *(const float4*)(xr + sl * P.sstride + c * 4);
        v[i].x += w.x; v[i].y += w.y; v[i].z += w.z; v[i].w += w.w;
      }
      if (P.bias) {
        const float4 w = *(const float4*)(P.bias + c * 4);
        v[i].x += w.x; v[i].y += w.y; v[i].z += w.z; v[i].w += w.w;
      }
      if (x2r) {
        const float4 w = *(const float4*)(x2r + c * 4);
        v[i].x += w.x; v[i].y += w.y; v[i].z += w.z; v[i].w += w.w;
      }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float4 gm = *(const float4*)(gamma + c * 4);
      const float4 bt = *(const float4*)(beta + c * 4);
      float4 o;
      o.x = (v[i].x - mean) * rstd * gm.x + bt.x;
      o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
      o.z = (v[i].z - mean) * rstd * gm.z + bt.z;
      o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
      const long long off = (long long)row * D + c * 4;
      if (y) *(float4*)(y + off) = o;
      if (yhi) {
        bf16_t h[4], l[4];
        split_bf16(o.x, h[0], l[0]); split_bf16(o.y, h[1], l[1]);
        split_bf16(o.z, h[2], l[2]); split_bf16(o.w, h[3], l[3]);
        const long long op = ser_is_il(yhi, ylo) ? ser_il_off(off) : off;
        *(uint2*)(yhi + op) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
        if (ylo) *(uint2*)(ylo + op) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
      }
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnProb P) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= P.rows) return;
  ln_row<NV>(P, row, threadIdx.x & 63);
}

// rows of two independent problems in one launch (same D); problem 0 first
template <int NV>
__global__ __launch_bounds__(256) void layernorm_pair_kernel(const LnProb P0, const LnProb P1) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < P0.rows) ln_row<NV>(P0, row, threadIdx.x & 63);
  else if (row - P0.rows < P1.rows) ln_row<NV>(P1, row - P0.rows, threadIdx.x & 63);
}

int ser_launch_layernorm(const float* x, const float* x2, const float* gamma, const float* beta, float eps, int rows,
                         int D, float* y, bf16_t* yhi, bf16_t* ylo, hipStream_t st) {
  SER_REQUIRE(D % 4 == 0 && D >= 4 && D <= 1024, "layernorm: D=%d unsupported (need D %% 4 == 0, D <= 1024)", D);
  SER_REQUIRE(!ser_is_il(yhi, ylo) || D % SER_IL_GROUP == 0, "layernorm: interleaved output planes need D %% %d == 0", SER_IL_GROUP);
  if (rows <= 0) return SER_OK;
  dim3 grid(ceil_div(rows, 4)), block(256);
  const int nv = ceil_div(D / 4, 64);
  const LnProb P{x, x2, gamma, beta, eps, rows, D, y, yhi, ylo, 1, 0, nullptr};
  switch (nv) {
    case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, P); break;
    case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, st, P); break;
    case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, st, P); break;
    default: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, st, P); break;
  }
  SER_LAUNCH_CHECK();
  return SER_OK;
}

int ser_launch_layernorm_ex(const SerLnArgs& a, hipStream_t st) {
  SER_REQUIRE(a.D % 4 == 0 && a.D >= 4 && a.D <= 1024, "layernorm: D=%d unsupported (need D %% 4 == 0, D <= 1024)", a.D);
  SER_REQUIRE(!ser_is_il(a.yhi, a.ylo) || a.D % SER_IL_GROUP == 0, "layernorm: interleaved output planes need D %% %d == 0", SER_IL_GROUP);
  if (a.rows <= 0) return SER_OK;
  const LnProb P{a.x, a.x2, a.gamma, a.beta, a.eps, a.rows, a.D, a.y, a.yhi, a.ylo, a.nsl < 1 ? 1 : a.nsl, a.sstride, a.bias};
  dim3 grid(ceil_div(a.rows, 4)), block(256);
  switch (ceil_div(a.D / 4, 64)) {
    case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, P); break;
    case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, st, P); break;
    case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, st, P); break;
    default: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, st, P); break;
  }
  SER_LAUNCH_CHECK();
  return SER_OK;
}

int ser_launch_layernorm_pair(const SerLnArgs& a, const SerLnArgs& b, hipStream_t st) {
  if (a.D != b.D || a.rows <= 0 || b.rows <= 0) {
    SER_TRY(ser_launch_layernorm_ex(a, st));
    return ser_launch_layernorm_ex(b, st);
  }
  SER_REQUIRE(a.D % 4 == 0 && a.D >= 4 && a.D <= 1024, "layernorm: D=%d unsupported", a.D);
  SER_REQUIRE(!(ser_is_il(a.yhi, a.ylo) || ser_is_il(b.yhi, b.ylo)) || a.D % SER_IL_GROUP == 0, "layernorm: interleaved output planes need D %% %d == 0", SER_IL_GROUP);
  const LnProb P0{a.x, a.x2, a.gamma, a.beta, a.eps, a.rows, a.D, a.y, a.yhi, a.ylo, a.nsl < 1 ? 1 : a.nsl, a.sstride, a.bias};
  const LnProb P1{b.x, b.x2, b.gamma, b.beta, b.eps, b.rows, b.D, b.y, b.yhi, b.ylo, b.nsl < 1 ? 1 : b.nsl, b.sstride, b.bias};
  dim3 grid(ceil_div(a.rows + b.rows, 4)), block(256);
  switch (ceil_div(a.D / 4, 64)) {
    case 1: hipLaunchKernelGGL(layernorm_pair_kernel<1>, grid, block, 0, st, P0, P1); break;
    case 2: hipLaunchKernelGGL(layernorm_pair_kernel<2>, grid, block, 0, st, P0, P1); break;
    case 3: hipLaunchKernelGGL(layernorm_pair_kernel<3>, grid, block, 0, st, P0, P1); break;
    default: hipLaunchKernelGGL(layernorm_pair_kernel<4>, grid, block, 0, st, P0, P1); break;
  }
  SER_LAUNCH_CHECK();
  return SER_OK;
}
extern "C" int ser_layernorm(const float* x, const float* x2, const float* gamma, const float* beta, float eps,
                             int rows, int D, float* y_f32, uint16_t* y_hi, uint16_t* y_lo, void* stream) {
  return ser_launch_layernorm(x, x2, gamma, beta, eps, rows, D, y_f32, y_hi, y_lo, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// raw waveform statistics: mean and 1/sqrt(var_biased + 1e-7) per clip
// (hf feature_extraction_wav2vec2.py:78-96)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void wave_stats_kernel(const float* __restrict__ wave, int T,
                                                          float2* __restrict__ stats) {
  __shared__ double sh[2][16];
  const float* x = wave + (long long)blockIdx.x * T;
  double s = 0.0, q = 0.0;
  for (int i0 = threadIdx.x; i0 < T; i0 += 8 * 1024) {     // eight independent loads in flight per thread
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = x[min(i0 + k * 1024, T - 1)];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * 1024 < T) {
        const double d = v[k];
        s += d;
        q += d * d;
      }
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ss = 0.0, qq = 0.0;
    for (int i = 0; i < 16; ++i) { ss += sh[0][i]; qq += sh[1][i]; }
    const double mean = ss / T;
    double var = qq / T - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[blockIdx.x] = make_float2((float)mean, (float)(1.0 / sqrt(var + 1e-7)));
  }
}

int ser_launch_wave_stats(const float* wave, int B, int T, void* stats, hipStream_t st) {
  hipLaunchKernelGGL(wave_stats_kernel, dim3(B), dim3(1024), 0, st, wave, T, (float2*)stats);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// ------------------------------------------------------------------------------------------
// conv0 (1 -> C0 channels, kernel KW, stride ST, no bias) + GroupNorm(C0 groups) + GELU
// (hf modeling_wav2vec2.py:302-323).  Pass 1 accumulates sum / sum-of-squares per (clip, channel)
// without storing the conv output; pass 2 recomputes the conv (KW MACs), normalises, applies
// GELU and stores split-bf16 channels-last once.
// ------------------------------------------------------------------------------------------
constexpr int C0_FT = 128;   // frames per workgroup
constexpr int C0_MAXK = 16;

template <bool APPLY>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ wave, const float2* __restrict__ wstats,
                                                    const float* __restrict__ w, int T, int L0, int C0, int KW, int ST,
                                                    float2* __restrict__ partial,          // [B][chunks][C0]   (pass 1)
                                                    const float2* __restrict__ cstats,     // [B][C0] mean,rstd (pass 2)
                                                    const float* __restrict__ gn_g, const float* __restrict__ gn_b,
                                                    bf16_t* __restrict__ yhi, bf16_t* __restrict__ ylo) {
  __shared__ float xs[C0_FT * 8 + C0_MAXK];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int t0 = chunk * C0_FT;
  const int nt = min(C0_FT, L0 - t0);
  const int nsamp = (nt - 1) * ST + KW;
  const float2 ws = wstats[b];
  const float* x = wave + (long long)b * T + (long long)t0 * ST;
  for (int i = threadIdx.x; i < nsamp; i += 256) xs[i] = (x[i] - ws.x) * ws.y;
  __syncthreads();
  for (int cp = threadIdx.x; cp < C0 / 2; cp += 256) {
    const int c = cp * 2;
    float w0[C0_MAXK], w1[C0_MAXK];
#pragma unroll
    for (int j = 0; j < C0_MAXK; ++j) {
      w0[j] = j < KW ? w[c * KW + j] : 0.f;
      w1[j] = j < KW ? w[(c + 1) * KW + j] : 0.f;
    }
    if (!APPLY) {
      float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
      for (int t = 0; t < nt; ++t) {
        float y0 = 0.f, y1 = 0.f;
#pragma unroll
        for (int j = 0; j < C0_MAXK; ++j)
          if (j < KW) {
            const float xv = xs[t * ST + j];
            y0 = fmaf(w0[j], xv, y0);
            y1 = fmaf(w1[j], xv, y1);
          }
        s0 += y0; q0 = fmaf(y0, y0, q0);
        s1 += y1; q1 = fmaf(y1, y1, q1);
      }
      float2* p = partial + ((long long)b * gridDim.x + chunk) * C0 + c;
      p[0] = make_float2(s0, q0);
      p[1] = make_float2(s1, q1);
    } else {
      const float2 st0 = cstats[(long long)b * C0 + c], st1 = cstats[(long long)b * C0 + c + 1];
      const float g0 = gn_g[c] * st0.y, g1 = gn_g[c + 1] * st1.y;
      const float o0 = gn_b[c] - st0.x * g0, o1 = gn_b[c + 1] - st1.x * g1;
      for (int t = 0; t < nt; ++t) {
        float y0 = 0.f, y1 = 0.f;
#pragma unroll
        for (int j = 0; j < C0_MAXK; ++j)
          if (j < KW) {
            const float xv = xs[t * ST + j];
            y0 = fmaf(w0[j], xv, y0);
            y1 = fmaf(w1[j], xv, y1);
          }
        const float v0 = gelu_erf(fmaf(y0, g0, o0)), v1 = gelu_erf(fmaf(y1, g1, o1));
        bf16_t h0, l0, h1, l1;
        split_bf16(v0, h0, l0);
        split_bf16(v1, h1, l1);
        long long o = ((long long)b * L0 + t0 + t) * C0 + c;
        if (ser_is_il(yhi, ylo)) o = ser_il_off(o);
        *(uint32_t*)(yhi + o) = h0 | ((uint32_t)h1 << 16);
        if (ylo) *(uint32_t*)(ylo + o) = l0 | ((uint32_t)l1 << 16);
      }
    }
  }
}

__global__ void conv0_finalize_kernel(const float2* __restrict__ partial, int chunks, int C0, int L0,
                                      float2* __restrict__ cstats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (c >= C0) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < chunks; ++k) {
    const float2 p = partial[((long long)b * chunks + k) * C0 + c];
    s += p.x;
    q += p.y;
  }
  const double mean = s / L0;
  double var = q / L0 - mean * mean;
  if (var < 0.0) var = 0.0;
  cstats[(long long)b * C0 + c] = make_float2((float)mean, (float)(1.0 / sqrt(var + 1e-5)));
}

// ------------------------------------------------------------------------------------------
// Fast path for a compile-time (KW, ST): the GroupNorm statistics never touch the conv output.
// With y_t = sum_j w_j x_{ST t + j}:   sum_t y_t = w . S,   sum_t y_t^2 = w^T R w,   where
// S_j = sum_t x_{ST t + j} and R_jk = sum_t x_{ST t + j} x_{ST t + k} depend on the clip only
// (KW + KW (KW+1)/2 numbers).  Three small launches: clip statistics (wave_stats_kernel), partial S / R of the
// normalised samples over C0_SLICES workgroups per clip, then mean / rstd of every channel in double precision.
// ------------------------------------------------------------------------------------------
constexpr int C0_SLICES = 8;     // workgroups per clip in the autocorrelation pass

// partial S_j / R_jk of one slice of a clip's frames, on the normalised samples  ->  part[b][slice][NV] (double)
template <int KW, int ST>
__global__ __launch_bounds__(256) void conv0_acorr_kernel(const float* __restrict__ wave, const float2* __restrict__ wstats,
                                                          int T, int L0, double* __restrict__ part) {
  constexpr int NR = KW * (KW + 1) / 2, NV = KW + NR;
  __shared__ float red[16][NV];                      // one entry per DPP row of 16 lanes (4 waves x 4 rows)
  const int b = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x;
  const float* x = wave + (long long)b * T;
  const float2 ws = wstats[b];
  const float m = ws.x, r = ws.y;
  const int per = (L0 + C0_SLICES - 1) / C0_SLICES;
  const int t_end = min(L0, (sl + 1) * per);
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;
  for (int t = sl * per + tid; t < t_end; t += 256) {
    float xv[KW];
#pragma unroll
    for (int j = 0; j < KW; ++j) xv[j] = (x[t * ST + j] - m) * r;
    int idx = KW;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      acc[j] += xv[j];
#pragma unroll
      for (int k = j; k < KW; ++k, ++idx) acc[idx] = fmaf(xv[j], xv[k], acc[idx]);
    }
  }
  const int rowid = tid >> 4;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float v = row16_sum(acc[i]);
    if ((tid & 15) == 0) red[rowid][i] = v;
  }
  __syncthreads();
  if (tid < NV) {
    double d = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) d += (double)red[i][tid];
    part[((long long)b * C0_SLICES + sl) * NV + tid] = d;
  }
}

// mean / rstd of every channel of a clip from the summed autocorrelation (double precision)
template <int KW>
__global__ __launch_bounds__(256) void conv0_cstats_kernel(const double* __restrict__ part, const float* __restrict__ w, int C0,
                                                           int L0, float2* __restrict__ cstats) {
  constexpr int NR = KW * (KW + 1) / 2, NV = KW + NR;
  __shared__ double tot[NV];
  const int b = blockIdx.y;
  if (threadIdx.x < NV) {
    double d = 0.0;
    for (int sl = 0; sl < C0_SLICES; ++sl) d += part[((long long)b * C0_SLICES + sl) * NV + threadIdx.x];
    tot[threadIdx.x] = d;
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C0) return;
  double wj[KW];
#pragma unroll
  for (int j = 0; j < KW; ++j) wj[j] = w[c * KW + j];
  double s = 0.0, q = 0.0;
  int idx = KW;
#pragma unroll
  for (int j = 0; j < KW; ++j) {
    s += wj[j] * tot[j];
#pragma unroll
    for (int k = j; k < KW; ++k, ++idx) q += (k == j ? 1.0 : 2.0) * wj[j] * wj[k] * tot[idx];
  }
  const double mean = s / L0;
  double var = q / L0 - mean * mean;
  if (var < 0.0) var = 0.0;
  cstats[(long long)b * C0 + c] = make_float2((float)mean, (float)(1.0 / sqrt(var + 1e-5)));
}

typedef float ser_v2f __attribute__((ext_vector_type(2)));

// conv (KW packed FMAs per channel pair) + GroupNorm affine + erf-GELU + bf16 planes, channels-last.
// One thread owns a channel pair over the workgroup's frames; the samples of a frame are a
// wave-uniform LDS read; every store instruction of a wave covers 256 contiguous bytes and the
// four waves of a workgroup complete a 1 KB row.
template <int KW, int ST, bool LO>
__global__ __launch_bounds__(256) void conv0_apply_kernel(const float* __restrict__ wave, const float2* __restrict__ wstats,
                                                          const float* __restrict__ w, int T, int L0, int C0,
                                                          const float2* __restrict__ cstats, const float* __restrict__ gn_g,
                                                          const float* __restrict__ gn_b, bf16_t* __restrict__ yhi,
                                                          bf16_t* __restrict__ ylo) {
  __shared__ float xs[C0_FT * ST + KW];
  const int b = blockIdx.y, t0 = blockIdx.x * C0_FT;
  const int nt = min(C0_FT, L0 - t0);
  const int nsamp = (nt - 1) * ST + KW;
  const float2 ws = wstats[b];
  const float* x = wave + (long long)b * T + (long long)t0 * ST;
  for (int i = threadIdx.x; i < nsamp; i += 256) xs[i] = (x[i] - ws.x) * ws.y;
  __syncthreads();
  for (int cp = threadIdx.x; cp < C0 / 2; cp += 256) {
    const int c = cp * 2;
    ser_v2f wv[KW];
#pragma unroll
    for (int j = 0; j < KW; ++j) wv[j] = ser_v2f{w[c * KW + j], w[(c + 1) * KW + j]};
    const float2 st0 = cstats[(long long)b * C0 + c], st1 = cstats[(long long)b * C0 + c + 1];
    const ser_v2f g = {gn_g[c] * st0.y, gn_g[c + 1] * st1.y};
    const ser_v2f o = {gn_b[c] - st0.x * g.x, gn_b[c + 1] - st1.x * g.y};
    // interleaved planes: a frame row is 2 * C0 bf16, channel c sits at ser_il_off(c), lo 32 elements further
    const bool il = LO && ser_is_il(yhi, ylo);
    const long long e0 = il ? ((long long)b * L0 + t0) * C0 * 2 + ser_il_off(c) : ((long long)b * L0 + t0) * C0 + c;
    uint32_t* ph = (uint32_t*)(yhi + e0);
    uint32_t* pl = LO ? (uint32_t*)(ylo + e0) : nullptr;
    const int rowu = il ? C0 : C0 / 2;
#pragma unroll 4
    for (int t = 0; t < nt; ++t) {
      ser_v2f acc = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KW; ++j) acc = wv[j] * xs[t * ST + j] + acc;
      acc = acc * g + o;
      const float v0 = gelu_erf(acc.x), v1 = gelu_erf(acc.y);
      if (LO) {
        uint32_t h, l;
        split_bf16x2(v0, v1, h, l);
        ph[(long long)t * rowu] = h;
        pl[(long long)t * rowu] = l;
      } else {
        ph[(long long)t * rowu] = pack_bf16x2(v0, v1);
      }
    }
  }
}

// partial-sum area shared by the two statistics paths: generic [B][chunks][C0] float2, fast [B][C0_SLICES][<= 152] double
static size_t conv0_partial_bytes(int B, int L0, int C0) {
  const size_t generic = (size_t)B * ceil_div(L0, C0_FT) * C0 * sizeof(float2);
  const size_t fast = (size_t)B * C0_SLICES * 152 * sizeof(double);
  return ((generic > fast ? generic : fast) + 255) & ~(size_t)255;
}

size_t ser_conv0_scratch_bytes(int B, int L0, int C0) {
  return conv0_partial_bytes(B, L0, C0) + (size_t)B * C0 * sizeof(float2) + (size_t)B * sizeof(float2) + 1024;
}

int ser_launch_conv0(const float* wave, int B, int T, const float* w, const float* gn_g, const float* gn_b, int C0,
                     int KW, int ST, int L0, bf16_t* yhi, bf16_t* ylo, void* scratch, hipStream_t st) {
  SER_REQUIRE(KW <= C0_MAXK && ST <= 8 && C0 % 2 == 0, "conv0: unsupported kernel=%d stride=%d channels=%d", KW, ST, C0);
  SER_REQUIRE(L0 == (T - KW) / ST + 1 && L0 > 0, "conv0: bad output length");
  SER_REQUIRE(!ser_is_il(yhi, ylo) || C0 % SER_IL_GROUP == 0, "conv0: interleaved output planes need C0 %% %d == 0", SER_IL_GROUP);
  const int chunks = ceil_div(L0, C0_FT);
  char* p = (char*)scratch;
  float2* wstats = (float2*)p; p += (((size_t)B * sizeof(float2)) + 255) & ~(size_t)255;
  float2* partial = (float2*)p; p += conv0_partial_bytes(B, L0, C0);
  float2* cstats = (float2*)p;
  if (KW == 10 && ST == 5) {
    hipLaunchKernelGGL(wave_stats_kernel, dim3(B), dim3(1024), 0, st, wave, T, wstats);
    hipLaunchKernelGGL((conv0_acorr_kernel<10, 5>), dim3(C0_SLICES, B), dim3(256), 0, st, wave, wstats, T, L0, (double*)partial);
    hipLaunchKernelGGL((conv0_cstats_kernel<10>), dim3(ceil_div(C0, 256), B), dim3(256), 0, st, (const double*)partial, w, C0,
                       L0, cstats);
    if (ylo)
      hipLaunchKernelGGL((conv0_apply_kernel<10, 5, true>), dim3(chunks, B), dim3(256), 0, st, wave, wstats, w, T, L0, C0,
                         cstats, gn_g, gn_b, yhi, ylo);
    else
      hipLaunchKernelGGL((conv0_apply_kernel<10, 5, false>), dim3(chunks, B), dim3(256), 0, st, wave, wstats, w, T, L0, C0,
                         cstats, gn_g, gn_b, yhi, ylo);
    SER_LAUNCH_CHECK();
    return SER_OK;
  }
  hipLaunchKernelGGL(wave_stats_kernel, dim3(B), dim3(1024), 0, st, wave, T, wstats);
  hipLaunchKernelGGL(conv0_kernel<false>, dim3(chunks, B), dim3(256), 0, st, wave, wstats, w, T, L0, C0, KW, ST, partial,
                     (const float2*)nullptr, gn_g, gn_b, (bf16_t*)nullptr, (bf16_t*)nullptr);
  hipLaunchKernelGGL(conv0_finalize_kernel, dim3(ceil_div(C0, 256), B), dim3(256), 0, st, partial, chunks, C0, L0, cstats);
  hipLaunchKernelGGL(conv0_kernel<true>, dim3(chunks, B), dim3(256), 0, st, wave, wstats, w, T, L0, C0, KW, ST,
                     (float2*)nullptr, cstats, gn_g, gn_b, yhi, ylo);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// ------------------------------------------------------------------------------------------
// positional-conv slab: z[B,S,H] fp32 -> planes [B][G][S+K-1][Cg] with K/2 zero rows in front
// (hf modeling_wav2vec2.py:326-368: Conv1d(H,H,K,padding=K/2,groups=G), last frame dropped).
// With this layout the im2col row of frame t in group g is the contiguous run
// slab[b][g][t*Cg : t*Cg + K*Cg], i.e. an NT GEMM with lda = Cg.
// ------------------------------------------------------------------------------------------
// Cp = channels per slab row in memory (>= Cg; the extra columns are zeros).  With interleaved planes (lo == hi + 32) a row
// is padded to a multiple of 32 channels so that every Toeplitz window row starts on a 32-group; the padded taps meet
// zero weights.
__global__ void posconv_slab_kernel(const float* __restrict__ z, int B, int S, int H, int G, int K, int Cp,
                                    bf16_t* __restrict__ hi, bf16_t* __restrict__ lo) {
  const int Cg = H / G, R = S + K - 1;
  const bool il = ser_is_il(hi, lo);
  const long long total = (long long)B * G * R * Cp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    long long r1 = i / Cp;
    const int r = (int)(r1 % R);
    r1 /= R;
    const int g = (int)(r1 % G);
    const int b = (int)(r1 / G);
    const int t = r - K / 2;
    float v = 0.f;
    if (c < Cg && t >= 0 && t < S) v = z[((long long)b * S + t) * H + g * Cg + c];
    bf16_t h, l;
    split_bf16(v, h, l);
    const long long o = il ? ser_il_off(i) : i;
    hi[o] = h;
    if (lo) lo[o] = l;
  }
}

int ser_launch_posconv_slab(const float* z, int B, int S, int H, int G, int K, bf16_t* hi, bf16_t* lo, hipStream_t st) {
  const int Cg = H / G;
  const int Cp = ser_is_il(hi, lo) ? (Cg + SER_IL_GROUP - 1) / SER_IL_GROUP * SER_IL_GROUP : Cg;
  const long long total = (long long)B * G * (S + K - 1) * Cp;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(posconv_slab_kernel, dim3((unsigned)blocks), dim3(256), 0, st, z, B, S, H, G, K, Cp, hi, lo);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// ------------------------------------------------------------------------------------------
// XLM-R embeddings: position ids (cumsum of non-pad, hf xlm_roberta :142-155), then
// word + type + position gather and LayerNorm (:75-121).  One wave per token.
// ------------------------------------------------------------------------------------------
__global__ void xlmr_posid_kernel(const int64_t* __restrict__ ids, int B, int S, int pad_id, int max_pos,
                                  int* __restrict__ pos) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int run = 0;
  for (int s = 0; s < S; ++s) {
    const bool np = ids[(long long)b * S + s] != pad_id;
    run += np ? 1 : 0;
    int p = (np ? run : 0) + pad_id;
    pos[(long long)b * S + s] = p < max_pos ? p : max_pos - 1;
  }
}

template <int NV>
__global__ __launch_bounds__(256) void xlmr_embed_kernel(const int64_t* __restrict__ ids, const int* __restrict__ pos,
                                                         const float* __restrict__ wemb, const float* __restrict__ pemb,
                                                         const float* __restrict__ temb, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, int rows, int D,
                                                         int vocab, float* __restrict__ y, bf16_t* __restrict__ yhi,
                                                         bf16_t* __restrict__ ylo) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  long long id = ids[row];
  if (id < 0) id = 0;
  if (id >= vocab) id = vocab - 1;
  const float* wr = wemb + id * D;
  const float* pr = pemb + (long long)pos[row] * D;
  const int nchunk = D >> 2;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float4 a = *(const float4*)(wr + c * 4), t = *(const float4*)(temb + c * 4), p = *(const float4*)(pr + c * 4);
      v[i].x = (a.x + t.x) + p.x; v[i].y = (a.y + t.y) + p.y; v[i].z = (a.z + t.z) + p.z; v[i].w = (a.w + t.w) + p.w;
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float4 gm = *(const float4*)(gamma + c * 4), bt = *(const float4*)(beta + c * 4);
      float4 o;
      o.x = (v[i].x - mean) * rstd * gm.x + bt.x;
      o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
      o.z = (v[i].z - mean) * rstd * gm.z + bt.z;
      o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
      const long long off = (long long)row * D + c * 4;
      *(float4*)(y + off) = o;
      bf16_t h[4], l[4];
      split_bf16(o.x, h[0], l[0]); split_bf16(o.y, h[1], l[1]);
      split_bf16(o.z, h[2], l[2]); split_bf16(o.w, h[3], l[3]);
      const long long op = ser_is_il(yhi, ylo) ? ser_il_off(off) : off;
      if (yhi) *(uint2*)(yhi + op) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
      if (ylo) *(uint2*)(ylo + op) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
    }
  }
}

int ser_launch_xlmr_embed(const int64_t* ids, int B, int S, const float* wemb, const float* pemb, const float* temb,
                          const float* gamma, const float* beta, float eps, int D, int vocab, int max_pos, int pad_id,
                          int* pos_scratch, float* y, bf16_t* yhi, bf16_t* ylo, hipStream_t st) {
  SER_REQUIRE(D % 4 == 0 && D <= 1024, "xlmr_embed: D=%d unsupported", D);
  SER_REQUIRE(!ser_is_il(yhi, ylo) || D % SER_IL_GROUP == 0, "xlmr_embed: interleaved output planes need D %% %d == 0", SER_IL_GROUP);
  hipLaunchKernelGGL(xlmr_posid_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, st, ids, B, S, pad_id, max_pos, pos_scratch);
  const int rows = B * S;
  dim3 grid(ceil_div(rows, 4)), block(256);
  const int nv = ceil_div(D / 4, 64);
#define EMB(NVV) hipLaunchKernelGGL(xlmr_embed_kernel<NVV>, grid, block, 0, st, ids, pos_scratch, wemb, pemb, temb, gamma, beta, eps, rows, D, vocab, y, yhi, ylo)
  switch (nv) { case 1: EMB(1); break; case 2: EMB(2); break; case 3: EMB(3); break; default: EMB(4); break; }
#undef EMB
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// ---- C-ABI entries for the front-end stages (one per kernel group K2 / K5 / K7 of SURVEY section 2.2) -------------
extern "C" size_t ser_conv0_workspace_bytes(int B, int L0, int C0) { return ser_conv0_scratch_bytes(B, L0, C0); }

extern "C" int ser_conv0_gn_gelu(const float* wave, int B, int T, const float* w, const float* gn_g, const float* gn_b, int C0,
                                 int KW, int ST, uint16_t* y_hi, uint16_t* y_lo, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  SER_REQUIRE(wave && w && gn_g && gn_b && y_hi && workspace && B > 0 && T >= KW, "conv0: bad arguments");
  const int L0 = (T - KW) / ST + 1;
  SER_REQUIRE(workspace_bytes >= ser_conv0_scratch_bytes(B, L0, C0), "conv0: workspace too small");
  return ser_launch_conv0(wave, B, T, w, gn_g, gn_b, C0, KW, ST, L0, y_hi, y_lo, workspace, (hipStream_t)stream);
}

extern "C" int ser_posconv_slab(const float* z, int B, int S, int H, int G, int K, uint16_t* slab_hi, uint16_t* slab_lo,
                                void* stream) {
  SER_REQUIRE(z && slab_hi && B > 0 && S > 0 && G > 0 && H % G == 0, "posconv_slab: bad arguments");
  return ser_launch_posconv_slab(z, B, S, H, G, K, slab_hi, slab_lo, (hipStream_t)stream);
}

extern "C" int ser_xlmr_embed(const int64_t* ids, int B, int S, const float* word_emb, const float* pos_emb,
                              const float* type_emb, const float* gamma, const float* beta, float eps, int D, int vocab,
                              int max_pos, int pad_id, int* pos_scratch, float* y, uint16_t* y_hi, uint16_t* y_lo, void* stream) {
  SER_REQUIRE(ids && word_emb && pos_emb && type_emb && gamma && beta && pos_scratch && y && B > 0 && S > 0,
              "xlmr_embed: bad arguments");
  return ser_launch_xlmr_embed(ids, B, S, word_emb, pos_emb, type_emb, gamma, beta, eps, D, vocab, max_pos, pad_id, pos_scratch,
                               y, y_hi, y_lo, (hipStream_t)stream);
}
