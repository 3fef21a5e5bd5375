// Kernels of the trainable head (fp32): LayerNorm forward-with-stats / backward, activation
// backward, column sums, cross-modal attention core (forward + backward), attentive statistics
// pooling core (forward + backward), gated-fusion mix (forward + backward), the fused training
// loss (value + gradients), OpenMax rescale, and flat multi-segment AdamW.
#include "ser_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// LayerNorm forward that also emits what backward needs: z = x (+ x2), mean, rstd
// ------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int rows, int D,
                                                           float* __restrict__ y, float* __restrict__ z,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           SerDropout drop) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = D >> 2;
  // drop active: z = dropout(x) + x2 - the mask of element (row, col) is that of a dropout layer applied to x [rows, D]
  const bool dropping = drop.state != nullptr && drop.p > 0.f;
  const unsigned long long dst = dropping ? *drop.state : 0ull;
  const unsigned dth = ser_drop_thresh(drop.p);
  const float dsc = 1.0f / (1.0f - drop.p);
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      v[i] = *(const float4*)(x + (long long)row * D + c * 4);
      if (dropping) {
        const unsigned e0 = (unsigned)((long long)row * D + c * 4);
        v[i].x = __fmul_rn(v[i].x, ser_drop_mult(dst, drop.site, e0, dth, dsc));
        v[i].y = __fmul_rn(v[i].y, ser_drop_mult(dst, drop.site, e0 + 1, dth, dsc));
        v[i].z = __fmul_rn(v[i].z, ser_drop_mult(dst, drop.site, e0 + 2, dth, dsc));
        v[i].w = __fmul_rn(v[i].w, ser_drop_mult(dst, drop.site, e0 + 3, dth, dsc));
      }
      if (x2) {
        const float4 w = *(const float4*)(x2 + (long long)row * D + c * 4);
        v[i].x += w.x; v[i].y += w.y; v[i].z += w.z; v[i].w += w.w;
      }
      if (z) *(float4*)(z + (long long)row * D + c * 4) = v[i];
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
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
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
      *(float4*)(y + (long long)row * D + c * 4) = o;
    }
  }
}

// dx = rstd * (g*dy - mean_D(g*dy) - xhat * mean_D(g*dy*xhat)) (+ dx_add)
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ dx_add,
                                                        int rows, int D, float* __restrict__ dx, float* __restrict__ dx_drop,
                                                        SerDropout drop) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = D >> 2;
  const float mu = mean[row], rs = rstd[row];
  // dx_drop: the gradient of the dropped addend of z = dropout(x) + x2, i.e. dx times the forward mask (dx itself goes to x2)
  const bool dropping = dx_drop != nullptr && drop.state != nullptr && drop.p > 0.f;
  const unsigned long long dst = dropping ? *drop.state : 0ull;
  const unsigned dth = ser_drop_thresh(drop.p);
  const float dsc = 1.0f / (1.0f - drop.p);
  float4 xh[NV], dg[NV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float4 zz = *(const float4*)(z + (long long)row * D + c * 4);
      const float4 d = *(const float4*)(dy + (long long)row * D + c * 4);
      const float4 gm = *(const float4*)(gamma + c * 4);
      xh[i] = make_float4((zz.x - mu) * rs, (zz.y - mu) * rs, (zz.z - mu) * rs, (zz.w - mu) * rs);
      dg[i] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
      s1 += (dg[i].x + dg[i].y) + (dg[i].z + dg[i].w);
      s2 += (dg[i].x * xh[i].x + dg[i].y * xh[i].y) + (dg[i].z * xh[i].z + dg[i].w * xh[i].w);
    }
  }
  const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float4 o;
      o.x = rs * (dg[i].x - m1 - xh[i].x * m2);
      o.y = rs * (dg[i].y - m1 - xh[i].y * m2);
      o.z = rs * (dg[i].z - m1 - xh[i].z * m2);
      o.w = rs * (dg[i].w - m1 - xh[i].w * m2);
      if (dx_add) {
        const float4 a = *(const float4*)(dx_add + (long long)row * D + c * 4);
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      }
      *(float4*)(dx + (long long)row * D + c * 4) = o;
      if (dx_drop) {
        if (dropping) {
          const unsigned e0 = (unsigned)((long long)row * D + c * 4);
          o.x *= ser_drop_mult(dst, drop.site, e0, dth, dsc);
          o.y *= ser_drop_mult(dst, drop.site, e0 + 1, dth, dsc);
          o.z *= ser_drop_mult(dst, drop.site, e0 + 2, dth, dsc);
          o.w *= ser_drop_mult(dst, drop.site, e0 + 3, dth, dsc);
        }
        *(float4*)(dx_drop + (long long)row * D + c * 4) = o;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------
// Two chained LayerNorms of a classifier block (ref classifier.py:209-210: x1 = LN_out(x); u = LN_in(x1))
// in one launch, and their joint backward including all four parameter gradients.
// ------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void ln2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g1,
                                                      const float* __restrict__ b1, const float* __restrict__ g2,
                                                      const float* __restrict__ b2, float eps, int rows, int D,
                                                      float* __restrict__ y1, float* __restrict__ y2,
                                                      float* __restrict__ stats /* [4][rows]: mean1,rstd1,mean2,rstd2 */) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = D >> 2;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < nchunk ? *(const float4*)(x + (long long)row * D + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const float* gm_p = pass == 0 ? g1 : g2;
    const float* bt_p = pass == 0 ? b1 : b2;
    float* yo = pass == 0 ? y1 : y2;
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
    if (lane == 0) {
      stats[(2 * pass) * rows + row] = mean;
      stats[(2 * pass + 1) * rows + row] = rstd;
    }
    s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const float4 gm = *(const float4*)(gm_p + c * 4), bt = *(const float4*)(bt_p + c * 4);
        float4 o;
        o.x = (v[i].x - mean) * rstd * gm.x + bt.x;
        o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
        o.z = (v[i].z - mean) * rstd * gm.z + bt.z;
        o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
        *(float4*)(yo + (long long)row * D + c * 4) = o;
        v[i] = o;
        s += (o.x + o.y) + (o.z + o.w);
      }
    }
  }
}

// du = grad wrt y2, dres = extra grad arriving at y1 (the residual branch).  dx = LN1'(LN2'(du) + dres).
// One workgroup walks all rows (rows = batch, small): wave w takes rows w, w+4, ...; the parameter gradients
// are accumulated per lane over the wave's rows and reduced across the four waves through LDS.
constexpr int LN2_WAVES = 8;
template <int NV>
__global__ __launch_bounds__(LN2_WAVES * 64) void ln2_bwd_kernel(const float* __restrict__ du, const float* __restrict__ dres,
                                                      const float* __restrict__ x, const float* __restrict__ y1,
                                                      const float* __restrict__ stats, const float* __restrict__ g1,
                                                      const float* __restrict__ g2, int rows, int D,
                                                      float* __restrict__ dx, float* __restrict__ dg1,
                                                      float* __restrict__ db1, float* __restrict__ dg2,
                                                      float* __restrict__ db2, int accumulate) {
  // 16 KB: small on purpose — this kernel runs beside the encoder GEMMs of the next batch, which leave little LDS
  // free per CU; a workgroup that asks for more waits for a GEMM workgroup to retire
  __shared__ float4 red[LN2_WAVES][NV * 64];       // [wave][chunk], reused for the four parameters in turn
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nchunk = D >> 2;
  float4 ag1[NV], ab1[NV], ag2[NV], ab2[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) ag1[i] = ab1[i] = ag2[i] = ab2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int row = w; row < rows; row += LN2_WAVES) {
    const float m1 = stats[row], r1 = stats[rows + row], m2 = stats[2 * rows + row], r2 = stats[3 * rows + row];
    float4 xh[NV], dgv[NV];
    float s1 = 0.f, s2 = 0.f;
    // ---- LN2 backward: input y1, upstream du
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const float4 zz = *(const float4*)(y1 + (long long)row * D + c * 4);
        const float4 d = *(const float4*)(du + (long long)row * D + c * 4);
        const float4 gm = *(const float4*)(g2 + c * 4);
        xh[i] = make_float4((zz.x - m2) * r2, (zz.y - m2) * r2, (zz.z - m2) * r2, (zz.w - m2) * r2);
        dgv[i] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
        ag2[i].x += d.x * xh[i].x; ag2[i].y += d.y * xh[i].y; ag2[i].z += d.z * xh[i].z; ag2[i].w += d.w * xh[i].w;
        ab2[i].x += d.x; ab2[i].y += d.y; ab2[i].z += d.z; ab2[i].w += d.w;
        s1 += (dgv[i].x + dgv[i].y) + (dgv[i].z + dgv[i].w);
        s2 += (dgv[i].x * xh[i].x + dgv[i].y * xh[i].y) + (dgv[i].z * xh[i].z + dgv[i].w * xh[i].w);
      }
    }
    float a1 = wave_sum(s1) / (float)D, a2 = wave_sum(s2) / (float)D;
    float4 d1[NV];   // gradient arriving at y1 (= output of LN1)
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        d1[i].x = r2 * (dgv[i].x - a1 - xh[i].x * a2);
        d1[i].y = r2 * (dgv[i].y - a1 - xh[i].y * a2);
        d1[i].z = r2 * (dgv[i].z - a1 - xh[i].z * a2);
        d1[i].w = r2 * (dgv[i].w - a1 - xh[i].w * a2);
        if (dres) {
          const float4 e = *(const float4*)(dres + (long long)row * D + c * 4);
          d1[i].x += e.x; d1[i].y += e.y; d1[i].z += e.z; d1[i].w += e.w;
        }
        // ---- LN1 backward: input x, upstream d1
        const float4 zz = *(const float4*)(x + (long long)row * D + c * 4);
        const float4 gm = *(const float4*)(g1 + c * 4);
        xh[i] = make_float4((zz.x - m1) * r1, (zz.y - m1) * r1, (zz.z - m1) * r1, (zz.w - m1) * r1);
        ag1[i].x += d1[i].x * xh[i].x; ag1[i].y += d1[i].y * xh[i].y; ag1[i].z += d1[i].z * xh[i].z; ag1[i].w += d1[i].w * xh[i].w;
        ab1[i].x += d1[i].x; ab1[i].y += d1[i].y; ab1[i].z += d1[i].z; ab1[i].w += d1[i].w;
        dgv[i] = make_float4(d1[i].x * gm.x, d1[i].y * gm.y, d1[i].z * gm.z, d1[i].w * gm.w);
        s1 += (dgv[i].x + dgv[i].y) + (dgv[i].z + dgv[i].w);
        s2 += (dgv[i].x * xh[i].x + dgv[i].y * xh[i].y) + (dgv[i].z * xh[i].z + dgv[i].w * xh[i].w);
      }
    }
    a1 = wave_sum(s1) / (float)D; a2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float4 o;
        o.x = r1 * (dgv[i].x - a1 - xh[i].x * a2);
        o.y = r1 * (dgv[i].y - a1 - xh[i].y * a2);
        o.z = r1 * (dgv[i].z - a1 - xh[i].z * a2);
        o.w = r1 * (dgv[i].w - a1 - xh[i].w * a2);
        *(float4*)(dx + (long long)row * D + c * 4) = o;
      }
    }
  }
#pragma unroll
  for (int which = 0; which < 4; ++which) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      red[w][lane + 64 * i] = which == 0 ? ag1[i] : (which == 1 ? ab1[i] : (which == 2 ? ag2[i] : ab2[i]));
    __syncthreads();
    float* out = which == 0 ? dg1 : (which == 1 ? db1 : (which == 2 ? dg2 : db2));
    // waves 0..NV-1 each finish one 64-chunk slice of this parameter
    if (w < NV) {
      const int c = lane + 64 * w;
      if (c < nchunk) {
        float4 t = red[0][c];
#pragma unroll
        for (int ww = 1; ww < LN2_WAVES; ++ww) { const float4 u = red[ww][c]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        float4* dst = (float4*)(out + c * 4);
        if (accumulate) { const float4 old = *dst; t.x += old.x; t.y += old.y; t.z += old.z; t.w += old.w; }
        *dst = t;
      }
    }
    __syncthreads();
  }
}

// dgamma[d] (+)= sum_rows dy*xhat ; dbeta[d] (+)= sum_rows dy.
// grid = (D/64 column tiles, row slices).  With one slice the result is written directly; with several, each
// workgroup writes a partial to `part[slice][2][D]` and ln_bwd_param_reduce_kernel adds them in slice order
// (deterministic; many-token LayerNorms would otherwise run on D/64 workgroups only).
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           int rows, int D, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int accumulate,
                                                           float* __restrict__ part) {
  __shared__ float sg[4][64], sb[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  const int nsl = gridDim.y, per = (rows + nsl - 1) / nsl;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float ag = 0.f, ab = 0.f;
  if (col < D)
    for (int r = r0 + rg; r < r1; r += 4) {
      const float d = dy[(long long)r * D + col];
      ag = fmaf(d, (z[(long long)r * D + col] - mean[r]) * rstd[r], ag);
      ab += d;
    }
  sg[rg][threadIdx.x & 63] = ag;
  sb[rg][threadIdx.x & 63] = ab;
  __syncthreads();
  if (rg == 0 && col < D) {
    const int c = threadIdx.x;
    const float g = (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]);
    const float b = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
    if (part) {
      part[((long long)blockIdx.y * 2) * D + col] = g;
      part[((long long)blockIdx.y * 2 + 1) * D + col] = b;
    } else {
      dgamma[col] = accumulate ? dgamma[col] + g : g;
      dbeta[col] = accumulate ? dbeta[col] + b : b;
    }
  }
}

__global__ void ln_bwd_param_reduce_kernel(const float* __restrict__ part, int nsl, int D, float* __restrict__ dgamma,
                                           float* __restrict__ dbeta, int accumulate) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= D) return;
  float g = 0.f, b = 0.f;
  for (int s = 0; s < nsl; ++s) {
    g += part[((long long)s * 2) * D + col];
    b += part[((long long)s * 2 + 1) * D + col];
  }
  dgamma[col] = accumulate ? dgamma[col] + g : g;
  dbeta[col] = accumulate ? dbeta[col] + b : b;
}

// out[n] (+)= sum_m x[m*ld + n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int M, int N, int ld,
                                                     float* __restrict__ out, int accumulate) {
  __shared__ float sh[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  float a = 0.f;
  if (col < N)
    for (int r = rg; r < M; r += 4) a += x[(long long)r * ld + col];
  sh[rg][threadIdx.x & 63] = a;
  __syncthreads();
  if (rg == 0 && col < N) {
    const int c = threadIdx.x;
    const float s = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
    out[col] = accumulate ? out[col] + s : s;
  }
}

// the same over many rows in two deterministic stages: part[c][n] = sum of row chunk c (grid.y chunks), then colsum_kernel over
// the chunk sums (a fixed summation order: no atomics)
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ x, int M, int N, int ld, float* __restrict__ part) {
  __shared__ float sh[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  const int per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(M, r0 + per);
  float a = 0.f;
  if (col < N)
    for (int r = r0 + rg; r < r1; r += 4) a += x[(long long)r * ld + col];
  sh[rg][threadIdx.x & 63] = a;
  __syncthreads();
  if (rg == 0 && col < N) {
    const int c = threadIdx.x;
    part[(long long)blockIdx.y * N + col] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
  }
}

// dx = dy * act'(y) given the activation OUTPUT y
// with `drop`: y was produced as dropout(act(x)); dx = dy * mask / (1 - p) * act'(y)  (for ReLU, y > 0 <=> active and kept)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int act, long long n,
                               float* __restrict__ dx, SerDropout drop) {
  const bool dropping = drop.state != nullptr && drop.p > 0.f;
  const unsigned long long dst = dropping ? *drop.state : 0ull;
  const unsigned dth = ser_drop_thresh(drop.p);
  const float dsc = 1.0f / (1.0f - drop.p);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float yy = y[i];
    float d;
    if (act == SER_ACT_RELU) d = yy > 0.f ? 1.f : 0.f;
    else if (act == SER_ACT_TANH) d = 1.f - yy * yy;
    else if (act == SER_ACT_SIGMOID) d = yy * (1.f - yy);
    else d = 1.f;
    if (dropping) d *= ser_drop_mult(dst, drop.site, (unsigned)i, dth, dsc);
    dx[i] = dy[i] * d;
  }
}

// y = dropout(act(x))
__global__ void act_fwd_kernel(const float* __restrict__ x, int act, long long n, float* __restrict__ y, SerDropout drop) {
  const bool dropping = drop.state != nullptr && drop.p > 0.f;
  const unsigned long long dst = dropping ? *drop.state : 0ull;
  const unsigned dth = ser_drop_thresh(drop.p);
  const float dsc = 1.0f / (1.0f - drop.p);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    float o;
    if (act == SER_ACT_RELU) o = fmaxf(v, 0.f);
    else if (act == SER_ACT_TANH) o = tanhf(v);
    else if (act == SER_ACT_SIGMOID) o = 1.f / (1.f + expf(-v));
    else if (act == SER_ACT_GELU) o = gelu_erf(v);
    else o = v;
    if (dropping) o *= ser_drop_mult(dst, drop.site, (unsigned)i, dth, dsc);
    y[i] = o;
  }
}

// x *= s[0]  (device scalar: keeps autograd's upstream gradient on the device, no host sync)
__global__ void scale_dev_kernel(float* __restrict__ x, const float* __restrict__ s, long long n) {
  const float f = s[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= f;
}

// y = a*x + b*y
__global__ void axpby_kernel(const float* __restrict__ x, float a, float b, long long n, float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = b == 0.f ? a * x[i] : fmaf(a, x[i], b * y[i]);
}

// ------------------------------------------------------------------------------------------
// cross-modal attention core (torch nn.functional.multi_head_attention_forward, eval):
// P = softmax(q k^T * hd^-1/2 + key mask), ctx = P v.  One wave per (batch, head, query row).
// q,k,v are [B,S,E] with per-tensor row strides so they can be column blocks of a fused buffer.
// ------------------------------------------------------------------------------------------
constexpr int XA_MAXK = 1536;   // keys per row held in LDS (30 s of audio = 1499 frames)

// attention-probability dropout (nn.MultiheadAttention(dropout=p), training mode): ctx uses P * m / (1 - p); the stored P
// stays the softmax output (its backward needs it) and the mask is regenerated from the element index in backward
struct XDrop {
  bool on;
  unsigned long long st;
  unsigned site, thresh;
  float scale;
};
SER_DEVFN XDrop xdrop_init(const SerDropout& d) {
  XDrop x;
  x.on = d.state != nullptr && d.p > 0.f;
  x.st = x.on ? *d.state : 0ull;
  x.site = d.site; x.thresh = ser_drop_thresh(d.p); x.scale = 1.0f / (1.0f - d.p);
  return x;
}
SER_DEVFN float xdrop_mult(const XDrop& x, long long idx) {
  return x.on ? ser_drop_mult(x.st, x.site, (unsigned)idx, x.thresh, x.scale) : 1.0f;
}

__global__ __launch_bounds__(256) void xattn_fwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                        int ldk, const float* __restrict__ v, int ldv,
                                                        const float* __restrict__ kmask, int B, int Sq, int Sk, int heads,
                                                        int hd, float* __restrict__ P, float* __restrict__ ctx, int ldc,
                                                        SerDropout drop, float* __restrict__ Pd) {
  __shared__ float ps[4][XA_MAXK];
  const XDrop xd = xdrop_init(drop);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + w;   // over B*heads*Sq
  if (row >= (long long)B * heads * Sq) return;
  const int i = (int)(row % Sq);
  const int h = (int)((row / Sq) % heads);
  const int b = (int)(row / ((long long)Sq * heads));
  const float scale = 1.0f / sqrtf((float)hd);
  const float* qr = q + ((long long)b * Sq + i) * ldq + h * hd;
  float mx = -INFINITY;
  for (int j = lane; j < Sk; j += 64) {
    float s = -INFINITY;
    if (!kmask || kmask[(long long)b * Sk + j] != 0.f) {
      const float* kr = k + ((long long)b * Sk + j) * ldk + h * hd;
      float acc = 0.f;
      for (int d = 0; d < hd; ++d) acc = fmaf(qr[d] * scale, kr[d], acc);
      s = acc;
    }
    ps[w][j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  const float mu = mx == -INFINITY ? 0.f : mx;
  float sum = 0.f;
  for (int j = lane; j < Sk; j += 64) {
    const float e = expf(ps[w][j] - mu);
    ps[w][j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  float* Pr = P + (((long long)b * heads + h) * Sq + i) * Sk;
  const long long prow0 = (((long long)b * heads + h) * Sq + i) * Sk;
  for (int j = lane; j < Sk; j += 64) {
    const float p = ps[w][j] * inv;
    ps[w][j] = p * xdrop_mult(xd, prow0 + j);
    Pr[j] = p;
    if (Pd) Pd[prow0 + j] = ps[w][j];                 // dropped probabilities, for dv in backward
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane < hd) {
    float acc = 0.f;
    for (int j = 0; j < Sk; ++j) acc = fmaf(ps[w][j], v[((long long)b * Sk + j) * ldv + h * hd + lane], acc);
    ctx[((long long)b * Sq + i) * ldc + h * hd + lane] = acc;
  }
}

// per query row: dP = dctx.v, dS = P*(dP - sum_j P dP) -> dS (overwrites dSbuf), dq = scale * dS k
__global__ __launch_bounds__(256) void xattn_bwd_q_kernel(const float* __restrict__ dctx, int ldc,
                                                          const float* __restrict__ k, int ldk,
                                                          const float* __restrict__ v, int ldv,
                                                          const float* __restrict__ P, int B, int Sq, int Sk, int heads,
                                                          int hd, float* __restrict__ dS, float* __restrict__ dq, int ldq,
                                                          SerDropout drop) {
  __shared__ float ds[4][XA_MAXK];
  const XDrop xd = xdrop_init(drop);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + w;
  if (row >= (long long)B * heads * Sq) return;
  const int i = (int)(row % Sq);
  const int h = (int)((row / Sq) % heads);
  const int b = (int)(row / ((long long)Sq * heads));
  const float scale = 1.0f / sqrtf((float)hd);
  const float* dc = dctx + ((long long)b * Sq + i) * ldc + h * hd;
  const float* Pr = P + (((long long)b * heads + h) * Sq + i) * Sk;
  float dot = 0.f;
  for (int j = lane; j < Sk; j += 64) {
    const float* vr = v + ((long long)b * Sk + j) * ldv + h * hd;
    float acc = 0.f;
    for (int d = 0; d < hd; ++d) acc = fmaf(dc[d], vr[d], acc);
    acc *= xdrop_mult(xd, (((long long)b * heads + h) * Sq + i) * Sk + j);     // gradient at the un-dropped probability
    ds[w][j] = acc;
    dot = fmaf(Pr[j], acc, dot);
  }
  dot = wave_sum(dot);
  float* dSr = dS + (((long long)b * heads + h) * Sq + i) * Sk;
  for (int j = lane; j < Sk; j += 64) {
    const float g = Pr[j] * (ds[w][j] - dot);
    ds[w][j] = g;
    dSr[j] = g;
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane < hd) {
    float acc = 0.f;
    for (int j = 0; j < Sk; ++j) acc = fmaf(ds[w][j], k[((long long)b * Sk + j) * ldk + h * hd + lane], acc);
    dq[((long long)b * Sq + i) * ldq + h * hd + lane] = acc * scale;
  }
}

// per key row: dk_j = scale * sum_i dS_ij q_i ; dv_j = sum_i P_ij dctx_i
__global__ __launch_bounds__(256) void xattn_bwd_kv_kernel(const float* __restrict__ dctx, int ldc,
                                                           const float* __restrict__ q, int ldq,
                                                           const float* __restrict__ P, const float* __restrict__ dS,
                                                           int B, int Sq, int Sk, int heads, int hd,
                                                           float* __restrict__ dk, int ldk, float* __restrict__ dv, int ldv) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + w;   // over B*heads*Sk
  if (row >= (long long)B * heads * Sk) return;
  const int j = (int)(row % Sk);
  const int h = (int)((row / Sk) % heads);
  const int b = (int)(row / ((long long)Sk * heads));
  const float scale = 1.0f / sqrtf((float)hd);
  const float* Pc = P + ((long long)b * heads + h) * Sq * Sk + j;
  const float* Sc = dS + ((long long)b * heads + h) * Sq * Sk + j;
  // lanes [0,hd) accumulate dk, lanes [32,32+hd) accumulate dv when hd <= 32; otherwise two passes
  if (hd <= 32) {
    const bool isv = lane >= 32;
    const int d = lane & 31;
    if (d < hd) {
      float acc = 0.f;
      for (int i = 0; i < Sq; ++i) {
        const float wgt = isv ? Pc[(long long)i * Sk] : Sc[(long long)i * Sk];
        const float* src = isv ? dctx + ((long long)b * Sq + i) * ldc : q + ((long long)b * Sq + i) * ldq;
        acc = fmaf(wgt, src[h * hd + d], acc);
      }
      if (isv) dv[((long long)b * Sk + j) * ldv + h * hd + d] = acc;
      else dk[((long long)b * Sk + j) * ldk + h * hd + d] = acc * scale;
    }
  } else if (lane < hd) {
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < Sq; ++i) {
      ak = fmaf(Sc[(long long)i * Sk], q[((long long)b * Sq + i) * ldq + h * hd + lane], ak);
      av = fmaf(Pc[(long long)i * Sk], dctx[((long long)b * Sq + i) * ldc + h * hd + lane], av);
    }
    dk[((long long)b * Sk + j) * ldk + h * hd + lane] = ak * scale;
    dv[((long long)b * Sk + j) * ldv + h * hd + lane] = av;
  }
}


// ------------------------------------------------------------------------------------------
// LDS-staged cross-attention core for S <= 256 on both sides (the 1-5 s clips of the path): K^T / V (forward) and
// V^T / K / Q / dctx (backward) of one (clip, head) are staged once per workgroup and every global access is
// coalesced.  Same math and same saved tensors (P, dS) as the generic kernels above, which remain the fallback
// for longer sequences.
// ------------------------------------------------------------------------------------------
constexpr int XF_MAXS = 256, XF_LD = XF_MAXS + 1;
// head dimension: 32 (the cross-modal attention's 256 / 8) or 64 (the encoders' self-attention in the fine-tuning path)
#define XF_HD HD

// grid (ceil(Sq/16), heads, B); 4 waves x 4 query rows
template <int HD>
__global__ __launch_bounds__(256) void xattn_fwd_fast_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                             int ldk, const float* __restrict__ v, int ldv,
                                                             const float* __restrict__ kmask, int Sq, int Sk, int heads,
                                                             float* __restrict__ P, float* __restrict__ ctx, int ldc,
                                                             SerDropout drop, float* __restrict__ Pd) {
  const XDrop xd = xdrop_init(drop);
  __shared__ float Kt[XF_HD][XF_LD];       // K^T: [d][key]
  __shared__ float Vs[XF_MAXS][XF_HD];     // V:   [key][d]
  __shared__ float prow[4][XF_MAXS];
  __shared__ float qrow[4][XF_HD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * 16;
  const float scale = 1.0f / sqrtf((float)XF_HD);
  for (int idx = tid; idx < Sk * XF_HD; idx += 256) {
    const int j = idx / XF_HD, d = idx % XF_HD;
    Kt[d][j] = k[((long long)b * Sk + j) * ldk + h * XF_HD + d];
    Vs[j][d] = v[((long long)b * Sk + j) * ldv + h * XF_HD + d];
  }
  __syncthreads();
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + w * 4 + r;
    if (i >= Sq) break;                       // wave-uniform
    if (lane < XF_HD) qrow[w][lane] = q[((long long)b * Sq + i) * ldq + h * XF_HD + lane] * scale;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float sc[4];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = lane + 64 * u;
      float s = -INFINITY;
      if (j < Sk && (!kmask || kmask[(long long)b * Sk + j] != 0.f)) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < XF_HD; ++d) acc = fmaf(qrow[w][d], Kt[d][j], acc);
        s = acc;
      }
      sc[u] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    const float mu = mx == -INFINITY ? 0.f : mx;
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) { sc[u] = expf(sc[u] - mu); sum += sc[u]; }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    float* Pr = P + (((long long)b * heads + h) * Sq + i) * Sk;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = lane + 64 * u;
      if (j < Sk) {
        const float p = sc[u] * inv;
        prow[w][j] = p * xdrop_mult(xd, (((long long)b * heads + h) * Sq + i) * Sk + j);
        Pr[j] = p;
        if (Pd) Pd[(((long long)b * heads + h) * Sq + i) * Sk + j] = prow[w][j];
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ctx[d] = sum_j p_j V[j][d]: with 32-wide heads lanes 0-31 take even keys, 32-63 odd keys; 64-wide: one lane per d
    constexpr int NPAR = 64 / XF_HD;
    const int d = lane % XF_HD, par = lane / XF_HD;
    float acc = 0.f;
    for (int j = par; j < Sk; j += NPAR) acc = fmaf(prow[w][j], Vs[j][d], acc);
    if (NPAR == 2) acc += __shfl_xor(acc, 32, 64);
    if (lane < XF_HD) ctx[((long long)b * Sq + i) * ldc + h * XF_HD + d] = acc;
    __builtin_amdgcn_wave_barrier();
  }
}

// per query row: dP = dctx.V^T, dS = P (dP - sum P dP), dq = scale * dS K
template <int HD>
__global__ __launch_bounds__(256) void xattn_bwd_q_fast_kernel(const float* __restrict__ dctx, int ldc, const float* __restrict__ k,
                                                               int ldk, const float* __restrict__ v, int ldv,
                                                               const float* __restrict__ P, int Sq, int Sk, int heads,
                                                               float* __restrict__ dS, float* __restrict__ dq, int ldq,
                                                               SerDropout drop) {
  const XDrop xd = xdrop_init(drop);
  __shared__ float Vt[XF_HD][XF_LD];       // V^T: [d][key]
  __shared__ float Ks[XF_MAXS][XF_HD];     // K:   [key][d]
  __shared__ float srow[4][XF_MAXS];
  __shared__ float drow[4][XF_HD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * 16;
  const float scale = 1.0f / sqrtf((float)XF_HD);
  for (int idx = tid; idx < Sk * XF_HD; idx += 256) {
    const int j = idx / XF_HD, d = idx % XF_HD;
    Vt[d][j] = v[((long long)b * Sk + j) * ldv + h * XF_HD + d];
    Ks[j][d] = k[((long long)b * Sk + j) * ldk + h * XF_HD + d];
  }
  __syncthreads();
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + w * 4 + r;
    if (i >= Sq) break;
    if (lane < XF_HD) drow[w][lane] = dctx[((long long)b * Sq + i) * ldc + h * XF_HD + lane];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float* Pr = P + (((long long)b * heads + h) * Sq + i) * Sk;
    float dp[4], pp[4];
    float dot = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = lane + 64 * u;
      dp[u] = 0.f; pp[u] = 0.f;
      if (j < Sk) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < XF_HD; ++d) acc = fmaf(drow[w][d], Vt[d][j], acc);
        acc *= xdrop_mult(xd, (((long long)b * heads + h) * Sq + i) * Sk + j);
        dp[u] = acc; pp[u] = Pr[j];
        dot = fmaf(pp[u], acc, dot);
      }
    }
    dot = wave_sum(dot);
    float* dSr = dS + (((long long)b * heads + h) * Sq + i) * Sk;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = lane + 64 * u;
      if (j < Sk) { const float gg = pp[u] * (dp[u] - dot); srow[w][j] = gg; dSr[j] = gg; }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int NPAR = 64 / XF_HD;
    const int d = lane % XF_HD, par = lane / XF_HD;
    float acc = 0.f;
    for (int j = par; j < Sk; j += NPAR) acc = fmaf(srow[w][j], Ks[j][d], acc);
    if (NPAR == 2) acc += __shfl_xor(acc, 32, 64);
    if (lane < XF_HD) dq[((long long)b * Sq + i) * ldq + h * XF_HD + d] = acc * scale;
    __builtin_amdgcn_wave_barrier();
  }
}

// per key: dk_j = scale * sum_i dS_ij q_i ; dv_j = sum_i P_ij dctx_i.  grid (ceil(Sk/64), heads, B), one key per lane
// of wave 0..: 256 threads = 4 waves, wave w handles the d-range [w HD/4, (w+1) HD/4) for all 64 keys of the tile.
template <int HD>
__global__ __launch_bounds__(256) void xattn_bwd_kv_fast_kernel(const float* __restrict__ dctx, int ldc, const float* __restrict__ q,
                                                                int ldq, const float* __restrict__ P, const float* __restrict__ dS,
                                                                int Sq, int Sk, int heads, float* __restrict__ dk, int ldk,
                                                                float* __restrict__ dv, int ldv) {
  __shared__ float Qs[XF_MAXS][XF_HD];
  __shared__ float Ds[XF_MAXS][XF_HD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, j = blockIdx.x * 64 + lane;
  const float scale = 1.0f / sqrtf((float)XF_HD);
  for (int idx = tid; idx < Sq * XF_HD; idx += 256) {
    const int i = idx / XF_HD, d = idx % XF_HD;
    Qs[i][d] = q[((long long)b * Sq + i) * ldq + h * XF_HD + d];
    Ds[i][d] = dctx[((long long)b * Sq + i) * ldc + h * XF_HD + d];
  }
  __syncthreads();
  const float* Pc = P + ((long long)b * heads + h) * Sq * Sk;
  const float* Sc = dS + ((long long)b * heads + h) * Sq * Sk;
  constexpr int DW = XF_HD / 4;          // head dims per wave
  float ak[DW], av[DW];
#pragma unroll
  for (int e = 0; e < DW; ++e) ak[e] = av[e] = 0.f;
  const int jj = j < Sk ? j : Sk - 1;
  for (int i = 0; i < Sq; ++i) {
    const float p = Pc[(long long)i * Sk + jj], s = Sc[(long long)i * Sk + jj];     // coalesced over the 64 keys
#pragma unroll
    for (int e = 0; e < DW; ++e) {
      ak[e] = fmaf(s, Qs[i][w * DW + e], ak[e]);       // LDS broadcast reads
      av[e] = fmaf(p, Ds[i][w * DW + e], av[e]);
    }
  }
  if (j < Sk) {
    float* dkr = dk + ((long long)b * Sk + j) * ldk + h * XF_HD + w * DW;
    float* dvr = dv + ((long long)b * Sk + j) * ldv + h * XF_HD + w * DW;
#pragma unroll
    for (int e = 0; e < DW; ++e) { dkr[e] = ak[e] * scale; dvr[e] = av[e]; }
  }
}
#undef XF_HD

// ------------------------------------------------------------------------------------------
// attentive statistics pooling core (ref src/models/pooling.py:21-28)
// ------------------------------------------------------------------------------------------
constexpr int POOL_MAXS = 2048;

__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ logits,
                                                       const float* __restrict__ mask, int S, int D,
                                                       float* __restrict__ alpha, float* __restrict__ out) {
  __shared__ float al[POOL_MAXS];
  __shared__ float red[4];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float mx = -INFINITY;
  for (int s = tid; s < S; s += 256) {
    float l = logits[(long long)b * S + s];
    if (mask && mask[(long long)b * S + s] == 0.f) l = -INFINITY;
    al[s] = l;
    mx = fmaxf(mx, l);
  }
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float mu = mx == -INFINITY ? 0.f : mx;
  __syncthreads();
  float sum = 0.f;
  for (int s = tid; s < S; s += 256) {
    const float e = expf(al[s] - mu);
    al[s] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[w] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  for (int s = tid; s < S; s += 256) {
    const float a = al[s] * inv;
    al[s] = a;
    if (blockIdx.x == 0) alpha[(long long)b * S + s] = a;
  }
  __syncthreads();
  const int d = blockIdx.x * 256 + tid;
  if (d < D) {
    const float* xb = x + (long long)b * S * D + d;
    float mean = 0.f;
    for (int s = 0; s < S; ++s) mean = fmaf(al[s], xb[(long long)s * D], mean);
    float var = 0.f;
    for (int s = 0; s < S; ++s) {
      const float c = xb[(long long)s * D] - mean;
      var = fmaf(al[s], c * c, var);
    }
    out[(long long)b * 2 * D + d] = mean;
    out[(long long)b * 2 * D + D + d] = sqrtf(var + 1e-6f);
  }
}

// one wave per (b,s): dx row and d(alpha)_s
__global__ __launch_bounds__(256) void pool_bwd_x_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                         const float* __restrict__ alpha, const float* __restrict__ out,
                                                         int B, int S, int D, float* __restrict__ dx,
                                                         float* __restrict__ dalpha) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)B * S) return;
  const int b = (int)(row / S);
  const float a = alpha[row];
  const float* xr = x + row * D;
  const float* dm = dout + (long long)b * 2 * D;
  const float* mo = out + (long long)b * 2 * D;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float dvar = dm[D + d] / (2.0f * mo[D + d]);
    const float c = xr[d] - mo[d];
    acc += dm[d] * xr[d] + dvar * c * c;
    dx[row * D + d] = a * (dm[d] + 2.0f * dvar * c);
  }
  acc = wave_sum(acc);
  if (lane == 0) dalpha[row] = acc;
}

// dlogit_s = alpha_s (dalpha_s - sum_s' alpha_s' dalpha_s')
__global__ __launch_bounds__(256) void pool_bwd_logit_kernel(const float* __restrict__ alpha,
                                                             const float* __restrict__ dalpha, int S,
                                                             float* __restrict__ dlogits) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float dot = 0.f;
  for (int s = tid; s < S; s += 256) dot = fmaf(alpha[(long long)b * S + s], dalpha[(long long)b * S + s], dot);
  dot = wave_sum(dot);
  if ((tid & 63) == 0) red[tid >> 6] = dot;
  __syncthreads();
  dot = (red[0] + red[1]) + (red[2] + red[3]);
  for (int s = tid; s < S; s += 256)
    dlogits[(long long)b * S + s] = alpha[(long long)b * S + s] * (dalpha[(long long)b * S + s] - dot);
}

// ------------------------------------------------------------------------------------------
// gated fusion mix (ref src/models/fusion.py:21-25): out = (wa a + wt t) / (wa + wt + 1e-8)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fusion_mix_fwd_kernel(const float* __restrict__ a, const float* __restrict__ t,
                                                             const float* __restrict__ ga, const float* __restrict__ gt,
                                                             int P, float* __restrict__ out) {
  const int b = blockIdx.x;
  const float wa = 1.f / (1.f + expf(-ga[b])), wt = 1.f / (1.f + expf(-gt[b]));
  const float ws = wa + wt + 1e-8f;
  const float na = wa / ws, nt = wt / ws;
  for (int p = threadIdx.x; p < P; p += 256)
    out[(long long)b * P + p] = na * a[(long long)b * P + p] + nt * t[(long long)b * P + p];
}

__global__ __launch_bounds__(256) void fusion_mix_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                             const float* __restrict__ t, const float* __restrict__ ga,
                                                             const float* __restrict__ gt, int P, float* __restrict__ da,
                                                             float* __restrict__ dt, float* __restrict__ dga,
                                                             float* __restrict__ dgt) {
  __shared__ float red[2][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float wa = 1.f / (1.f + expf(-ga[b])), wt = 1.f / (1.f + expf(-gt[b]));
  const float ws = wa + wt + 1e-8f;
  const float na = wa / ws, nt = wt / ws;
  float sa = 0.f, st = 0.f;
  for (int p = tid; p < P; p += 256) {
    const float d = dout[(long long)b * P + p];
    sa = fmaf(d, a[(long long)b * P + p], sa);
    st = fmaf(d, t[(long long)b * P + p], st);
    da[(long long)b * P + p] = na * d;
    dt[(long long)b * P + p] = nt * d;
  }
  sa = wave_sum(sa);
  st = wave_sum(st);
  if ((tid & 63) == 0) { red[0][tid >> 6] = sa; red[1][tid >> 6] = st; }
  __syncthreads();
  if (tid == 0) {
    const float dna = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float dnt = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float iw = 1.f / ws, iw2 = iw * iw;
    const float dwa = dna * (iw - wa * iw2) + dnt * (-wt * iw2);
    const float dwt = dna * (-wa * iw2) + dnt * (iw - wt * iw2);
    dga[b] = dwa * wa * (1.f - wa);
    dgt[b] = dwt * wt * (1.f - wt);
  }
}

// ------------------------------------------------------------------------------------------
// fused training loss (ref src/train.py:154-168; models/losses.py:12-31,41-64; prototypes.py:13-53)
//   L = CE_ls + w_focal*focal + w_unc*mean(unc (x) correct) + w_proto*proto
// value + gradients in one single-workgroup launch (B <= 1024, C <= 16).
// ------------------------------------------------------------------------------------------
constexpr int LOSS_MAXB = 1024, LOSS_MAXC = 16;

__global__ __launch_bounds__(256) void train_loss_kernel(const float* __restrict__ logits, const float* __restrict__ unc,
                                                         const float* __restrict__ fused, const float* __restrict__ protos,
                                                         const int64_t* __restrict__ labels, int B, int C, int D,
                                                         float smoothing, float cb_beta, float gamma, float w_focal,
                                                         float w_unc, float w_proto, float margin, int use_proto,
                                                         const float* __restrict__ gscale, float* __restrict__ losses,
                                                         float* __restrict__ dlogits, float* __restrict__ dunc,
                                                         float* __restrict__ dfused, float* __restrict__ dprotos) {
  __shared__ float cw[LOSS_MAXC];
  __shared__ float red[5][4];
  __shared__ float s_ncorrect;
  __shared__ float pos_n[LOSS_MAXB];                 // ||e_b - P_y||
  __shared__ float dist[LOSS_MAXB][LOSS_MAXC];       // only filled when use_proto; B*C floats
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float g = gscale ? gscale[0] : 1.f;
  if (tid == 0) {
    float cnt[LOSS_MAXC];
    for (int c = 0; c < C; ++c) cnt[c] = 0.f;
    for (int b = 0; b < B; ++b) cnt[(int)labels[b]] += 1.f;
    float wsum = 0.f;
    for (int c = 0; c < C; ++c) {
      const float n = fmaxf(cnt[c], 1.f);
      const float eff = fmaxf(1.0f - powf(cb_beta, n), 1e-6f);
      cw[c] = (1.0f - cb_beta) / eff;
      wsum += cw[c];
    }
    for (int c = 0; c < C; ++c) cw[c] = cw[c] / (wsum + 1e-8f) * (float)C;
  }
  __syncthreads();
  float l_ce = 0.f, l_fo = 0.f, n_corr = 0.f, s_unc = 0.f;
  for (int b = tid; b < B; b += 256) {
    const int y = (int)labels[b];
    float z[LOSS_MAXC];
    float mx = -INFINITY;
    int am = 0;
    float amv = -INFINITY;
    for (int c = 0; c < C; ++c) {
      const float raw = logits[(long long)b * C + c];
      if (raw > amv) { amv = raw; am = c; }
      z[c] = fminf(fmaxf(raw, -10.f), 10.f);
      mx = fmaxf(mx, z[c]);
    }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
    const float lse = mx + logf(se);
    const float off = smoothing / (float)(C - 1);
    float ce = 0.f;
    for (int c = 0; c < C; ++c) ce -= (c == y ? 1.f - smoothing : off) * (z[c] - lse);
    l_ce += ce;
    const float lpt = z[y] - lse;
    const float pt_raw = expf(lpt);
    const float pt = fminf(fmaxf(pt_raw, 1e-6f), 1.0f);
    const float omp = 1.0f - pt;
    const float fw = powf(omp, gamma);
    const float cey = -cw[y] * lpt;
    l_fo += fw * cey;
    const float dfw_dpt = (pt_raw >= 1e-6f && pt_raw <= 1.0f) ? -gamma * powf(omp, gamma - 1.0f) : 0.f;
    n_corr += (am == y) ? 1.f : 0.f;
    s_unc += unc[b];
    for (int c = 0; c < C; ++c) {
      const float raw = logits[(long long)b * C + c];
      const float sm = expf(z[c] - lse);
      const float ind = c == y ? 1.f : 0.f;
      float dz = (sm - (c == y ? 1.f - smoothing : off)) / (float)B;                       // label-smoothed CE
      dz += w_focal / (float)B * (cey * dfw_dpt * pt_raw * (ind - sm) + fw * cw[y] * (sm - ind));
      dlogits[(long long)b * C + c] = (raw >= -10.f && raw <= 10.f) ? g * dz : 0.f;
    }
  }
  l_ce = wave_sum(l_ce); l_fo = wave_sum(l_fo); n_corr = wave_sum(n_corr); s_unc = wave_sum(s_unc);
  if (lane == 0) { red[0][w] = l_ce; red[1][w] = l_fo; red[2][w] = n_corr; red[3][w] = s_unc; }
  __syncthreads();
  if (tid == 0) s_ncorrect = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
  __syncthreads();
  const float ncorrect = s_ncorrect;
  for (int b = tid; b < B; b += 256) dunc[b] = g * w_unc * ncorrect / ((float)B * (float)B);

  // ---- prototype loss
  float l_pos = 0.f, l_neg = 0.f;
  if (use_proto) {
    // distances: one wave per sample, lanes over D
    for (int b = w; b < B; b += 4) {
      const int y = (int)labels[b];
      for (int c = 0; c < C; ++c) {
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) {
          const float e = fminf(fmaxf(fused[(long long)b * D + d], -10.f), 10.f);
          const float df = e - protos[(long long)c * D + d];
          acc = fmaf(df, df, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) {
          dist[b][c] = sqrtf(acc + 1e-6f);
          if (c == y) pos_n[b] = sqrtf(acc);
        }
      }
    }
    __syncthreads();
    // per-sample soft-min weights: s_bc = softmax_c(-nd_bc) over c != y ; stored back into dist as the
    // gradient coefficient  coef_bc = -(s_bc/B) * 1[d<=10] / d_bc   (own class: +1/(B*||.||))
    for (int b = tid; b < B; b += 256) {
      const int y = (int)labels[b];
      // the reference masks the own class with +inf and THEN clamps to max 10, so the own class
      // stays in the soft-min as the constant 10 (prototypes.py:44-48); it carries no gradient
      float mxn = -10.f;
      for (int c = 0; c < C; ++c)
        if (c != y) mxn = fmaxf(mxn, -fminf(dist[b][c], 10.f));
      float se = expf(-10.f - mxn);
      for (int c = 0; c < C; ++c)
        if (c != y) se += expf(-fminf(dist[b][c], 10.f) - mxn);
      const float lse = mxn + logf(se);
      l_pos += pos_n[b];
      l_neg += -lse;
      for (int c = 0; c < C; ++c) {
        if (c == y) {
          dist[b][c] = pos_n[b] > 0.f ? 1.0f / ((float)B * pos_n[b]) : 0.f;
        } else {
          const float dd = dist[b][c];
          const float s = expf(-fminf(dd, 10.f) - lse);
          dist[b][c] = dd <= 10.f ? -(s / (float)B) / dd : 0.f;
        }
      }
    }
    __syncthreads();
    // gradients: thread per feature column
    for (int d = tid; d < D; d += 256) {
      float dp[LOSS_MAXC];
      for (int c = 0; c < C; ++c) dp[c] = 0.f;
      for (int b = 0; b < B; ++b) {
        const float raw = fused[(long long)b * D + d];
        const float e = fminf(fmaxf(raw, -10.f), 10.f);
        float de = 0.f;
        for (int c = 0; c < C; ++c) {
          const float t = dist[b][c] * (e - protos[(long long)c * D + d]);
          de += t;
          dp[c] -= t;
        }
        dfused[(long long)b * D + d] = (raw >= -10.f && raw <= 10.f) ? g * w_proto * de : 0.f;
      }
      for (int c = 0; c < C; ++c) dprotos[(long long)c * D + d] = g * w_proto * dp[c];
    }
  } else {
    for (long long i = tid; i < (long long)B * D; i += 256) dfused[i] = 0.f;
    for (long long i = tid; i < (long long)C * D; i += 256) dprotos[i] = 0.f;
  }
  l_pos = wave_sum(l_pos); l_neg = wave_sum(l_neg);
  __syncthreads();
  if (lane == 0) { red[2][w] = l_pos; red[4][w] = l_neg; }
  __syncthreads();
  if (tid == 0) {
    const float ce = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)B;
    const float fo = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)B;
    const float su = (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]);
    const float ul = su * ncorrect / ((float)B * (float)B);
    float pl = 0.f;
    if (use_proto) {
      const float pos = ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) / (float)B;
      const float neg = ((red[4][0] + red[4][1]) + (red[4][2] + red[4][3])) / (float)B;
      pl = pos + margin - neg;
    }
    losses[0] = ce + w_focal * fo + w_unc * ul + (use_proto ? w_proto * pl : 0.f);
    losses[1] = ce; losses[2] = fo; losses[3] = ul; losses[4] = pl;
  }
}

// ------------------------------------------------------------------------------------------
// OpenMax rescale at inference (ref src/models/classifier.py:240-275): one wave per sample
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void openmax_kernel(const float* __restrict__ feats, const float* __restrict__ act_vec,
                                                      const float* __restrict__ walpha, const float* __restrict__ wbeta,
                                                      const float* __restrict__ wtau, int B, int C, int F, float thresh,
                                                      float reduce, float* __restrict__ logits) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float p = 0.f;
  for (int c = 0; c < C; ++c) {
    float acc = 0.f;
    for (int d = lane; d < F; d += 64) {
      const float df = feats[(long long)b * F + d] - act_vec[(long long)c * F + d];
      acc = fmaf(df, df, acc);
    }
    const float dist = sqrtf(wave_sum(acc));
    const float sb = fmaxf(wbeta[c], 1e-6f);
    const float sx = fmaxf(dist - wtau[c], 0.f);
    const float cdf = 1.0f - expf(-powf(sx / sb, walpha[c]));
    p = fmaxf(p, cdf);
  }
  if (p > thresh && lane < C) logits[(long long)b * C + lane] *= (1.0f - reduce * p);
}

// ------------------------------------------------------------------------------------------
// AdamW over a flat fp32 segment (torch.optim.AdamW single-tensor semantics).
// hyper = {lr, bias_correction1, sqrt(bias_correction2)} lives in device memory so a captured
// graph can be replayed with a new learning rate / step count.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    const float* __restrict__ hyper, float lr_mult, float wd, float b1,
                                                    float b2, float eps) {
  const float lr = hyper[0] * lr_mult, bc1 = hyper[1], bc2s = hyper[2];
  const float step = lr / bc1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
    const float den = sqrtf(vi) / bc2s + eps;
    pi -= step * (mi / den);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// y = x * mask / (1 - p) for one dropout site (in place when y == x); forward and backward apply the same multiplier
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, long long n, SerDropout d, float* __restrict__ y) {
  const unsigned long long st = *d.state;
  const unsigned thresh = ser_drop_thresh(d.p);
  const float scale = 1.0f / (1.0f - d.p);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = x[i] * ser_drop_mult(st, d.site, (unsigned)i, thresh, scale);
}

// the same update for up to 16 flat segments (one per optimizer group x bucket) in ONE launch: blockIdx.y = segment
constexpr int ADAMW_MAX_SEG = 16;
struct AdamwSeg {
  float* p; const float* g; float* m; float* v;
  long long n;
  float lr_mult, wd;
  const int* gate;   // optional device word: non-zero = this segment takes no update in this step (LayerDrop skipped its layer)
};
struct AdamwBatch {
  AdamwSeg s[ADAMW_MAX_SEG];
};
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamwBatch bt, const float* __restrict__ hyper, float b1,
                                                          float b2, float eps) {
  const AdamwSeg sg = bt.s[blockIdx.y];
  if (sg.gate != nullptr && *sg.gate != 0) return;   // torch.optim.AdamW leaves a parameter without a gradient untouched (moments too)
  const float lr = hyper[0] * sg.lr_mult, bc1 = hyper[1], bc2s = hyper[2];
  const float step = lr / bc1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < sg.n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = sg.g[i];
    float pi = sg.p[i] * (1.0f - lr * sg.wd);
    const float mi = sg.m[i] + (gi - sg.m[i]) * (1.0f - b1);
    const float vi = sg.v[i] * b2 + (1.0f - b2) * gi * gi;
    const float den = sqrtf(vi) / bc2s + eps;
    pi -= step * (mi / den);
    sg.p[i] = pi; sg.m[i] = mi; sg.v[i] = vi;
  }
}

static unsigned ew_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

#define LN_DISPATCH(KERNEL, nv, ...)                                                     \
  switch (nv) {                                                                          \
    case 1: hipLaunchKernelGGL(KERNEL<1>, grid, block, 0, st, __VA_ARGS__); break;       \
    case 2: hipLaunchKernelGGL(KERNEL<2>, grid, block, 0, st, __VA_ARGS__); break;       \
    case 3: hipLaunchKernelGGL(KERNEL<3>, grid, block, 0, st, __VA_ARGS__); break;       \
    default: hipLaunchKernelGGL(KERNEL<4>, grid, block, 0, st, __VA_ARGS__); break;      \
  }

extern "C" int ser_layernorm_fwd(const float* x, const float* x2, const float* gamma, const float* beta, float eps,
                                 int rows, int D, float* y, float* z, float* mean, float* rstd, void* stream) {
  SER_REQUIRE(D % 4 == 0 && D >= 4 && D <= 1024, "layernorm_fwd: D=%d unsupported", D);
  if (rows <= 0) return SER_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ceil_div(rows, 4)), block(256);
  LN_DISPATCH(ln_fwd_stats_kernel, ceil_div(D / 4, 64), x, x2, gamma, beta, eps, rows, D, y, z, mean, rstd, (SerDropout{nullptr, 0u, 0.f}));
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// y = LN(dropout(x) + x2): the hidden dropout, the residual add and the LayerNorm of a post-LN transformer block in one pass
extern "C" int ser_layernorm_drop_fwd(const float* x, const float* x2, const float* gamma, const float* beta, float eps, int rows, int D,
                                      float* y, float* z, float* mean, float* rstd, const void* drop_state, unsigned drop_site,
                                      float drop_p, void* stream) {
  SER_REQUIRE(D % 4 == 0 && D >= 4 && D <= 1024, "layernorm_drop_fwd: D=%d unsupported", D);
  SER_REQUIRE(drop_p >= 0.f && drop_p < 1.f && z, "layernorm_drop_fwd: p=%f out of range, or no z output", drop_p);
  if (rows <= 0) return SER_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ceil_div(rows, 4)), block(256);
  LN_DISPATCH(ln_fwd_stats_kernel, ceil_div(D / 4, 64), x, x2, gamma, beta, eps, rows, D, y, z, mean, rstd,
              (SerDropout{(const unsigned long long*)drop_state, drop_site, drop_p}));
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" size_t ser_layernorm_bwd_workspace_bytes(int rows, int D) {
  const int nsl = rows >= 512 ? 32 : 1;
  return nsl > 1 ? (size_t)nsl * 2 * D * sizeof(float) : 0;
}

static int layernorm_bwd_impl(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                              const float* dx_add, int rows, int D, float* dx, float* dx_drop, SerDropout drop, float* dgamma,
                              float* dbeta, int accumulate_params, void* workspace, void* stream) {
  SER_REQUIRE(D % 4 == 0 && D >= 4 && D <= 1024, "layernorm_bwd: D=%d unsupported", D);
  if (rows <= 0) return SER_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    dim3 grid(ceil_div(rows, 4)), block(256);
    LN_DISPATCH(ln_bwd_dx_kernel, ceil_div(D / 4, 64), dy, z, mean, rstd, gamma, dx_add, rows, D, dx, dx_drop, drop);
  }
  if (dgamma && dbeta) {
    const int nsl = (rows >= 512 && workspace) ? 32 : 1;
    float* part = nsl > 1 ? (float*)workspace : nullptr;
    hipLaunchKernelGGL(ln_bwd_param_kernel, dim3(ceil_div(D, 64), nsl), dim3(256), 0, st, dy, z, mean, rstd, rows, D, dgamma,
                       dbeta, accumulate_params, part);
    if (part)
      hipLaunchKernelGGL(ln_bwd_param_reduce_kernel, dim3(ceil_div(D, 256)), dim3(256), 0, st, part, nsl, D, dgamma, dbeta,
                         accumulate_params);
  }
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_layernorm_bwd(const float* dy, const float* z, const float* mean, const float* rstd,
                                 const float* gamma, const float* dx_add, int rows, int D, float* dx, float* dgamma,
                                 float* dbeta, int accumulate_params, void* workspace, void* stream) {
  return layernorm_bwd_impl(dy, z, mean, rstd, gamma, dx_add, rows, D, dx, nullptr, SerDropout{nullptr, 0u, 0.f}, dgamma, dbeta,
                            accumulate_params, workspace, stream);
}

// backward of y = LN(dropout(x) + x2): dx2 = the LayerNorm input gradient, dx = dx2 times the forward mask, in the same pass
extern "C" int ser_layernorm_drop_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                                      int rows, int D, float* dx, float* dx2, float* dgamma, float* dbeta, int accumulate_params,
                                      void* workspace, const void* drop_state, unsigned drop_site, float drop_p, void* stream) {
  SER_REQUIRE(dx && dx2 && drop_p >= 0.f && drop_p < 1.f, "layernorm_drop_bwd: bad argument");
  return layernorm_bwd_impl(dy, z, mean, rstd, gamma, nullptr, rows, D, dx2, dx, SerDropout{(const unsigned long long*)drop_state,
                            drop_site, drop_p}, dgamma, dbeta, accumulate_params, workspace, stream);
}

extern "C" int ser_layernorm2_fwd(const float* x, const float* g1, const float* b1, const float* g2, const float* b2,
                                  float eps, int rows, int D, float* y1, float* y2, float* stats, void* stream) {
  SER_REQUIRE(D % 4 == 0 && D >= 4 && D <= 1024, "layernorm2_fwd: D=%d unsupported", D);
  if (rows <= 0) return SER_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ceil_div(rows, 4)), block(256);
  LN_DISPATCH(ln2_fwd_kernel, ceil_div(D / 4, 64), x, g1, b1, g2, b2, eps, rows, D, y1, y2, stats);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_layernorm2_bwd(const float* du, const float* dres, const float* x, const float* y1, const float* stats,
                                  const float* g1, const float* g2, int rows, int D, float* dx, float* dg1, float* db1,
                                  float* dg2, float* db2, int accumulate, void* stream) {
  SER_REQUIRE(D % 4 == 0 && D >= 4 && D <= 512, "layernorm2_bwd: D=%d unsupported (<= 512)", D);
  if (rows <= 0) return SER_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(1), block(LN2_WAVES * 64);
  if (D <= 256)
    hipLaunchKernelGGL(ln2_bwd_kernel<1>, grid, block, 0, st, du, dres, x, y1, stats, g1, g2, rows, D, dx, dg1, db1, dg2, db2, accumulate);
  else
    hipLaunchKernelGGL(ln2_bwd_kernel<2>, grid, block, 0, st, du, dres, x, y1, stats, g1, g2, rows, D, dx, dg1, db1, dg2, db2, accumulate);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_colsum(const float* x, int M, int N, int ld, float* out, int accumulate, void* stream) {
  if (M <= 0 || N <= 0) return SER_OK;
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 64)), dim3(256), 0, (hipStream_t)stream, x, M, N, ld, out, accumulate);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// out[n] = sum_m x[m*ld + n] for tall x: SER_COLSUM_CHUNKS row chunks in parallel, then their sum; ws: SER_COLSUM_CHUNKS * N floats
#define SER_COLSUM_CHUNKS 32
extern "C" size_t ser_colsum_tall_workspace_bytes(int N) { return (size_t)SER_COLSUM_CHUNKS * (size_t)(N > 0 ? N : 0) * sizeof(float); }
extern "C" int ser_colsum_tall(const float* x, int M, int N, int ld, float* out, void* ws, void* stream) {
  if (M <= 0 || N <= 0) return SER_OK;
  SER_REQUIRE(ws, "colsum_tall: no workspace");
  hipLaunchKernelGGL(colsum_part_kernel, dim3(ceil_div(N, 64), SER_COLSUM_CHUNKS), dim3(256), 0, (hipStream_t)stream, x, M, N, ld, (float*)ws);
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 64)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, SER_COLSUM_CHUNKS, N, N, out, 0);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_act_drop_bwd(const float* dy, const float* y, int act, long long n, float* dx, const void* drop_state,
                                unsigned drop_site, float drop_p, void* stream) {
  if (n <= 0) return SER_OK;
  SER_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "act_drop_bwd: p=%f out of range", drop_p);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, act, n, dx,
                     SerDropout{(const unsigned long long*)drop_state, drop_site, drop_p});
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_act_drop_fwd(const float* x, int act, long long n, float* y, const void* drop_state, unsigned drop_site,
                                float drop_p, void* stream) {
  if (n <= 0) return SER_OK;
  SER_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "act_drop_fwd: p=%f out of range", drop_p);
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, act, n, y,
                     SerDropout{(const unsigned long long*)drop_state, drop_site, drop_p});
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_act_bwd(const float* dy, const float* y, int act, long long n, float* dx, void* stream) {
  if (n <= 0) return SER_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, act, n, dx, SerDropout{nullptr, 0u, 0.f});
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_act_fwd(const float* x, int act, long long n, float* y, void* stream) {
  if (n <= 0) return SER_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, act, n, y, SerDropout{nullptr, 0u, 0.f});
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_scale_dev(float* x, const float* s, long long n, void* stream) {
  if (n <= 0) return SER_OK;
  hipLaunchKernelGGL(scale_dev_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, s, n);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_axpby(const float* x, float a, float b, long long n, float* y, void* stream) {
  if (n <= 0) return SER_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, a, b, n, y);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// csrc/xattn_mfma.hip: the same operators on the fp32 matrix pipe (head_dim 32 / 64, <= 256 positions per side)
int ser_xattn_mfma_ok(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* c, int ldc, int Sq, int Sk,
                      int head_dim);
int ser_launch_xattn_fwd_mfma(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* key_mask, int B,
                              int Sq, int Sk, int heads, int head_dim, float* P, float* ctx, int ldc, SerDropout drop, float* P_dropped,
                              hipStream_t st);
int ser_launch_xattn_bwd_mfma(const float* dctx, int ldc, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                              const float* P, const float* Pv, int B, int Sq, int Sk, int heads, int head_dim, float* dS, float* dq,
                              int lddq, float* dk, int lddk, float* dv, int lddv, SerDropout drop, hipStream_t st);

extern "C" int ser_xattn_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                             const float* key_mask, int B, int Sq, int Sk, int heads, int head_dim, float* P, float* ctx,
                             int ldc, const void* drop_state, unsigned drop_site, float drop_p, float* P_dropped, void* stream) {
  SER_REQUIRE(Sk <= XA_MAXK && head_dim <= 64 && head_dim > 0, "xattn: Sk=%d (max %d) head_dim=%d (max 64)", Sk, XA_MAXK, head_dim);
  SER_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "xattn: dropout p=%f out of range", drop_p);
  SER_REQUIRE(!(drop_state && drop_p > 0.f) || P_dropped, "xattn: dropout needs the P_dropped output");
  const long long rows = (long long)B * heads * Sq;
  if (rows <= 0) return SER_OK;
  const SerDropout drop{(const unsigned long long*)drop_state, drop_site, drop_p};
  if (ser_xattn_mfma_ok(q, ldq, k, ldk, v, ldv, ctx, ldc, Sq, Sk, head_dim))
    return ser_launch_xattn_fwd_mfma(q, ldq, k, ldk, v, ldv, key_mask, B, Sq, Sk, heads, head_dim, P, ctx, ldc, drop, P_dropped,
                                     (hipStream_t)stream);
  if ((head_dim == 32 || head_dim == 64) && Sk <= XF_MAXS && Sq <= XF_MAXS) {
    if (head_dim == 32)
      hipLaunchKernelGGL(xattn_fwd_fast_kernel<32>, dim3(ceil_div(Sq, 16), heads, B), dim3(256), 0, (hipStream_t)stream, q, ldq, k, ldk,
                         v, ldv, key_mask, Sq, Sk, heads, P, ctx, ldc, drop, P_dropped);
    else
      hipLaunchKernelGGL(xattn_fwd_fast_kernel<64>, dim3(ceil_div(Sq, 16), heads, B), dim3(256), 0, (hipStream_t)stream, q, ldq, k, ldk,
                         v, ldv, key_mask, Sq, Sk, heads, P, ctx, ldc, drop, P_dropped);
    SER_LAUNCH_CHECK();
    return SER_OK;
  }
  hipLaunchKernelGGL(xattn_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, q, ldq, k, ldk, v,
                     ldv, key_mask, B, Sq, Sk, heads, head_dim, P, ctx, ldc, drop, P_dropped);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_xattn_bwd(const float* dctx, int ldc, const float* q, int ldq, const float* k, int ldk, const float* v,
                             int ldv, const float* P, int B, int Sq, int Sk, int heads, int head_dim, float* dS,
                             float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, const void* drop_state,
                             unsigned drop_site, float drop_p, const float* P_dropped, void* stream) {
  SER_REQUIRE(Sk <= XA_MAXK && head_dim <= 64 && head_dim > 0, "xattn_bwd: Sk=%d head_dim=%d unsupported", Sk, head_dim);
  hipStream_t st = (hipStream_t)stream;
  const long long rq = (long long)B * heads * Sq, rk = (long long)B * heads * Sk;
  if (rq <= 0 || rk <= 0) return SER_OK;
  const SerDropout drop{(const unsigned long long*)drop_state, drop_site, drop_p};
  const bool dropping = drop_state && drop_p > 0.f;
  SER_REQUIRE(!dropping || P_dropped, "xattn_bwd: dropout needs P_dropped from the forward call");
  const float* Pv = dropping ? P_dropped : P;       // weights of dv = Pv^T dctx
  if (ser_xattn_mfma_ok(q, ldq, k, ldk, v, ldv, dctx, ldc, Sq, Sk, head_dim))
    return ser_launch_xattn_bwd_mfma(dctx, ldc, q, ldq, k, ldk, v, ldv, P, Pv, B, Sq, Sk, heads, head_dim, dS, dq, lddq, dk, lddk, dv,
                                     lddv, drop, st);
  if ((head_dim == 32 || head_dim == 64) && Sk <= XF_MAXS && Sq <= XF_MAXS) {
    if (head_dim == 32) {
      hipLaunchKernelGGL(xattn_bwd_q_fast_kernel<32>, dim3(ceil_div(Sq, 16), heads, B), dim3(256), 0, st, dctx, ldc, k, ldk, v, ldv, P,
                         Sq, Sk, heads, dS, dq, lddq, drop);
      hipLaunchKernelGGL(xattn_bwd_kv_fast_kernel<32>, dim3(ceil_div(Sk, 64), heads, B), dim3(256), 0, st, dctx, ldc, q, ldq, Pv, dS, Sq,
                         Sk, heads, dk, lddk, dv, lddv);
    } else {
      hipLaunchKernelGGL(xattn_bwd_q_fast_kernel<64>, dim3(ceil_div(Sq, 16), heads, B), dim3(256), 0, st, dctx, ldc, k, ldk, v, ldv, P,
                         Sq, Sk, heads, dS, dq, lddq, drop);
      hipLaunchKernelGGL(xattn_bwd_kv_fast_kernel<64>, dim3(ceil_div(Sk, 64), heads, B), dim3(256), 0, st, dctx, ldc, q, ldq, Pv, dS, Sq,
                         Sk, heads, dk, lddk, dv, lddv);
    }
    SER_LAUNCH_CHECK();
    return SER_OK;
  }
  hipLaunchKernelGGL(xattn_bwd_q_kernel, dim3((unsigned)((rq + 3) / 4)), dim3(256), 0, st, dctx, ldc, k, ldk, v, ldv, P, B, Sq,
                     Sk, heads, head_dim, dS, dq, lddq, drop);
  hipLaunchKernelGGL(xattn_bwd_kv_kernel, dim3((unsigned)((rk + 3) / 4)), dim3(256), 0, st, dctx, ldc, q, ldq, Pv, dS, B, Sq, Sk,
                     heads, head_dim, dk, lddk, dv, lddv);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_pool_fwd(const float* x, const float* logits, const float* mask, int B, int S, int D, float* alpha,
                            float* out, void* stream) {
  SER_REQUIRE(S <= POOL_MAXS && S > 0, "pool_fwd: S=%d (max %d)", S, POOL_MAXS);
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(ceil_div(D, 256), B), dim3(256), 0, (hipStream_t)stream, x, logits, mask, S, D,
                     alpha, out);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_pool_bwd(const float* dout, const float* x, const float* alpha, const float* out, int B, int S, int D,
                            float* dx, float* dalpha_scratch, float* dlogits, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const long long rows = (long long)B * S;
  if (rows <= 0) return SER_OK;
  hipLaunchKernelGGL(pool_bwd_x_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, dout, x, alpha, out, B, S, D, dx,
                     dalpha_scratch);
  hipLaunchKernelGGL(pool_bwd_logit_kernel, dim3(B), dim3(256), 0, st, alpha, dalpha_scratch, S, dlogits);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_fusion_mix_fwd(const float* a, const float* t, const float* ga, const float* gt, int B, int P,
                                  float* out, void* stream) {
  if (B <= 0) return SER_OK;
  hipLaunchKernelGGL(fusion_mix_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a, t, ga, gt, P, out);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_fusion_mix_bwd(const float* dout, const float* a, const float* t, const float* ga, const float* gt,
                                  int B, int P, float* da, float* dt, float* dga, float* dgt, void* stream) {
  if (B <= 0) return SER_OK;
  hipLaunchKernelGGL(fusion_mix_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dout, a, t, ga, gt, P, da, dt, dga,
                     dgt);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_train_loss(const float* logits, const float* unc, const float* fused, const float* protos,
                              const int64_t* labels, int B, int C, int D, float smoothing, float cb_beta, float gamma,
                              float w_focal, float w_unc, float w_proto, float margin, int use_proto,
                              const float* grad_scale, float* losses, float* dlogits, float* dunc, float* dfused,
                              float* dprotos, void* stream) {
  SER_REQUIRE(B > 0 && B <= LOSS_MAXB && C >= 2 && C <= LOSS_MAXC, "train_loss: B=%d (max %d) C=%d (2..%d)", B, LOSS_MAXB, C, LOSS_MAXC);
  hipLaunchKernelGGL(train_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, unc, fused, protos, labels, B, C, D,
                     smoothing, cb_beta, gamma, w_focal, w_unc, w_proto, margin, use_proto, grad_scale, losses, dlogits,
                     dunc, dfused, dprotos);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_openmax(const float* feats, const float* act_vec, const float* walpha, const float* wbeta,
                           const float* wtau, int B, int C, int F, float thresh, float reduce, float* logits, void* stream) {
  SER_REQUIRE(C <= 64, "openmax: C=%d (max 64)", C);
  if (B <= 0) return SER_OK;
  hipLaunchKernelGGL(openmax_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, feats, act_vec, walpha, wbeta,
                     wtau, B, C, F, thresh, reduce, logits);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_dropout(const float* x, long long n, const void* state, unsigned site, float p, float* y, void* stream) {
  SER_REQUIRE(p >= 0.f && p < 1.f, "dropout: p=%f out of range", p);
  if (n <= 0) return SER_OK;
  if (!state || p == 0.f) {
    if (y != x) SER_CHECK_HIP(hipMemcpyAsync(y, x, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SER_OK;
  }
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, n,
                     SerDropout{(const unsigned long long*)state, site, p}, y);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// ptrs = host array {p, g, m, v} per segment, n = elements per segment, lr_mult / weight_decay per segment; gates = optional host
// array of device pointers (null entries allowed): a segment whose word is non-zero when the kernel runs is left untouched
extern "C" int ser_adamw_multi_gated(const void* const* ptrs, const long long* n, const float* lr_mult, const float* weight_decay,
                                     const void* const* gates, int nseg, const float* hyper, float beta1, float beta2, float eps,
                                     void* stream) {
  for (int s0 = 0; s0 < nseg; s0 += ADAMW_MAX_SEG) {
    const int cnt = nseg - s0 < ADAMW_MAX_SEG ? nseg - s0 : ADAMW_MAX_SEG;
    AdamwBatch bt;
    memset(&bt, 0, sizeof(bt));
    long long nmax = 0;
    for (int i = 0; i < cnt; ++i) {
      const int k = s0 + i;
      bt.s[i] = AdamwSeg{(float*)ptrs[4 * k], (const float*)ptrs[4 * k + 1], (float*)ptrs[4 * k + 2], (float*)ptrs[4 * k + 3],
                         n[k], lr_mult[k], weight_decay[k], gates ? (const int*)gates[k] : nullptr};
      if (n[k] > nmax) nmax = n[k];
    }
    if (nmax <= 0) continue;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(ew_blocks(nmax), cnt), dim3(256), 0, (hipStream_t)stream, bt, hyper, beta1,
                       beta2, eps);
    SER_LAUNCH_CHECK();
  }
  return SER_OK;
}

extern "C" int ser_adamw_multi(const void* const* ptrs, const long long* n, const float* lr_mult, const float* weight_decay,
                               int nseg, const float* hyper, float beta1, float beta2, float eps, void* stream) {
  return ser_adamw_multi_gated(ptrs, n, lr_mult, weight_decay, nullptr, nseg, hyper, beta1, beta2, eps, stream);
}

extern "C" int ser_adamw(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float lr_mult,
                         float weight_decay, float beta1, float beta2, float eps, void* stream) {
  if (n <= 0) return SER_OK;
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper, lr_mult,
                     weight_decay, beta1, beta2, eps);
  SER_LAUNCH_CHECK();
  return SER_OK;
}
