// Waveform augmentations of the data feed on the device (ref src/data/preprocess.py:50-73):
// band-limited resampling (torchaudio.functional.resample's windowed-sinc interpolation, evaluated tap by tap
// instead of through a [new, 2 width + orig] kernel table: with coprime rates such as 16000 -> 17123 that table has
// 17123 x 16012 entries of which ~13 per row are non-zero) and additive Gaussian noise at a target SNR.
#include "ser_common.h"

namespace {

// y[b][j], j < Lout:  sum_i x[b][(j / nw) * og + i] * k(phase = j % nw, i),   i in [-width, width + og)
//   k = sinc(pi t) * cos^2(pi t / (2 lpw)) * (base / og),  t = clamp((-phase / nw + i / og) * base, -lpw, lpw)
// (hf/torchaudio `_get_sinc_resample_kernel` + `_apply_sinc_resample_kernel`; og, nw = rates / gcd; zero padding
// outside the clip).  Only taps with |t| < lpw contribute: i in (c - lpw og / base, c + lpw og / base), c = phase og / nw.
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int T, int og, int nw, double base,
                                                       int lpw, int Lout, float* __restrict__ y) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (j >= Lout) return;
  const long long blk = j / nw;
  const int phase = (int)(j - blk * nw);
  const double c = (double)phase * og / nw;                       // centre of the kernel, in input samples
  const double half = (double)lpw * og / base;
  const long long i0 = (long long)floor(c - half), i1 = (long long)ceil(c + half);
  const float* xb = x + (long long)b * T;
  const double scale = base / og;
  double acc = 0.0;
  for (long long i = i0; i <= i1; ++i) {
    double t = (-(double)phase / nw + (double)i / og) * base;
    if (t <= -(double)lpw || t >= (double)lpw) continue;         // window is exactly zero there
    const long long src = blk * og + i;
    if (src < 0 || src >= T) continue;
    const double wnd = cos(t * M_PI / lpw / 2.0);
    const double a = t * M_PI;
    const double s = a == 0.0 ? 1.0 : sin(a) / a;
    // the reference builds the taps in float64 and rounds them to the waveform dtype before the convolution
    acc += (double)((float)(s * wnd * wnd * scale)) * (double)xb[src];
  }
  y[(long long)b * Lout + j] = (float)acc;
}

// mean power per clip -> noise standard deviation for the requested SNR (clamp(min=1e-12) as the reference)
__global__ __launch_bounds__(1024) void noise_sigma_kernel(const float* __restrict__ x, int T, const float* __restrict__ snr_db,
                                                           float* __restrict__ sigma) {
  __shared__ double sh[16];
  const float* xb = x + (long long)blockIdx.x * T;
  double s = 0.0;
  for (int i0 = threadIdx.x; i0 < T; i0 += 8 * 1024) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = xb[min(i0 + k * 1024, T - 1)];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * 1024 < T) s += (double)v[k] * (double)v[k];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < 16; ++i) tot += sh[i];
    double p = tot / T;
    if (p < 1e-12) p = 1e-12;
    sigma[blockIdx.x] = (float)sqrt(p / pow(10.0, (double)snr_db[blockIdx.x] / 10.0));
  }
}

// counter-based normal deviates: two 32-bit hashes of (seed, clip, sample) -> Box-Muller
SER_DEVFN unsigned mix32(unsigned h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x, int T, const float* __restrict__ sigma,
                                                        unsigned long long seed, float* __restrict__ y) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= T) return;
  const unsigned k0 = (unsigned)seed ^ (unsigned)(seed >> 32) * 0x9E3779B1u, ctr = (unsigned)i * 2u;
  const unsigned h1 = mix32(mix32(ctr ^ k0) + (unsigned)b * 0x632BE5ABu);
  const unsigned h2 = mix32(mix32((ctr + 1u) ^ k0) + (unsigned)b * 0x632BE5ABu + 0x7F4A7C15u);
  const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
  const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);               // [0, 1)
  const float n = sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
  const float v = x[(long long)b * T + i] + n * sigma[b];
  y[(long long)b * T + i] = fminf(fmaxf(v, -1.0f), 1.0f);
}

}  // namespace

static long long gcd_ll(long long a, long long b) { return b == 0 ? a : gcd_ll(b, a % b); }

extern "C" int ser_resample_out_len(int T, int orig_freq, int new_freq) {
  if (T <= 0 || orig_freq <= 0 || new_freq <= 0) return -1;
  const long long g = gcd_ll(orig_freq, new_freq);
  const long long og = orig_freq / g, nw = new_freq / g;
  return (int)((nw * (long long)T + og - 1) / og);                // ceil(new * T / orig)
}

// x [B, T] -> y [B, ser_resample_out_len(T, orig, new)]; lowpass_filter_width 6 and rolloff 0.99 are torchaudio's defaults
extern "C" int ser_resample(const float* x, int B, int T, int orig_freq, int new_freq, int lowpass_filter_width, float rolloff,
                            float* y, void* stream) {
  SER_REQUIRE(x && y && B > 0 && T > 0 && orig_freq > 0 && new_freq > 0 && lowpass_filter_width > 0, "resample: bad arguments");
  const long long g = gcd_ll(orig_freq, new_freq);
  const int og = (int)(orig_freq / g), nw = (int)(new_freq / g);
  const int Lout = ser_resample_out_len(T, orig_freq, new_freq);
  if (og == nw) {
    SER_CHECK_HIP(hipMemcpyAsync(y, x, (size_t)B * T * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SER_OK;
  }
  const double base = (double)(og < nw ? og : nw) * (double)rolloff;
  hipLaunchKernelGGL(resample_kernel, dim3((Lout + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, T, og, nw, base,
                     lowpass_filter_width, Lout, y);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

// y = clamp(x + N(0, sigma_b^2), -1, 1), sigma_b^2 = mean(x_b^2) / 10^(snr_db[b] / 10)   (preprocess.py:65-73)
// sigma: device scratch [B]; noise is a counter-based generator keyed by (seed, clip, sample).
extern "C" int ser_add_noise_snr(const float* x, int B, int T, const float* snr_db, unsigned long long seed, float* sigma,
                                 float* y, void* stream) {
  SER_REQUIRE(x && y && snr_db && sigma && B > 0 && T > 0, "add_noise_snr: bad arguments");
  hipLaunchKernelGGL(noise_sigma_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, T, snr_db, sigma);
  hipLaunchKernelGGL(add_noise_kernel, dim3((T + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, T, sigma, seed, y);
  SER_LAUNCH_CHECK();
  return SER_OK;
}
