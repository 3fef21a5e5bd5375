// Quality-gate and audio-conditioning front end of the reference's default AudioEncoder(), batched on the device.
//
//   reference (CPU, one clip at a time, numpy / librosa / scipy, a device -> host -> device round trip per clip):
//     src/models/quality_gates.py:497-560      VAD + STFT SNR + clipping + spectral descriptors -> 8 features + decision
//     src/models/audio_conditioning.py:503-584 hum notch -> high-pass -> Wiener -> (de-reverb) -> loudness -> 12 features
//     src/models/audio_encoder.py:65-87        clip := 0 unless 'accept'; conditioned clip goes to Wav2Vec2
//
// Everything a clip needs stays on the device: every data-dependent choice of the reference (is there hum, which
// high-pass cutoff, denoise or not, compress or not) is a per-clip flag in device memory that the next kernel reads, so a
// batch is a fixed sequence of launches with no host synchronisation.  Arithmetic is fp64 (the reference is float32 until
// the first scipy filter runs and float64 after; fp64 is within float32 rounding of either), transforms are a radix-2
// FFT in LDS, IIR filters run as chunked state-space recurrences (one chunk per thread, states chained through LDS),
// order statistics are a radix select on the IEEE bit patterns.  These kernels are HBM / latency bound and tiny next to
// the encoders (about 6 k FFTs of 2048 points per batch of 16 four-second clips); the point is that the default-flag
// encoder exists without librosa / scipy in the loop.
//
// Not built: webrtcvad (the caller gets the reference's own ValueError on the Python side), langdetect (the caller passes
// the language entropy / confidence pair, ref quality_gates.py:252-301), noisereduce (absent -> the reference's Wiener
// branch, which is what runs here), de-reverberation (unreachable: see fe_finish_kernel).
#include "ser_common.h"
#include <math.h>
#include <mutex>

namespace {

constexpr int NFFT = 2048;
constexpr int FE_T = 256;          // threads of the per-frame and per-clip kernels
constexpr int FE_IIR_T = 512;      // threads (= chunks) of the filter / scan kernels

struct FeTables {
  double2 tw[NFFT / 2];            // exp(-2 pi i k / 2048)
  double hann[NFFT];               // periodic Hann (scipy get_window('hann', N, fftbins=True)); hann_1024[n] = hann[2 n]
};

struct CondState {                 // per clip, device memory
  int hum50, hum60, hpf_on, denoise_on, noise_type, accept;
  double hpf_cutoff, snr_before, snr_after, gain_db, t60, lufs, adj, peak_db, ratio, e_mean;
};

FeTables* g_tables = nullptr;
std::mutex g_tables_mu;

int fe_tables(const FeTables** out) {
  std::lock_guard<std::mutex> lk(g_tables_mu);
  if (!g_tables) {
    FeTables* h = new FeTables;
    for (int k = 0; k < NFFT / 2; ++k) {
      h->tw[k].x = cos(-2.0 * M_PI * k / NFFT);
      h->tw[k].y = sin(-2.0 * M_PI * k / NFFT);
    }
    for (int n = 0; n < NFFT; ++n) h->hann[n] = 0.5 - 0.5 * cos(2.0 * M_PI * n / NFFT);
    FeTables* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, sizeof(FeTables));
    if (e == hipSuccess) e = hipMemcpy(d, h, sizeof(FeTables), hipMemcpyHostToDevice);
    delete h;
    if (e != hipSuccess) {
      ser_set_error("front end: table upload failed: %s", hipGetErrorString(e));
      return SER_E_HIP;
    }
    g_tables = d;
  }
  *out = g_tables;
  return SER_OK;
}

// ---- block-wide reductions (blockDim.x a power of two <= 512); result on every thread -----------------------------
SER_DEVFN double block_sum(double v, double* red) {
  const int t = threadIdx.x;
  __syncthreads();
  red[t] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  return red[0];
}
SER_DEVFN double block_max(double v, double* red) {
  const int t = threadIdx.x;
  __syncthreads();
  red[t] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (t < s) red[t] = fmax(red[t], red[t + s]);
    __syncthreads();
  }
  return red[0];
}
SER_DEVFN unsigned long long block_min_u64(unsigned long long v, double* red) {
  unsigned long long* r = (unsigned long long*)red;
  const int t = threadIdx.x;
  __syncthreads();
  r[t] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (t < s) r[t] = r[t] < r[t + s] ? r[t] : r[t + s];
    __syncthreads();
  }
  return r[0];
}

// k-th smallest (0-based) and its successor among n non-negative values v(i) (a radix select on the bit patterns, 8 bits
// per pass), then numpy's 'linear' percentile interpolation (numpy/lib/_function_base_impl.py _lerp).
template <typename F>
SER_DEVFN double block_percentile(F v, int n, double q, unsigned* hist /*[256]*/, double* red, unsigned long long* bc /*[2]*/) {
  const double vi = (double)(n - 1) * (q / 100.0);
  int k = (int)floor(vi);
  const double gamma = vi - (double)k;
  const int k0 = k;
  unsigned long long prefix = 0, mask = 0;
  unsigned eq = 0;
  for (int shift = 56; shift >= 0; shift -= 8) {
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(v(i));
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned cum = 0;
      int d = 0;
      for (; d < 255; ++d) {
        if (cum + hist[d] > (unsigned)k) break;
        cum += hist[d];
      }
      bc[0] = (unsigned long long)d;
      bc[1] = ((unsigned long long)cum << 32) | hist[d];
    }
    __syncthreads();
    const unsigned long long d = bc[0];
    k -= (int)(bc[1] >> 32);
    eq = (unsigned)(bc[1] & 0xffffffffu);
    prefix |= d << shift;
    mask |= 255ull << shift;
  }
  const double a = __longlong_as_double((long long)prefix);
  double b = a;
  if (k0 + 1 < n && (unsigned)(k + 1) >= eq) {      // the successor is the smallest value above a
    unsigned long long best = ~0ull;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(v(i));
      if (key > prefix && key < best) best = key;
    }
    b = __longlong_as_double((long long)block_min_u64(best, red));
  }
  const double d = b - a;
  return gamma >= 0.5 ? b - d * (1.0 - gamma) : a + d * gamma;
}

// ---- one frame: load (centre padding / detrend / window) -> 2048- or 1024-point FFT in LDS -> |X|^2 or |X| ----------
// grid (frames, clips).  desc (STFT only): per frame spectral centroid, bandwidth (p = 2) and 85 % roll-off in Hz
// (librosa.feature.spectral_*: magnitudes L1-normalised per frame, all-zero frames stay zero).
template <int N>
__global__ __launch_bounds__(FE_T) void fe_fft_kernel(const void* __restrict__ src, int src_f64, long long clip_stride, int T,
                                                       int hop, int pad, int reflect, int detrend, int nframes, double fs,
                                                       const FeTables* __restrict__ tab, double* __restrict__ power,
                                                       float* __restrict__ mag, double* __restrict__ desc) {
  constexpr int LOG = N == 2048 ? 11 : 10, TWS = NFFT / N, PER = N / FE_T, NB = N / 2 + 1;
  __shared__ double2 X[N];
  __shared__ double red[FE_T];
  const int f = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const long long base = (long long)f * hop - pad;
  double vals[PER], part = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    long long p = base + t + i * FE_T;
    if (reflect) {
      if (p < 0) p = -p;
      if (p >= T) p = 2ll * (T - 1) - p;
    }
    double v = 0;
    if (p >= 0 && p < T)
      v = src_f64 ? ((const double*)src)[b * clip_stride + p] : (double)((const float*)src)[b * clip_stride + p];
    vals[i] = v;
    part += v;
  }
  double mean = 0;
  if (detrend) mean = block_sum(part, red) / N;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int n = t + i * FE_T;
    X[__brev((unsigned)n) >> (32 - LOG)] = make_double2((vals[i] - mean) * tab->hann[n * TWS], 0.0);
  }
  __syncthreads();
  for (int s = 1; s <= LOG; ++s) {
    const int half = 1 << (s - 1);
#pragma unroll
    for (int i = 0; i < PER / 2; ++i) {
      const int j = t + i * FE_T;
      const int pos = j & (half - 1);
      const int i0 = ((j >> (s - 1)) << s) + pos, i1 = i0 + half;
      const double2 w = tab->tw[pos * (NFFT >> s)];
      const double2 a = X[i0], c = X[i1];
      const double tr = w.x * c.x - w.y * c.y, ti = w.x * c.y + w.y * c.x;
      X[i0] = make_double2(a.x + tr, a.y + ti);
      X[i1] = make_double2(a.x - tr, a.y - ti);
    }
    __syncthreads();
  }
  const long long row = ((long long)b * nframes + f) * NB;
  for (int k = t; k < NB; k += FE_T) {
    const double p2 = X[k].x * X[k].x + X[k].y * X[k].y;
    if (power) power[row + k] = p2;
    if (mag) mag[row + k] = (float)sqrt(p2);
  }
  if constexpr (N == 2048) {
    if (desc) {
      // thread t owns bins 4t..4t+3; bin 1024 (Nyquist) is thread 255's fifth
      const double df = fs / N;
      double s[5], tot = 0, fsum = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int k = 4 * t + i;
        const double2 xv = X[k < NB ? k : 0];
        s[i] = (i < 4 || t == FE_T - 1) ? (double)(float)sqrt(xv.x * xv.x + xv.y * xv.y) : 0.0;      // float32 magnitudes, as librosa holds them
        tot += s[i];
        fsum += s[i] * (k * df);
      }
      const double total = block_sum(tot, red);
      const double len = total < 1.1754943508222875e-38 ? 1.0 : total;
      const double cen = block_sum(fsum, red) / len;
      double bsum = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const double dv = (4 * t + i) * df - cen;
        bsum += s[i] / len * dv * dv;
      }
      const double bw = sqrt(block_sum(bsum, red));
      // roll-off: first bin whose running sum reaches 0.85 * total
      __syncthreads();
      red[t] = tot;
      __syncthreads();
      for (int o = 1; o < FE_T; o <<= 1) {        // inclusive scan of the per-thread sums
        const double add = t >= o ? red[t - o] : 0.0;
        __syncthreads();
        red[t] += add;
        __syncthreads();
      }
      double run = t > 0 ? red[t - 1] : 0.0;
      const double thr = 0.85 * total;
      unsigned long long first = ~0ull;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (i < 4 || t == FE_T - 1) {
          run += s[i];
          if (!(run < thr) && first == ~0ull) first = (unsigned long long)(4 * t + i);
        }
      }
      const unsigned long long kk = block_min_u64(first, red);
      if (t == 0) {
        double* o = desc + ((long long)b * nframes + f) * 3;
        o[0] = cen;
        o[1] = bw;
        o[2] = kk == ~0ull ? 0.0 : (double)kk * df;
      }
    }
  }
}

// ---- Welch PSD of one clip from its per-segment |X|^2 (scipy.signal.welch: Hann, 50 % overlap, constant detrend,
// density scaling, one-sided) and the decision the reference takes from it -----------------------------------------------
enum { WELCH_HUM = 0, WELCH_HPF = 1, WELCH_NOISE = 2 };

__global__ __launch_bounds__(FE_T) void fe_welch_kernel(const double* __restrict__ power, int nseg, int N, double fs,
                                                         const FeTables* __restrict__ tab, int mode, CondState* __restrict__ st) {
  __shared__ double psd[NFFT / 2 + 1];
  __shared__ double red[FE_T];
  const int b = blockIdx.x, t = threadIdx.x, NB = N / 2 + 1, tws = NFFT / N;
  double w2 = 0;
  for (int n = t; n < N; n += FE_T) w2 += tab->hann[n * tws] * tab->hann[n * tws];
  const double scale = 1.0 / (fs * block_sum(w2, red));
  const double* p = power + (long long)b * nseg * NB;
  double tot = 0;
  for (int k = t; k < NB; k += FE_T) {
    double a = 0;
    for (int s = 0; s < nseg; ++s) a += p[(long long)s * NB + k];
    a = a / nseg * scale * ((k == 0 || k == N / 2) ? 1.0 : 2.0);
    psd[k] = a;
    tot += a;
  }
  const double total = block_sum(tot, red);
  const double df = fs / N;
  CondState& c = st[b];
  if (mode == WELCH_HUM) {                       // ref audio_conditioning.py:66-82
    const double mean = total / NB;
    double dv = 0;
    for (int k = t; k < NB; k += FE_T) dv += (psd[k] - mean) * (psd[k] - mean);
    const double thr = mean + 2.0 * sqrt(block_sum(dv, red) / NB);
    if (t == 0) {
      for (int h = 0; h < 2; ++h) {
        const double hz = h == 0 ? 50.0 : 60.0;
        int k0 = (int)floor(hz / df);
        if (k0 + 1 < NB && fabs((k0 + 1) * df - hz) < fabs(k0 * df - hz)) ++k0;     // argmin |f - hz|, first on ties
        (h == 0 ? c.hum50 : c.hum60) = psd[k0] > thr;
      }
    }
  } else if (mode == WELCH_HPF) {                // ref :107-137
    double low = 0;
    for (int k = t; k < NB; k += FE_T)
      if (k * df < 200.0) low += psd[k];
    low = block_sum(low, red);
    if (t == 0) {
      const double ratio = total > 0 ? low / total : 0.0;
      c.hpf_on = ratio > 0.2;
      double cutoff = 80.0;
      if (c.hpf_on) {
        double cum = 0;
        for (int k = 0; k < NB; ++k) cum += psd[k];
        const double thr = 0.1 * cum;
        double run = 0;
        for (int k = 0; k < NB; ++k) {
          run += psd[k];
          if (run > thr) {
            cutoff = fmax(80.0, fmin(100.0, k * df));
            break;
          }
        }
      }
      c.hpf_cutoff = c.hpf_on ? cutoff : 0.0;
    }
  } else {                                       // ref :175-203
    double lo = 0, mid = 0, hi = 0;
    for (int k = t; k < NB; k += FE_T) {
      const double f = k * df;
      if (f < 500.0) lo += psd[k];
      else if (f < 2000.0) mid += psd[k];
      else hi += psd[k];
    }
    lo = block_sum(lo, red);
    mid = block_sum(mid, red);
    hi = block_sum(hi, red);
    if (t == 0) {
      const double s = lo + mid + hi;
      int ty = 0;                                // unknown
      if (s > 0) ty = lo / s > 0.5 ? 1 : (hi / s > 0.4 ? 2 : (mid / s > 0.6 ? 3 : 4));
      c.noise_type = ty;
    }
  }
}

// ---- zero-phase IIR filtering of a clip in place (scipy.signal.filtfilt: odd extension by 3 * taps samples, initial
// state = steady state of the first sample, forward then backward) -------------------------------------------------------
struct Iir {
  double b[5], a[5];
};
SER_DEVFN void iir_step(const Iir& f, double x, double* z, double& y) {       // direct form II transposed, as lfilter
  y = f.b[0] * x + z[0];
  z[0] = z[1] + f.b[1] * x - f.a[1] * y;
  z[1] = z[2] + f.b[2] * x - f.a[2] * y;
  z[2] = z[3] + f.b[3] * x - f.a[3] * y;
  z[3] = f.b[4] * x - f.a[4] * y;
}
// scipy.signal.iirnotch(w0, Q, fs)
SER_DEVFN void design_notch(double hz, double q, double fs, Iir& f) {
  const double w0 = hz / (fs / 2) * M_PI, bw = w0 / q;
  const double beta = tan(bw / 2.0);                       // sqrt(1 - gb^2) / gb = 1 for gb = 1/sqrt(2)
  const double gain = 1.0 / (1.0 + beta);
  f.b[0] = gain; f.b[1] = -2.0 * cos(w0) * gain; f.b[2] = gain; f.b[3] = f.b[4] = 0;
  f.a[0] = 1.0; f.a[1] = -2.0 * gain * cos(w0); f.a[2] = 2.0 * gain - 1.0; f.a[3] = f.a[4] = 0;
}
// scipy.signal.butter(4, wn, 'high'): analog prototype -> lp2hp -> bilinear (fs = 2) -> polynomial coefficients
SER_DEVFN void design_highpass4(double wn, Iir& f) {
  const double warped = 4.0 * tan(M_PI * wn / 2.0);
  double pr[4], pi[4];
  for (int m = 0; m < 4; ++m) {                            // buttap: p = -exp(i pi k / 8), k = -3, -1, 1, 3
    const double ang = M_PI * (2 * m - 3) / 8.0;
    const double ar = -cos(ang), ai = -sin(ang);
    const double den = ar * ar + ai * ai;                  // lp2hp: p -> warped / p
    const double hr = warped * ar / den, hi = -warped * ai / den;
    const double nr = 4.0 + hr, ni = hi, dr = 4.0 - hr, di = -hi, dd = dr * dr + di * di;   // bilinear: (4 + p) / (4 - p)
    pr[m] = (nr * dr + ni * di) / dd;
    pi[m] = (ni * dr - nr * di) / dd;
  }
  // gain: k = real(prod(4 - 0) / prod(4 - p_hp)) with four zeros at the origin -> 4^4 / prod(4 - p_hp)
  double gr = 1, gi = 0;
  for (int m = 0; m < 4; ++m) {
    const double ang = M_PI * (2 * m - 3) / 8.0;
    const double ar = -cos(ang), ai = -sin(ang), den = ar * ar + ai * ai;
    const double dr = 4.0 - warped * ar / den, di = warped * ai / den;
    const double tr = gr * dr - gi * di, ti = gr * di + gi * dr;
    gr = tr; gi = ti;
  }
  const double k = 256.0 * gr / (gr * gr + gi * gi);       // real part of 256 / (gr + i gi)
  double cr[5] = {1, 0, 0, 0, 0}, ci[5] = {0, 0, 0, 0, 0};    // poly(p): multiply out (x - p_m)
  for (int m = 0; m < 4; ++m) {
    for (int j = m + 1; j >= 1; --j) {
      const double tr = cr[j] - (pr[m] * cr[j - 1] - pi[m] * ci[j - 1]);
      const double ti = ci[j] - (pr[m] * ci[j - 1] + pi[m] * cr[j - 1]);
      cr[j] = tr; ci[j] = ti;
    }
  }
  const double bz[5] = {1, -4, 6, -4, 1};
  for (int j = 0; j < 5; ++j) { f.b[j] = k * bz[j]; f.a[j] = cr[j]; }
}

__global__ __launch_bounds__(FE_IIR_T) void fe_filtfilt_kernel(double* __restrict__ xs, int T, int which, double fs,
                                                                CondState* __restrict__ st, double* __restrict__ tmp, int tmp_stride) {
  __shared__ Iir F;
  __shared__ double zi[4], Ap[4][4];
  __shared__ double fin[FE_IIR_T][4], start[FE_IIR_T][4];
  const int b = blockIdx.x, t = threadIdx.x;
  const CondState& c = st[b];
  if (which == 0 ? !c.hum50 : (which == 1 ? !c.hum60 : !c.hpf_on)) return;
  double* x = xs + (long long)b * T;
  double* y1 = tmp + (long long)b * tmp_stride;
  const int padlen = which == 2 ? 15 : 9;                 // 3 * max(len(a), len(b))
  const int L = T + 2 * padlen, Lc = (L + FE_IIR_T - 1) / FE_IIR_T, nch = (L + Lc - 1) / Lc;
  if (t == 0) {
    if (which == 2) design_highpass4(c.hpf_cutoff / (fs / 2), F);
    else design_notch(which == 0 ? 50.0 : 60.0, 30.0, fs, F);
    // lfilter_zi: (I - A) zi = B, A = companion(a)^T, B = b[1:] - a[1:] b[0]
    double M[4][5];
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 4; ++j) M[i][j] = (i == j ? 1.0 : 0.0) - ((j == 0 ? -F.a[i + 1] : 0.0) + (j == i + 1 ? 1.0 : 0.0));
      M[i][4] = F.b[i + 1] - F.a[i + 1] * F.b[0];
    }
    for (int col = 0; col < 4; ++col) {
      int piv = col;
      for (int r = col + 1; r < 4; ++r)
        if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
      for (int j = 0; j < 5; ++j) { const double tt = M[col][j]; M[col][j] = M[piv][j]; M[piv][j] = tt; }
      for (int r = 0; r < 4; ++r) {
        if (r == col) continue;
        const double m = M[r][col] / M[col][col];
        for (int j = col; j < 5; ++j) M[r][j] -= m * M[col][j];
      }
    }
    for (int i = 0; i < 4; ++i) zi[i] = M[i][4] / M[i][i];
  }
  __syncthreads();
  const Iir f = F;
  if (t < 4) {            // column t of A^Lc: Lc zero-input steps from the unit state e_t
    double z[4] = {0, 0, 0, 0}, y;
    z[t] = 1.0;
    for (int i = 0; i < Lc; ++i) iir_step(f, 0.0, z, y);
    for (int i = 0; i < 4; ++i) Ap[i][t] = z[i];
  }
  const double x0 = x[0], xl = x[T - 1];
  auto ext = [&](int i) -> double {                        // scipy odd_ext
    if (i < padlen) return 2.0 * x0 - x[padlen - i];
    if (i < padlen + T) return x[i - padlen];
    return 2.0 * xl - x[T - 2 - (i - padlen - T)];
  };
  for (int pass = 0; pass < 2; ++pass) {
    auto in = [&](int i) -> double { return pass == 0 ? ext(i) : y1[L - 1 - i]; };
    const int i0 = t * Lc, i1 = min(L, i0 + Lc);
    {                      // chunk response from the zero state
      double z[4] = {0, 0, 0, 0}, y;
      for (int i = i0; i < i1; ++i) iir_step(f, in(i), z, y);
      for (int j = 0; j < 4; ++j) fin[t][j] = z[j];
    }
    __syncthreads();
    if (t == 0) {          // chain the chunk start states: s_{c+1} = A^Lc s_c + fin_c
      const double first = in(0);
      double s[4];
      for (int j = 0; j < 4; ++j) s[j] = zi[j] * first;
      for (int cidx = 0; cidx < nch; ++cidx) {
        for (int j = 0; j < 4; ++j) start[cidx][j] = s[j];
        double n[4];
        for (int i = 0; i < 4; ++i) n[i] = Ap[i][0] * s[0] + Ap[i][1] * s[1] + Ap[i][2] * s[2] + Ap[i][3] * s[3] + fin[cidx][i];
        for (int j = 0; j < 4; ++j) s[j] = n[j];
      }
    }
    __syncthreads();
    if (i0 < L) {
      double z[4] = {start[t][0], start[t][1], start[t][2], start[t][3]}, y;
      if (pass == 0) {
        // the forward output may not overwrite x yet (ext() of other chunks still reads it): it goes to y1
        for (int i = i0; i < i1; ++i) { iir_step(f, in(i), z, y); y1[i] = y; }
      } else {
        for (int i = i0; i < i1; ++i) {
          iir_step(f, in(i), z, y);
          const int tt = L - 1 - i - padlen;              // result = reversed backward output without the padding
          if (tt >= 0 && tt < T) x[tt] = y;
        }
      }
    }
    __syncthreads();
  }
}

// ---- clip := wave (or zeros when the quality gates did not say 'accept'), widened to fp64 -----------------------------
__global__ void fe_load_kernel(const float* __restrict__ wave, const int* __restrict__ decision, int B, int T,
                               double* __restrict__ x, CondState* __restrict__ st) {
  const long long n = (long long)B * T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / T);
    const bool acc = !decision || decision[b] == 2;
    x[i] = acc ? (double)wave[i] : 0.0;
  }
  for (int b = threadIdx.x; blockIdx.x == 0 && b < B; b += blockDim.x) {
    CondState& c = st[b];
    c.hum50 = c.hum60 = c.hpf_on = c.denoise_on = c.noise_type = 0;
    c.accept = !decision || decision[b] == 2;
    c.hpf_cutoff = c.gain_db = 0;
  }
}

// ---- energy-ratio SNR (ref audio_conditioning.py:161-173): 10 log10(mean x^2 / 10th percentile of x^2), in [0, 50] ------
__global__ __launch_bounds__(FE_T) void fe_snr_kernel(const double* __restrict__ xs, int T, int after, CondState* __restrict__ st) {
  __shared__ double red[FE_T];
  __shared__ unsigned hist[256];
  __shared__ unsigned long long bc[2];
  const int b = blockIdx.x;
  const double* x = xs + (long long)b * T;
  double s = 0;
  for (int i = threadIdx.x; i < T; i += FE_T) s += x[i] * x[i];
  const double e = block_sum(s, red) / T;
  const double floor_ = block_percentile([&](int i) { return x[i] * x[i]; }, T, 10.0, hist, red, bc);
  if (threadIdx.x == 0) {
    double snr = floor_ > 0 ? 10.0 * log10(e / floor_) : 50.0;
    snr = fmax(0.0, fmin(50.0, snr));
    CondState& c = st[b];
    if (after) c.snr_after = snr;
    else {
      c.snr_before = snr;
      c.e_mean = e;
      c.denoise_on = snr < 15.0;          // ref :244-256 (the flag the features report is "gain != 0", set by the Wiener kernel)
    }
  }
}

// ---- scipy.signal.wiener with a window of M = 2 * int(0.1 T) samples, in place (ref :197-215) --------------------------
// local mean / variance are box sums over [i - M/2, i + M/2 - 1] (correlate(..., 'same') with an even kernel), zero outside
// the clip, divided by M; noise power = mean local variance.
__global__ __launch_bounds__(FE_IIR_T) void fe_wiener_kernel(double* __restrict__ xs, int T, int M, CondState* __restrict__ st,
                                                              double* __restrict__ P1s, double* __restrict__ P2s) {
  __shared__ double red[FE_IIR_T];
  __shared__ double c1[FE_IIR_T], c2[FE_IIR_T];
  const int b = blockIdx.x, t = threadIdx.x;
  CondState& c = st[b];
  if (!c.denoise_on) return;
  double* x = xs + (long long)b * T;
  double* P1 = P1s + (long long)b * T;
  double* P2 = P2s + (long long)b * T;
  const int Lc = (T + FE_IIR_T - 1) / FE_IIR_T, i0 = min(T, t * Lc), i1 = min(T, i0 + Lc);
  double a1 = 0, a2 = 0;
  for (int i = i0; i < i1; ++i) { a1 += x[i]; a2 += x[i] * x[i]; }
  c1[t] = a1; c2[t] = a2;
  __syncthreads();
  if (t == 0) {
    double r1 = 0, r2 = 0;
    for (int j = 0; j < FE_IIR_T; ++j) {
      const double v1 = c1[j], v2 = c2[j];
      c1[j] = r1; c2[j] = r2;
      r1 += v1; r2 += v2;
    }
  }
  __syncthreads();
  a1 = c1[t]; a2 = c2[t];
  for (int i = i0; i < i1; ++i) { a1 += x[i]; a2 += x[i] * x[i]; P1[i] = a1; P2[i] = a2; }
  __syncthreads();
  const int h = M / 2;
  auto stats = [&](int i, double& mean, double& var) {
    const int lo = max(0, i - h), hi = min(T - 1, i + h - 1);
    const double s1 = P1[hi] - (lo > 0 ? P1[lo - 1] : 0.0), s2 = P2[hi] - (lo > 0 ? P2[lo - 1] : 0.0);
    mean = s1 / M;
    var = s2 / M - mean * mean;
  };
  double sv = 0;
  for (int i = t; i < T; i += FE_IIR_T) { double m, v; stats(i, m, v); sv += v; }
  const double noise = block_sum(sv, red) / T;
  double e1 = 0;
  for (int i = t; i < T; i += FE_IIR_T) {
    double m, v;
    stats(i, m, v);
    double r = (x[i] - m) * (1.0 - noise / v) + m;
    if (v < noise) r = m;
    x[i] = r;
    e1 += r * r;
  }
  e1 = block_sum(e1, red) / T;
  if (t == 0) c.gain_db = e1 > 0 ? 10.0 * log10(e1 / c.e_mean) : 0.0;
}

// ---- reverberation estimate, loudness normalisation, the 12 features (ref :274-301, :364-440, :556-577) ----------------
// The reference's T60 is where(cumsum(decay^2) < 0.001 total)[0][0] / sr: the running sum never decreases, so that index is
// 0 (when the peak sample alone is below -30 dB of the tail energy) or missing (-> 0.1 s).  Both are below the 0.5 s
// threshold, so its `simple_dereverb` never runs and feature 3 is always 0.
__global__ __launch_bounds__(FE_T) void fe_finish_kernel(const double* __restrict__ xs, int T, double fs, CondState* __restrict__ st,
                                                          float* __restrict__ out, float* __restrict__ raw, float* __restrict__ meta) {
  __shared__ double red[FE_T];
  const int b = blockIdx.x, t = threadIdx.x;
  const double* x = xs + (long long)b * T;
  CondState& c = st[b];
  double mx = 0, s2 = 0;
  for (int i = t; i < T; i += FE_T) { mx = fmax(mx, fabs(x[i])); s2 += x[i] * x[i]; }
  const double peak = block_max(mx, red);
  const double ms = block_sum(s2, red) / T;
  unsigned long long first = ~0ull;                         // np.argmax: first index of the maximum
  for (int i = t; i < T; i += FE_T)
    if (fabs(x[i]) == peak) { first = (unsigned long long)i; break; }
  const int p = (int)block_min_u64(first, red);
  double tail = 0;
  for (int i = p + t; i < T; i += FE_T) tail += x[i] * x[i];
  tail = block_sum(tail, red);
  double t60 = 0.1;
  if ((double)(T - p) >= fs && tail != 0.0 && x[p] * x[p] < tail * 0.001) t60 = 0.0;
  const double r = sqrt(ms);
  const double lufs = r > 0 ? 20.0 * log10(r) - 70.0 : -60.0;
  const double dr = r > 0 ? 20.0 * log10(peak / r) : 0.0;
  const bool comp = dr > 40.0;
  const double thr = 2.0 * r, ratio = comp ? fmin(4.0, dr / 40.0) : 1.0;
  const double adj = fmax(-6.0, fmin(6.0, -23.0 - lufs));
  const double g = pow(10.0, adj / 20.0);
  double mo = 0;
  for (int i = t; i < T; i += FE_T) {
    double v = x[i];
    if (comp && fabs(v) > thr) v = copysign(thr + (fabs(v) - thr) / ratio, v);
    v *= g;
    mo = fmax(mo, fabs(v));
    out[(long long)b * T + i] = (float)v;
  }
  const double p1 = block_max(mo, red);
  if (t == 0) {
    const double peak_db = peak > 0 ? 20.0 * log10(p1 / peak) : 0.0;
    const int den = c.gain_db != 0.0;
    c.t60 = t60; c.lufs = lufs; c.adj = adj; c.peak_db = peak_db; c.ratio = ratio;
    float* f = raw + b * 12;
    f[0] = (float)(c.hum50 || c.hum60); f[1] = (float)c.hpf_on; f[2] = (float)den; f[3] = 0.f;
    f[4] = (float)(c.snr_before / 50.0); f[5] = (float)(c.snr_after / 50.0); f[6] = (float)(c.gain_db / 20.0);
    f[7] = (float)(t60 / 2.0); f[8] = (float)((lufs + 60.0) / 60.0); f[9] = (float)(adj / 20.0);
    f[10] = (float)(peak_db / 20.0); f[11] = (float)(ratio / 4.0);
    if (meta) {
      float* m = meta + b * 12;
      m[0] = (float)c.hpf_cutoff; m[1] = (float)c.hum50; m[2] = (float)c.hum60; m[3] = (float)c.snr_before;
      m[4] = (float)c.snr_after; m[5] = (float)c.gain_db; m[6] = (float)t60; m[7] = (float)lufs; m[8] = (float)adj;
      m[9] = (float)peak_db; m[10] = (float)ratio; m[11] = (float)c.noise_type;
    }
  }
}

// =====================================================================================================================
// quality gates
// =====================================================================================================================

// librosa.feature.rms: centre-padded frames, sqrt(mean x^2); one wave per frame
__global__ __launch_bounds__(FE_T) void fe_rms_kernel(const float* __restrict__ wave, int T, int flen, int hop, int reflect,
                                                       int nframes, float* __restrict__ out) {
  const int b = blockIdx.y, f = blockIdx.x * (FE_T / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
  if (f >= nframes) return;
  const float* x = wave + (long long)b * T;
  const long long base = (long long)f * hop - flen / 2;
  double s = 0;
  for (int n = lane; n < flen; n += 64) {
    long long p = base + n;
    if (reflect) {
      if (p < 0) p = -p;
      if (p >= T) p = 2ll * (T - 1) - p;
    }
    if (p >= 0 && p < T) s += (double)x[p] * (double)x[p];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[(long long)b * nframes + f] = (float)sqrt(s / flen);
}

// energy VAD (ref quality_gates.py:111-137): speech = rms > 30th percentile + 0.1 std, 5-tap median (scipy.ndimage
// 'reflect' boundary), speech_prob = mean
constexpr int FE_MAX_VAD_FRAMES = 16384;
__global__ __launch_bounds__(FE_T) void fe_vad_kernel(const float* __restrict__ energy, int nfr, double* __restrict__ speech_prob) {
  __shared__ double red[FE_T];
  __shared__ unsigned hist[256];
  __shared__ unsigned long long bc[2];
  __shared__ unsigned char sp[FE_MAX_VAD_FRAMES];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* e = energy + (long long)b * nfr;
  double s = 0;
  for (int i = t; i < nfr; i += FE_T) s += e[i];
  const double mean = block_sum(s, red) / nfr;
  double dv = 0;
  for (int i = t; i < nfr; i += FE_T) dv += (e[i] - mean) * (e[i] - mean);
  const double sd = sqrt(block_sum(dv, red) / nfr);
  const double p30 = block_percentile([&](int i) { return (double)e[i]; }, nfr, 30.0, hist, red, bc);
  const double thr = p30 + 0.1 * sd;
  for (int i = t; i < nfr; i += FE_T) sp[i] = (double)e[i] > thr;
  __syncthreads();
  double cnt = 0;
  for (int i = t; i < nfr; i += FE_T) {
    int on = 0;
    for (int d = -2; d <= 2; ++d) {
      int j = i + d;
      if (j < 0) j = -j - 1;
      if (j >= nfr) j = 2 * nfr - 1 - j;
      j = max(0, min(nfr - 1, j));
      on += sp[j];
    }
    cnt += on >= 3;
  }
  cnt = block_sum(cnt, red);
  if (t == 0) speech_prob[b] = cnt / nfr;
}

// SNR from the STFT magnitudes, clipping, spectral naturalness, music / laughter scores, the abstain policy, the quality
// score and the 8 features (ref :189-247, :320-403, :497-560)
__global__ __launch_bounds__(FE_T) void fe_quality_kernel(const float* __restrict__ wave, int T, const float* __restrict__ mags,
                                                           const double* __restrict__ descs, int F, int nf,
                                                           const float* __restrict__ rms_long, int nfr_long,
                                                           const double* __restrict__ speech_prob, const float* __restrict__ lid,
                                                           float* __restrict__ raw, float* __restrict__ metrics, int* __restrict__ decision) {
  __shared__ double red[FE_T];
  const int b = blockIdx.x, t = threadIdx.x, NB = NFFT / 2 + 1;
  const float* x = wave + (long long)b * T;
  float mx = 0.f;
  for (int i = t; i < T; i += FE_T) mx = fmaxf(mx, fabsf(x[i]));
  mx = (float)block_max((double)mx, red);
  double cnt = 0;
  for (int i = t; i < T; i += FE_T) {
    const float v = mx > 0.f ? __fdiv_rn(x[i], mx) : x[i];        // float32 division, as numpy does on the float32 clip
    cnt += fabsf(v) > 0.95f;
  }
  const double clip = block_sum(cnt, red) / T * 100.0;
  double snr = 50.0;
  if (nf > 0) {
    const float* m = mags + (long long)b * F * NB;
    double sp = 0, np_ = 0;
    for (int k = t; k < NB; k += FE_T) {
      double sg = 0, no = 0;
      for (int f = nf; f < F - nf; ++f) sg += m[(long long)f * NB + k];
      for (int f = F - nf; f < F; ++f) no += m[(long long)f * NB + k];
      sg /= (F - 2 * nf);
      no /= nf;
      sp += sg * sg;
      np_ += no * no;
    }
    sp = block_sum(sp, red) / NB;
    np_ = block_sum(np_, red) / NB;
    snr = np_ > 0 ? 10.0 * log10(sp / np_) : 50.0;
    snr = fmax(0.0, fmin(50.0, snr));
  }
  const double* d = descs + (long long)b * F * 3;
  double c0 = 0, c1 = 0, c2 = 0;
  for (int f = t; f < F; f += FE_T) { c0 += d[f * 3]; c1 += d[f * 3 + 1]; c2 += d[f * 3 + 2]; }
  const double cen = block_sum(c0, red) / F, bw = block_sum(c1, red) / F, roll = block_sum(c2, red) / F;
  const float* rl = rms_long + (long long)b * nfr_long;
  double s = 0;
  for (int i = t; i < nfr_long; i += FE_T) s += rl[i];
  const double rmean = block_sum(s, red) / nfr_long;
  double dv = 0;
  for (int i = t; i < nfr_long; i += FE_T) dv += (rl[i] - rmean) * (rl[i] - rmean);
  const double rvar = block_sum(dv, red) / nfr_long;
  if (t == 0) {
    auto clip01 = [](double v) { return fmax(0.0, fmin(1.0, v)); };
    const double nat = ((1.0 - clip01(fabs(cen - 2000.0) / 2000.0)) + (1.0 - clip01(fabs(roll - 0.85) / 0.15)) +
                        (1.0 - clip01(fabs(bw - 1000.0) / 1000.0))) / 3.0;
    const double music = clip01(cen / 4000.0), laugh = clip01(rvar / 0.1);
    const double spp = speech_prob[b], ent = lid[b * 2], conf = lid[b * 2 + 1];
    int dec = 1;                                     // 0 reject, 1 uncertain, 2 accept (ref :347-380)
    if (snr < 5.0 || clip > 30.0 || spp < 0.4) dec = 0;
    else if ((snr >= 5.0 && snr < 10.0) || ent > 1.5 || music > 0.2) dec = 1;
    else if (snr >= 10.0 && spp >= 0.8 && ent < 1.5) dec = 2;
    const double score = 0.25 * clip01(snr / 20.0) + 0.25 * spp + 0.15 * (1.0 - clip01(clip / 100.0)) + 0.15 * nat +
                         0.10 * (1.0 - clip01(ent / 2.0)) + 0.10 * (1.0 - music);
    float* f = raw + b * 8;
    f[0] = (float)spp; f[1] = (float)(snr / 50.0); f[2] = (float)(clip / 100.0); f[3] = (float)nat;
    f[4] = (float)(ent / 2.0); f[5] = (float)conf; f[6] = (float)music; f[7] = (float)laugh;
    if (metrics) {
      float* q = metrics + b * 8;
      q[0] = (float)spp; q[1] = (float)snr; q[2] = (float)clip; q[3] = (float)nat; q[4] = (float)music; q[5] = (float)laugh;
      q[6] = (float)score; q[7] = (float)dec;
    }
    decision[b] = dec;
  }
}

struct FeLayout {
  double *x, *tmp, *P1, *P2, *power, *desc, *speech;
  float *mag, *rms_vad, *rms_long;
  CondState* st;
  int tmp_stride, nseg2048, nseg1024, F, nfr_vad, nfr_long;
};

int fe_layout(SerArena& ar, int B, int T, FeLayout& l) {
  l.tmp_stride = T + 32;
  l.nseg2048 = T >= 2048 ? (T - 1024) / 1024 : 0;           // scipy: (T - noverlap) // (nperseg - noverlap)
  l.nseg1024 = T >= 1024 ? (T - 512) / 512 : 0;
  l.F = 1 + T / 512;
  l.nfr_vad = 1 + T / 160;
  l.nfr_long = l.F;
  const size_t pw = (size_t)max(l.nseg2048 * 1025, l.nseg1024 * 513);
  l.x = ar.get<double>((size_t)B * T);
  l.tmp = ar.get<double>((size_t)B * l.tmp_stride);
  l.P1 = ar.get<double>((size_t)B * T);
  l.P2 = ar.get<double>((size_t)B * T);
  l.power = ar.get<double>((size_t)B * pw);
  l.desc = ar.get<double>((size_t)B * l.F * 3);
  l.speech = ar.get<double>((size_t)B);
  l.mag = ar.get<float>((size_t)B * l.F * 1025);
  l.rms_vad = ar.get<float>((size_t)B * l.nfr_vad);
  l.rms_long = ar.get<float>((size_t)B * l.nfr_long);
  l.st = ar.get<CondState>((size_t)B);
  return SER_OK;
}

int fe_check(const char* who, int B, int T, int sample_rate) {
  SER_REQUIRE(B >= 1 && B <= 65535, "%s: batch %d out of range", who, B);
  SER_REQUIRE(sample_rate == 16000, "%s: the reference builds its front end for 16 kHz (audio_encoder.py:26,36); got %d", who, sample_rate);
  SER_REQUIRE(T >= 2048, "%s: clips must hold at least 2048 samples (one analysis window); got %d", who, T);
  SER_REQUIRE(1 + T / 160 <= FE_MAX_VAD_FRAMES, "%s: clip of %d samples is longer than the %d VAD frames the kernel holds", who, T, FE_MAX_VAD_FRAMES);
  return SER_OK;
}

}  // namespace

extern "C" size_t ser_frontend_workspace_bytes(int B, int T) {
  if (B < 1 || T < 1) return 0;
  SerArena ar(nullptr, 0);
  FeLayout l;
  fe_layout(ar, B, T, l);
  return ar.off + 256;
}

extern "C" int ser_frontend_init(void) {
  const FeTables* tab;
  return fe_tables(&tab);
}

extern "C" int ser_quality_gates(const float* wave, int B, int T, int sample_rate, const float* lid, int pad_reflect,
                                 float* q_raw, float* q_metrics, int* decision, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  SER_TRY(fe_check("ser_quality_gates", B, T, sample_rate));
  SER_REQUIRE(wave && lid && q_raw && decision && workspace, "ser_quality_gates: null argument");
  SER_REQUIRE(workspace_bytes >= ser_frontend_workspace_bytes(B, T), "ser_quality_gates: workspace too small");
  const FeTables* tab;
  SER_TRY(fe_tables(&tab));
  hipStream_t st = (hipStream_t)stream;
  SerArena ar(workspace, workspace_bytes);
  FeLayout l;
  fe_layout(ar, B, T, l);
  const double fs = sample_rate;
  hipLaunchKernelGGL(fe_rms_kernel, dim3(ceil_div(l.nfr_vad, FE_T / 64), B), dim3(FE_T), 0, st, wave, T, (int)(fs * 0.025),
                     (int)(fs * 0.010), pad_reflect, l.nfr_vad, l.rms_vad);
  hipLaunchKernelGGL(fe_rms_kernel, dim3(ceil_div(l.nfr_long, FE_T / 64), B), dim3(FE_T), 0, st, wave, T, 2048, 512, pad_reflect,
                     l.nfr_long, l.rms_long);
  hipLaunchKernelGGL(fe_vad_kernel, dim3(B), dim3(FE_T), 0, st, l.rms_vad, l.nfr_vad, l.speech);
  hipLaunchKernelGGL(fe_fft_kernel<2048>, dim3(l.F, B), dim3(FE_T), 0, st, (const void*)wave, 0, (long long)T, T, 512, 1024,
                     pad_reflect, 0, l.F, fs, tab, (double*)nullptr, l.mag, l.desc);
  const int nf = (int)(0.1 * (double)l.F);
  hipLaunchKernelGGL(fe_quality_kernel, dim3(B), dim3(FE_T), 0, st, wave, T, l.mag, l.desc, l.F, nf, l.rms_long, l.nfr_long,
                     l.speech, lid, q_raw, q_metrics, decision);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_audio_conditioning(const float* wave, const int* decision, int B, int T, int sample_rate, float* out,
                                      float* c_raw, float* c_meta, void* workspace, size_t workspace_bytes, void* stream) {
  SER_TRY(fe_check("ser_audio_conditioning", B, T, sample_rate));
  SER_REQUIRE(wave && out && c_raw && workspace, "ser_audio_conditioning: null argument");
  SER_REQUIRE(workspace_bytes >= ser_frontend_workspace_bytes(B, T), "ser_audio_conditioning: workspace too small");
  const FeTables* tab;
  SER_TRY(fe_tables(&tab));
  hipStream_t st = (hipStream_t)stream;
  SerArena ar(workspace, workspace_bytes);
  FeLayout l;
  fe_layout(ar, B, T, l);
  const double fs = sample_rate;
  long long blocks = ((long long)B * T + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fe_load_kernel, dim3((unsigned)blocks), dim3(FE_T), 0, st, wave, decision, B, T, l.x, l.st);
  auto welch = [&](int N, int nseg, int mode) {
    if (N == 2048)
      hipLaunchKernelGGL(fe_fft_kernel<2048>, dim3(nseg, B), dim3(FE_T), 0, st, (const void*)l.x, 1, (long long)T, T, 1024, 0, 0, 1,
                         nseg, fs, tab, l.power, (float*)nullptr, (double*)nullptr);
    else
      hipLaunchKernelGGL(fe_fft_kernel<1024>, dim3(nseg, B), dim3(FE_T), 0, st, (const void*)l.x, 1, (long long)T, T, 512, 0, 0, 1,
                         nseg, fs, tab, l.power, (float*)nullptr, (double*)nullptr);
    hipLaunchKernelGGL(fe_welch_kernel, dim3(B), dim3(FE_T), 0, st, l.power, nseg, N, fs, tab, mode, l.st);
  };
  welch(2048, l.nseg2048, WELCH_HUM);                                   // 1. hum notch
  for (int which = 0; which < 2; ++which)
    hipLaunchKernelGGL(fe_filtfilt_kernel, dim3(B), dim3(FE_IIR_T), 0, st, l.x, T, which, fs, l.st, l.tmp, l.tmp_stride);
  welch(2048, l.nseg2048, WELCH_HPF);                                   // 2. high-pass
  hipLaunchKernelGGL(fe_filtfilt_kernel, dim3(B), dim3(FE_IIR_T), 0, st, l.x, T, 2, fs, l.st, l.tmp, l.tmp_stride);
  hipLaunchKernelGGL(fe_snr_kernel, dim3(B), dim3(FE_T), 0, st, l.x, T, 0, l.st);   // 3. adaptive denoise
  welch(1024, l.nseg1024, WELCH_NOISE);
  const int M = 2 * (int)(0.1 * (double)T);
  hipLaunchKernelGGL(fe_wiener_kernel, dim3(B), dim3(FE_IIR_T), 0, st, l.x, T, M, l.st, l.P1, l.P2);
  hipLaunchKernelGGL(fe_snr_kernel, dim3(B), dim3(FE_T), 0, st, l.x, T, 1, l.st);
  hipLaunchKernelGGL(fe_finish_kernel, dim3(B), dim3(FE_T), 0, st, l.x, T, fs, l.st, out, c_raw, c_meta);   // 4./5. + features
  SER_LAUNCH_CHECK();
  return SER_OK;
}
