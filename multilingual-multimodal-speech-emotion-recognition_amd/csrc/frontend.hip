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
constexpr int FE_T = 256;          // threads of the per-frame (FFT, rms) kernels
constexpr int FE_C = 1024;         // threads of the one-workgroup-per-clip kernels (statistics, decisions)
constexpr int FE_IIR_T = 512;      // threads (= chunks) of the filter / scan kernels
constexpr int FE_CAND = 1024;      // radix select: candidates that are finished by direct ranking in LDS

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

// k-th smallest (0-based) and its successor among n non-negative values v(i), then numpy's 'linear' percentile
// interpolation (numpy/lib/_function_base_impl.py _lerp).  A radix select on the IEEE bit patterns, 8 bits per pass; squared
// samples share their leading bits, so the histogram increments are combined per wave (one atomic per distinct digit in
// the wave) and, as soon as at most FE_CAND elements share the prefix, those are gathered into LDS and ranked directly.
struct SelectLds {
  unsigned hist[256];
  unsigned long long bc[2];
  unsigned long long cand[FE_CAND];
  unsigned ncand;
};
SER_DEVFN void wave_hist_add(unsigned* hist, bool active, unsigned digit) {
  // one combined increment for the digit of the first active lane (squared samples: usually the whole wave), plain LDS
  // atomics for the lanes that differ (few, or spread over many counters)
  const unsigned long long todo = __ballot(active);
  if (!todo) return;
  const int leader = __ffsll((long long)todo) - 1;
  const unsigned d = (unsigned)__shfl((int)digit, leader, 64);
  const unsigned long long same = __ballot(active && digit == d);
  if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[d], (unsigned)__popcll(same));
  else if (active && digit != d) atomicAdd(&hist[digit], 1u);
}
template <typename F>
SER_DEVFN double block_percentile(F v, int n, double q, SelectLds& S, double* red) {
  const double vi = (double)(n - 1) * (q / 100.0);
  int k = (int)floor(vi);
  const double gamma = vi - (double)k;
  const int k0 = k, nt = blockDim.x, t = threadIdx.x;
  unsigned long long prefix = 0, mask = 0;
  unsigned group = (unsigned)n;                   // elements whose key matches the prefix
  int shift = 56;
  for (; shift >= 0 && group > FE_CAND; shift -= 8) {
    __syncthreads();
    for (int i = t; i < 256; i += nt) S.hist[i] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += nt) {
      const int i = base + t;
      unsigned long long key = 0;
      if (i < n) key = (unsigned long long)__double_as_longlong(v(i));
      wave_hist_add(S.hist, i < n && (key & mask) == prefix, (unsigned)(key >> shift) & 255u);
    }
    __syncthreads();
    if (t == 0) {
      unsigned cum = 0;
      int d = 0;
      for (; d < 255; ++d) {
        if (cum + S.hist[d] > (unsigned)k) break;
        cum += S.hist[d];
      }
      S.bc[0] = (unsigned long long)d;
      S.bc[1] = ((unsigned long long)cum << 32) | S.hist[d];
    }
    __syncthreads();
    k -= (int)(S.bc[1] >> 32);
    group = (unsigned)(S.bc[1] & 0xffffffffu);
    prefix |= S.bc[0] << shift;
    mask |= 255ull << shift;
  }
  unsigned long long ka = prefix, kb = prefix;
  bool have_next = (unsigned)(k + 1) < group;     // all 64 bits fixed: the group is `group` copies of one value
  if (shift >= 0) {                               // rank the remaining candidates in LDS
    __syncthreads();
    if (t == 0) S.ncand = 0;
    __syncthreads();
    for (int i = t; i < n; i += nt) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(v(i));
      if ((key & mask) == prefix) S.cand[atomicAdd(&S.ncand, 1u)] = key;
    }
    __syncthreads();
    const int m = (int)S.ncand;
    for (int c = t; c < m; c += nt) {
      const unsigned long long key = S.cand[c];
      int r = 0;
      for (int j = 0; j < m; ++j) r += (S.cand[j] < key) || (S.cand[j] == key && j < c);
      if (r == k) S.bc[0] = key;
      if (r == k + 1) S.bc[1] = key;
    }
    __syncthreads();
    ka = S.bc[0];
    have_next = k + 1 < m;
    kb = have_next ? S.bc[1] : ka;
  }
  if (!have_next && k0 + 1 < n) {                 // the successor lies outside the group: smallest key above ka
    unsigned long long best = ~0ull;
    for (int i = t; i < n; i += nt) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(v(i));
      if (key > ka && key < best) best = key;
    }
    kb = block_min_u64(best, red);
  }
  const double a = __longlong_as_double((long long)ka), b = __longlong_as_double((long long)kb);
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

__global__ __launch_bounds__(FE_C) void fe_welch_kernel(const double* __restrict__ power, int nseg, int N, double fs,
                                                         const FeTables* __restrict__ tab, int mode, CondState* __restrict__ st) {
  __shared__ double psd[NFFT / 2 + 1];
  __shared__ double red[FE_C];
  const int b = blockIdx.x, t = threadIdx.x, NB = N / 2 + 1, tws = NFFT / N;
  double w2 = 0;
  for (int n = t; n < N; n += FE_C) w2 += tab->hann[n * tws] * tab->hann[n * tws];
  const double scale = 1.0 / (fs * block_sum(w2, red));
  const double* p = power + (long long)b * nseg * NB;
  double tot = 0;
  for (int k = t; k < NB; k += FE_C) {
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
    for (int k = t; k < NB; k += FE_C) dv += (psd[k] - mean) * (psd[k] - mean);
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
    for (int k = t; k < NB; k += FE_C)
      if (k * df < 200.0) low += psd[k];
    low = block_sum(low, red);
    // first bin whose running sum exceeds 10 % of the total (np.cumsum + np.where, ref :125-133): each thread owns a run
    // of consecutive bins, the run totals are scanned, the crossing is the smallest index any run reports
    const int per = (NB + FE_C - 1) / FE_C, k0 = t * per, k1 = min(NB, k0 + per);
    double mine = 0;
    for (int k = k0; k < k1; ++k) mine += psd[k];
    __syncthreads();
    red[t] = mine;
    __syncthreads();
    for (int o = 1; o < FE_C; o <<= 1) {
      const double add = t >= o ? red[t - o] : 0.0;
      __syncthreads();
      red[t] += add;
      __syncthreads();
    }
    const double thr = 0.1 * red[FE_C - 1];
    double run = t > 0 ? red[t - 1] : 0.0;
    unsigned long long first = ~0ull;
    for (int k = k0; k < k1; ++k) {
      run += psd[k];
      if (run > thr && first == ~0ull) first = (unsigned long long)k;
    }
    const unsigned long long kc = block_min_u64(first, red);
    if (t == 0) {
      const double ratio = total > 0 ? low / total : 0.0;
      c.hpf_on = ratio > 0.2;
      double cutoff = 80.0;
      if (c.hpf_on && kc != ~0ull) cutoff = fmax(80.0, fmin(100.0, (double)kc * df));
      c.hpf_cutoff = c.hpf_on ? cutoff : 0.0;
    }
  } else {                                       // ref :175-203
    double lo = 0, mid = 0, hi = 0;
    for (int k = t; k < NB; k += FE_C) {
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
// The transfer function is run as a cascade of second-order sections in direct form II transposed, not as scipy's
// single fourth-order recurrence: the same linear operator, but a parallel evaluation has to carry filter states across
// chunk boundaries, and the fourth-order companion states of a Butterworth high-pass at 80-100 Hz / 16 kHz (four poles
// within 0.04 of z = 1) amplify a rounding error in a carried state by ~1e5 (its powers reach 1e3 per 16 samples),
// which put a two-level state chain 4e-5 away from scipy.  Biquad states amplify by < 1e2; this form agrees with
// scipy.signal.filtfilt to 7e-11 on the fixture clips (numpy emulation of this schedule), i.e. at scipy's own rounding level.
//
// Schedule: the extended clip is processed in tiles of FE_IIR_T x FE_SUB samples held in LDS (coalesced global reads and
// writes, the next tile's reads in flight while this one is filtered); inside a tile every thread owns FE_SUB consecutive
// samples.  Per section: (1) every thread runs its samples from the zero state -> fin_t; (2) the start states
// s_{t+1} = P s_t + fin_t (P = A^FE_SUB) come from a scan with the uniform matrix P: inside a wave a Hillis-Steele scan
// e_t += P^d e_{t-d} (d = 1..32, shuffles), across the waves W_{w+1} = P^64 W_w + E_w from the state carried out of the
// previous tile, and s_t = P^lane W_w + e_{t-1}; (3) every thread re-runs its samples from s_t, in place.
constexpr int FE_SUB = 16;                              // samples per thread and tile
constexpr int FE_TILE = FE_IIR_T * FE_SUB;              // 8192
constexpr int FE_LDS_STRIDE = FE_SUB + 1;               // odd stride in doubles: the lanes of a wave hit different banks

struct Biquad {
  double b0, b1, b2, a1, a2;
};
struct Cascade {
  int n, padlen;
  Biquad s[2];
};
SER_DEVFN void bq_step(const Biquad& f, double x, double& z0, double& z1, double& y) {
  y = f.b0 * x + z0;
  z0 = z1 + f.b1 * x - f.a1 * y;
  z1 = f.b2 * x - f.a2 * y;
}
// scipy.signal.iirnotch(w0, Q, fs): one section
SER_DEVFN void design_notch(double hz, double q, double fs, Cascade& c) {
  const double w0 = hz / (fs / 2) * M_PI, bw = w0 / q;
  const double beta = tan(bw / 2.0);                       // sqrt(1 - gb^2) / gb = 1 for gb = 1/sqrt(2)
  const double gain = 1.0 / (1.0 + beta);
  c.n = 1;
  c.padlen = 9;                                            // 3 * max(len(a), len(b))
  c.s[0] = Biquad{gain, -2.0 * cos(w0) * gain, gain, -2.0 * gain * cos(w0), 2.0 * gain - 1.0};
}
// scipy.signal.butter(4, wn, 'high') (analog prototype -> lp2hp -> bilinear with fs = 2) with its four poles paired into
// two sections, numerator (1 - z^-1)^2 each, the gain in the first
SER_DEVFN void design_highpass4(double wn, Cascade& c) {
  const double warped = 4.0 * tan(M_PI * wn / 2.0);
  double gr = 1, gi = 0;                                   // prod(4 - p_hp) over all four poles
  for (int m = 0; m < 4; ++m) {
    const double ang = M_PI * (2 * m - 3) / 8.0;           // buttap: p = -exp(i pi k / 8), k = -3, -1, 1, 3
    const double ar = -cos(ang), ai = -sin(ang), den = ar * ar + ai * ai;
    const double hr = warped * ar / den, hi = -warped * ai / den;        // lp2hp: p -> warped / p
    const double dr = 4.0 - hr, di = -hi;
    const double tr = gr * dr - gi * di, ti = gr * di + gi * dr;
    gr = tr; gi = ti;
    if (m < 2) {                                           // k = -3 and k = -1; their conjugates are k = 3 and k = 1
      const double nr = 4.0 + hr, ni = hi, dd = dr * dr + di * di;       // bilinear: (4 + p) / (4 - p)
      const double pr = (nr * dr + ni * di) / dd, pi = (ni * dr - nr * di) / dd;
      c.s[m] = Biquad{1.0, -2.0, 1.0, -2.0 * pr, pr * pr + pi * pi};
    }
  }
  const double k = 256.0 * gr / (gr * gr + gi * gi);       // real(4^4 / prod(4 - p_hp)): four zeros at the origin
  c.s[0].b0 *= k; c.s[0].b1 *= k; c.s[0].b2 *= k;
  c.n = 2;
  c.padlen = 15;
}

SER_DEVFN void m2_apply(const double* M /*[4] row-major*/, double& v0, double& v1) {
  const double n0 = M[0] * v0 + M[1] * v1, n1 = M[2] * v0 + M[3] * v1;
  v0 = n0;
  v1 = n1;
}

__global__ __launch_bounds__(FE_IIR_T) void fe_filtfilt_kernel(double* __restrict__ xs, int T, int which, double fs,
                                                                CondState* __restrict__ st, double* __restrict__ tmp, int tmp_stride) {
  constexpr int NW = FE_IIR_T / 64;
  __shared__ Cascade F;
  __shared__ double zi[2][2], dc[2];
  __shared__ double Pw[2][7][4];                  // per section: P^(2^j), j = 0..6, P = A^FE_SUB (row-major 2 x 2)
  __shared__ double carry[2][2][2];               // [tile parity][section][state]
  __shared__ double Ew[2][NW][2];                 // per section: zero-state response across each wave
  __shared__ double buf[FE_IIR_T * FE_LDS_STRIDE];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const CondState& c = st[b];
  if (which == 0 ? !c.hum50 : (which == 1 ? !c.hum60 : !c.hpf_on)) return;
  double* x = xs + (long long)b * T;
  double* y1 = tmp + (long long)b * tmp_stride;
  if (t == 0) {
    if (which == 2) design_highpass4(c.hpf_cutoff / (fs / 2), F);
    else design_notch(which == 0 ? 50.0 : 60.0, 30.0, fs, F);
    for (int s = 0; s < F.n; ++s) {
      // steady state for a unit step (scipy lfilter_zi / sosfilt_zi): (I - A) zi = B, A = [[-a1, 1], [-a2, 0]],
      // B = [b1 - a1 b0, b2 - a2 b0]; the next section sees the step scaled by this one's DC gain
      const Biquad& q = F.s[s];
      const double B0 = q.b1 - q.a1 * q.b0, B1 = q.b2 - q.a2 * q.b0;
      const double det = (1.0 + q.a1) + q.a2;              // det [[1 + a1, -1], [a2, 1]]
      zi[s][0] = (B0 + B1) / det;
      zi[s][1] = ((1.0 + q.a1) * B1 - q.a2 * B0) / det;
      dc[s] = (q.b0 + q.b1 + q.b2) / det;
    }
  }
  __syncthreads();
  const Cascade f = F;
  const int padlen = f.padlen, L = T + 2 * padlen;
  if (t < 2 * f.n) {       // column (t & 1) of P = A^FE_SUB of section t >> 1: FE_SUB zero-input steps from a unit state
    const int s = t >> 1, col = t & 1;
    double z0 = col == 0 ? 1.0 : 0.0, z1 = col == 1 ? 1.0 : 0.0, y;
    for (int i = 0; i < FE_SUB; ++i) bq_step(f.s[s], 0.0, z0, z1, y);
    Pw[s][0][col] = z0;
    Pw[s][0][2 + col] = z1;
  }
  __syncthreads();
  if (t < f.n) {           // P^2, P^4, ..., P^64 by squaring
    for (int j = 1; j < 7; ++j) {
      const double* m = Pw[t][j - 1];
      Pw[t][j][0] = m[0] * m[0] + m[1] * m[2];
      Pw[t][j][1] = m[0] * m[1] + m[1] * m[3];
      Pw[t][j][2] = m[2] * m[0] + m[3] * m[2];
      Pw[t][j][3] = m[2] * m[1] + m[3] * m[3];
    }
  }
  const double x0 = x[0], xl = x[T - 1];
  auto ext = [&](int i) -> double {                        // scipy odd_ext
    if (i < padlen) return 2.0 * x0 - x[padlen - i];
    if (i < padlen + T) return x[i - padlen];
    return 2.0 * xl - x[T - 2 - (i - padlen - T)];
  };
  double* mine = buf + t * FE_LDS_STRIDE;
  for (int pass = 0; pass < 2; ++pass) {
    auto in = [&](int i) -> double { return i < L ? (pass == 0 ? ext(i) : y1[L - 1 - i]) : 0.0; };
    __syncthreads();
    if (t == 0) {
      double first = in(0);
      for (int s = 0; s < f.n; ++s) {
        carry[0][s][0] = zi[s][0] * first;
        carry[0][s][1] = zi[s][1] * first;
        first *= dc[s];
      }
    }
    double nx[FE_SUB];
#pragma unroll
    for (int k = 0; k < FE_SUB; ++k) nx[k] = in(t + k * FE_IIR_T);
    int par = 0;
    for (int t0 = 0; t0 < L; t0 += FE_TILE, par ^= 1) {
      const int n = min(FE_TILE, L - t0);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < FE_SUB; ++k) {
        const int i = t + k * FE_IIR_T;
        buf[(i / FE_SUB) * FE_LDS_STRIDE + (i % FE_SUB)] = nx[k];
      }
      __syncthreads();
      if (t0 + FE_TILE < L) {
#pragma unroll
        for (int k = 0; k < FE_SUB; ++k) nx[k] = in(t0 + FE_TILE + t + k * FE_IIR_T);
      }
      const int cnt = max(0, min(FE_SUB, n - t * FE_SUB));           // my samples in this tile
      const int last = (n - 1) / FE_SUB;                             // the thread that holds the tile's last sample
      double xr[FE_SUB];
#pragma unroll
      for (int i = 0; i < FE_SUB; ++i) xr[i] = mine[i];
      for (int s = 0; s < f.n; ++s) {
        const Biquad q = f.s[s];
        double e0 = 0, e1 = 0, y;
#pragma unroll
        for (int i = 0; i < FE_SUB; ++i)
          if (i < cnt) bq_step(q, xr[i], e0, e1, y);
#pragma unroll
        for (int j = 0; j < 6; ++j) {                                // e_t = sum_{i <= t} P^(t - i) fin_i inside the wave
          double u0 = __shfl_up(e0, 1u << j, 64), u1 = __shfl_up(e1, 1u << j, 64);
          m2_apply(Pw[s][j], u0, u1);
          if (lane >= (1 << j)) { e0 += u0; e1 += u1; }
        }
        if (lane == 63) { Ew[s][wave][0] = e0; Ew[s][wave][1] = e1; }
        double p0 = __shfl_up(e0, 1, 64), p1 = __shfl_up(e1, 1, 64);
        if (lane == 0) p0 = p1 = 0.0;
        __syncthreads();
        double w0 = carry[par][s][0], w1 = carry[par][s][1];         // state at the start of my wave
        for (int w = 0; w < wave; ++w) {
          m2_apply(Pw[s][6], w0, w1);
          w0 += Ew[s][w][0];
          w1 += Ew[s][w][1];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
          if ((lane >> j) & 1) m2_apply(Pw[s][j], w0, w1);           // P^lane
        double z0 = w0 + p0, z1 = w1 + p1;
#pragma unroll
        for (int i = 0; i < FE_SUB; ++i)
          if (i < cnt) { bq_step(q, xr[i], z0, z1, y); xr[i] = y; }
        if (t == last) {                                              // the true state after the tile's last sample
          carry[par ^ 1][s][0] = z0;
          carry[par ^ 1][s][1] = z1;
        }
      }
#pragma unroll
      for (int i = 0; i < FE_SUB; ++i) mine[i] = xr[i];
      __syncthreads();
      for (int i = t; i < n; i += FE_IIR_T) {
        const double y = buf[(i / FE_SUB) * FE_LDS_STRIDE + (i % FE_SUB)];
        const int gi = t0 + i;
        if (pass == 0) y1[gi] = y;                         // x is still read by ext() of later tiles: forward output -> y1
        else {
          const int tt = L - 1 - gi - padlen;              // result = reversed backward output without the padding
          if (tt >= 0 && tt < T) x[tt] = y;
        }
      }
    }
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
__global__ __launch_bounds__(FE_C) void fe_snr_kernel(const double* __restrict__ xs, int T, int after, CondState* __restrict__ st) {
  __shared__ double red[FE_C];
  __shared__ SelectLds sel;
  const int b = blockIdx.x;
  const double* x = xs + (long long)b * T;
  double s = 0;
  for (int i = threadIdx.x; i < T; i += FE_C) s += x[i] * x[i];
  const double e = block_sum(s, red) / T;
  const double floor_ = block_percentile([&](int i) { return x[i] * x[i]; }, T, 10.0, sel, red);
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
__global__ __launch_bounds__(FE_C) void fe_finish_kernel(const double* __restrict__ xs, int T, double fs, CondState* __restrict__ st,
                                                          float* __restrict__ out, float* __restrict__ raw, float* __restrict__ meta) {
  __shared__ double red[FE_C];
  const int b = blockIdx.x, t = threadIdx.x;
  const double* x = xs + (long long)b * T;
  CondState& c = st[b];
  double mx = 0, s2 = 0;
  for (int i = t; i < T; i += FE_C) { mx = fmax(mx, fabs(x[i])); s2 += x[i] * x[i]; }
  const double peak = block_max(mx, red);
  const double ms = block_sum(s2, red) / T;
  unsigned long long first = ~0ull;                         // np.argmax: first index of the maximum
  for (int i = t; i < T; i += FE_C)
    if (fabs(x[i]) == peak) { first = (unsigned long long)i; break; }
  const int p = (int)block_min_u64(first, red);
  double tail = 0;
  for (int i = p + t; i < T; i += FE_C) tail += x[i] * x[i];
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
  for (int i = t; i < T; i += FE_C) {
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
__global__ __launch_bounds__(FE_C) void fe_vad_kernel(const float* __restrict__ energy, int nfr, double* __restrict__ speech_prob) {
  __shared__ double red[FE_C];
  __shared__ SelectLds sel;
  __shared__ unsigned char sp[FE_MAX_VAD_FRAMES];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* e = energy + (long long)b * nfr;
  double s = 0;
  for (int i = t; i < nfr; i += FE_C) s += e[i];
  const double mean = block_sum(s, red) / nfr;
  double dv = 0;
  for (int i = t; i < nfr; i += FE_C) dv += (e[i] - mean) * (e[i] - mean);
  const double sd = sqrt(block_sum(dv, red) / nfr);
  const double p30 = block_percentile([&](int i) { return (double)e[i]; }, nfr, 30.0, sel, red);
  const double thr = p30 + 0.1 * sd;
  for (int i = t; i < nfr; i += FE_C) sp[i] = (double)e[i] > thr;
  __syncthreads();
  double cnt = 0;
  for (int i = t; i < nfr; i += FE_C) {
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
__global__ __launch_bounds__(FE_C) void fe_quality_kernel(const float* __restrict__ wave, int T, const float* __restrict__ mags,
                                                           const double* __restrict__ descs, int F, int nf,
                                                           const float* __restrict__ rms_long, int nfr_long,
                                                           const double* __restrict__ speech_prob, const float* __restrict__ lid,
                                                           float* __restrict__ raw, float* __restrict__ metrics, int* __restrict__ decision) {
  __shared__ double red[FE_C];
  const int b = blockIdx.x, t = threadIdx.x, NB = NFFT / 2 + 1;
  const float* x = wave + (long long)b * T;
  float mx = 0.f;
  for (int i = t; i < T; i += FE_C) mx = fmaxf(mx, fabsf(x[i]));
  mx = (float)block_max((double)mx, red);
  double cnt = 0;
  for (int i = t; i < T; i += FE_C) {
    const float v = mx > 0.f ? __fdiv_rn(x[i], mx) : x[i];        // float32 division, as numpy does on the float32 clip
    cnt += fabsf(v) > 0.95f;
  }
  const double clip = block_sum(cnt, red) / T * 100.0;
  double snr = 50.0;
  if (nf > 0) {
    const float* m = mags + (long long)b * F * NB;
    double sp = 0, np_ = 0;
    for (int k = t; k < NB; k += FE_C) {
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
  for (int f = t; f < F; f += FE_C) { c0 += d[f * 3]; c1 += d[f * 3 + 1]; c2 += d[f * 3 + 2]; }
  const double cen = block_sum(c0, red) / F, bw = block_sum(c1, red) / F, roll = block_sum(c2, red) / F;
  const float* rl = rms_long + (long long)b * nfr_long;
  double s = 0;
  for (int i = t; i < nfr_long; i += FE_C) s += rl[i];
  const double rmean = block_sum(s, red) / nfr_long;
  double dv = 0;
  for (int i = t; i < nfr_long; i += FE_C) dv += (rl[i] - rmean) * (rl[i] - rmean);
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
  hipLaunchKernelGGL(fe_vad_kernel, dim3(B), dim3(FE_C), 0, st, l.rms_vad, l.nfr_vad, l.speech);
  hipLaunchKernelGGL(fe_fft_kernel<2048>, dim3(l.F, B), dim3(FE_T), 0, st, (const void*)wave, 0, (long long)T, T, 512, 1024,
                     pad_reflect, 0, l.F, fs, tab, (double*)nullptr, l.mag, l.desc);
  const int nf = (int)(0.1 * (double)l.F);
  hipLaunchKernelGGL(fe_quality_kernel, dim3(B), dim3(FE_C), 0, st, wave, T, l.mag, l.desc, l.F, nf, l.rms_long, l.nfr_long,
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
    hipLaunchKernelGGL(fe_welch_kernel, dim3(B), dim3(FE_C), 0, st, l.power, nseg, N, fs, tab, mode, l.st);
  };
  welch(2048, l.nseg2048, WELCH_HUM);                                   // 1. hum notch
  for (int which = 0; which < 2; ++which)
    hipLaunchKernelGGL(fe_filtfilt_kernel, dim3(B), dim3(FE_IIR_T), 0, st, l.x, T, which, fs, l.st, l.tmp, l.tmp_stride);
  welch(2048, l.nseg2048, WELCH_HPF);                                   // 2. high-pass
  hipLaunchKernelGGL(fe_filtfilt_kernel, dim3(B), dim3(FE_IIR_T), 0, st, l.x, T, 2, fs, l.st, l.tmp, l.tmp_stride);
  hipLaunchKernelGGL(fe_snr_kernel, dim3(B), dim3(FE_C), 0, st, l.x, T, 0, l.st);   // 3. adaptive denoise
  welch(1024, l.nseg1024, WELCH_NOISE);
  const int M = 2 * (int)(0.1 * (double)T);
  hipLaunchKernelGGL(fe_wiener_kernel, dim3(B), dim3(FE_IIR_T), 0, st, l.x, T, M, l.st, l.P1, l.P2);
  hipLaunchKernelGGL(fe_snr_kernel, dim3(B), dim3(FE_C), 0, st, l.x, T, 1, l.st);
  hipLaunchKernelGGL(fe_finish_kernel, dim3(B), dim3(FE_C), 0, st, l.x, T, fs, l.st, out, c_raw, c_meta);   // 4./5. + features
  SER_LAUNCH_CHECK();
  return SER_OK;
}
