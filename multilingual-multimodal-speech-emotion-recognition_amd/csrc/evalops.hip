// Eval-side consumers of the logits (ref src/eval.py:49-67, 192-206; src/utils.py:11-14): temperature scaling, softmax,
// arg-max, energy score  -logsumexp, and the temperature grid search of --calibrate.  Tiny [B, C] problems: one wave per
// row / one workgroup per temperature, so that the eval loop issues no torch arithmetic on the path.
#include "ser_common.h"

namespace {

constexpr int EV_MAXC = 64;     // classes per row handled by one lane each

// one wave per row: z = logits * inv_t; probs = softmax(z); pred = first arg-max; energy = -logsumexp(z)
__global__ __launch_bounds__(256) void eval_consumers_kernel(const float* __restrict__ logits, int B, int C, float inv_t,
                                                             float* __restrict__ probs, int64_t* __restrict__ pred,
                                                             float* __restrict__ energy) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  const float z = lane < C ? logits[(long long)row * C + lane] * inv_t : -INFINITY;
  const float m = wave_max(z);
  const float e = lane < C ? __expf(z - m) : 0.f;
  const float s = wave_sum(e);
  if (probs && lane < C) probs[(long long)row * C + lane] = e / s;
  // first index attaining the maximum (torch.argmax / torch.max semantics on ties)
  unsigned long long ball = __ballot(lane < C && z == m);
  if (lane == 0) {
    if (pred) pred[row] = (int64_t)(__ffsll((long long)ball) - 1);
    if (energy) energy[row] = -(m + __logf(s));
  }
}

// one workgroup per temperature: mean_n | max softmax(logits_n / t) - [argmax_n == label_n] |
__global__ __launch_bounds__(256) void temperature_grid_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                               int N, int C, const float* __restrict__ temps,
                                                               float* __restrict__ ece) {
  __shared__ double sh[4];
  const float inv_t = 1.0f / temps[blockIdx.x];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double acc = 0.0;
  for (int row = wave; row < N; row += 4) {
    const float z = lane < C ? logits[(long long)row * C + lane] * inv_t : -INFINITY;
    const float m = wave_max(z);
    const float e = lane < C ? __expf(z - m) : 0.f;
    const float s = wave_sum(e);
    const unsigned long long ball = __ballot(lane < C && z == m);
    const int arg = __ffsll((long long)ball) - 1;
    const float conf = 1.0f / s;                       // exp(m - m) / s
    acc += (double)fabsf(conf - (arg == (int)labels[row] ? 1.f : 0.f));
  }
  if (lane == 0) sh[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ece[blockIdx.x] = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) / (double)N);
}

}  // namespace

extern "C" int ser_eval_consumers(const float* logits, int B, int C, float temperature, float* probs, int64_t* pred, float* energy,
                                  void* stream) {
  SER_REQUIRE(logits && B > 0 && C > 0 && C <= EV_MAXC, "eval_consumers: B=%d C=%d (at most %d classes)", B, C, EV_MAXC);
  SER_REQUIRE(temperature > 0.f, "eval_consumers: temperature %f must be positive", temperature);
  hipLaunchKernelGGL(eval_consumers_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, logits, B, C, 1.0f / temperature,
                     probs, pred, energy);
  SER_LAUNCH_CHECK();
  return SER_OK;
}

extern "C" int ser_temperature_grid(const float* logits, const int64_t* labels, int N, int C, const float* temps, int G, float* ece,
                                    void* stream) {
  SER_REQUIRE(logits && labels && temps && ece && N > 0 && G > 0 && C > 0 && C <= EV_MAXC, "temperature_grid: bad argument");
  hipLaunchKernelGGL(temperature_grid_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, logits, labels, N, C, temps, ece);
  SER_LAUNCH_CHECK();
  return SER_OK;
}
