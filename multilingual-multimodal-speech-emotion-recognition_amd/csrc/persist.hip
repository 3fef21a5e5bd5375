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
