// What a PERSISTENT decode-layer kernel could recover, measured before building one (round 4).
//
//     hipcc -O3 --offload-arch=gfx950 tools/persist_probe.hip -o tools/build/persist_probe && tools/build/persist_probe
//
// The captured decode step spends ~85 us per layer beyond what the per-CU memory-pipe law (tools/cu_pipe_probe.hip) prices for
// its projections: launch ramps, tails, slab bursts -- one boundary per kernel, ~10 kernels per layer.  A persistent kernel keeps
// the WEIGHT stream running across those boundaries (weights depend on nothing) and replaces each boundary by a grid-wide
// barrier.  Three numbers decide whether that pays:
//   A  what a launch boundary costs a streaming kernel: the same 335 MB (256 workgroups, 1.31 MB each) read by 1, 2, 4, 8, 16
//      back-to-back launches;
//   B  what a grid-wide barrier costs: 256 co-resident workgroups (one per CU), one device-scope counter, R rounds; flat and
//      two-level (one counter per group of 32 workgroups, then one across the 8 groups);
//   C  whether the stream survives barriers: three waves of every workgroup stream their window without interruption while
//      the fourth takes part in a barrier every `gap` microseconds;
//   D  where re-read bytes come from when they no longer fit L2: a 64 / 128 / 192 MiB window read four times in a row by 256
//      workgroups (pass 1 cold, passes 2-4 from the Infinity Cache if it kept them).
// Every spin is bounded: a barrier that does not complete in ~20 ms sets a flag and the kernel runs on (no hang).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;
constexpr int kU = 16;
constexpr int kLdsKeepAlone = 100 * 1024;  // one workgroup per CU
constexpr long long kSpinLimit = 4000000;  // ~20 ms of polling

// ---- A / D: plain streaming of [first, first + packs) 16-byte packs of the workgroup's window ---------------------------------
template <bool NT>
__global__ __launch_bounds__(kThreads) void stream_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ sink,
                                                          int64_t window_packs, int64_t first, int64_t packs) {
  extern __shared__ unsigned char lds[];
  const u32x4* mine = src + (int64_t)blockIdx.x * window_packs + first;
  u32x4 acc = {0, 0, 0, 0};
  const int64_t steps = packs / (kThreads * kU);
  for (int64_t s = 0; s < steps; ++s) {
    u32x4 v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const u32x4* p = mine + (s * kU + u) * kThreads + threadIdx.x;
      v[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x * kThreads + threadIdx.x] = acc.x;
  if (threadIdx.x == 0) lds[0] = 0;
}

// ---- B: grid barrier ---------------------------------------------------------------------------------------------------------
// counters: [0] flat / top-level, [16 (g + 1)] group g (64-byte apart).  Monotonic targets: round r waits for (r + 1) * members.
__device__ __forceinline__ bool wait_at_least(unsigned* ctr, unsigned target, unsigned* timed_out) {
  long long spins = 0;
  while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kSpinLimit) {
      *timed_out = 1;
      return false;
    }
  }
  return true;
}

__device__ __forceinline__ void grid_barrier(unsigned* ctr, int round, int G, int two_level, unsigned* timed_out) {
  // called by ONE lane per workgroup
  if (!two_level) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    wait_at_least(ctr, (unsigned)(round + 1) * G, timed_out);
  } else {
    const int groups = (G + 31) / 32;
    const int gi = two_level == 2 ? (int)(blockIdx.x / 32) : (int)(blockIdx.x & 7);  // 2: consecutive ids, 1: same XCD (id % 8)
    const int ngroups = two_level == 2 ? groups : (G < 8 ? G : 8);
    const int members = two_level == 2 ? (G - gi * 32 < 32 ? G - gi * 32 : 32) : (G - gi + 7) / 8;
    unsigned* gc = ctr + 16 * (gi + 1);
    const unsigned before = __hip_atomic_fetch_add(gc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1 == (unsigned)(round + 1) * members)  // last of the group reports upward
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    wait_at_least(ctr, (unsigned)(round + 1) * ngroups, timed_out);
  }
}

__global__ __launch_bounds__(kThreads) void barrier_kernel(unsigned* ctr, unsigned* timed_out, int rounds, int G, int two_level) {
  extern __shared__ unsigned char lds[];
  for (int r = 0; r < rounds; ++r) {
    if (threadIdx.x == 0) grid_barrier(ctr, r, G, two_level, timed_out);
    __syncthreads();
  }
  if (threadIdx.x == 0) lds[0] = 0;
}

// ---- C: waves 0-2 stream, wave 3 takes a barrier every `gap_clocks` -------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void stream_with_barriers_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ sink,
                                                                        int64_t window_packs, unsigned* ctr, unsigned* timed_out,
                                                                        unsigned* stream_ticks, int G, long long gap_ticks,
                                                                        int max_rounds) {
  extern __shared__ unsigned char lds[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wv < 3) {
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    const u32x4* mine = src + (int64_t)blockIdx.x * window_packs;
    u32x4 acc = {0, 0, 0, 0};
    const int64_t steps = window_packs / (192 * kU);
    for (int64_t s = 0; s < steps; ++s) {
      u32x4 v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) v[u] = __builtin_nontemporal_load(mine + (s * kU + u) * 192 + wv * 64 + lane);
#pragma unroll
      for (int u = 0; u < kU; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x * kThreads + threadIdx.x] = acc.x;
    if (lane == 0) atomicMax(stream_ticks + blockIdx.x, (unsigned)(__builtin_amdgcn_s_memrealtime() - t_begin));
  } else if (lane == 0 && gap_ticks > 0) {
    // every workgroup takes the SAME number of rounds (max_rounds): the barrier count cannot depend on local progress
    for (int r = 0; r < max_rounds; ++r) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < gap_ticks) __builtin_amdgcn_s_sleep(8);
      grid_barrier(ctr, r, G, 0, timed_out);
    }
  }
  if (threadIdx.x == 0) lds[0] = 0;
}

template <typename F>
static float best_us(int reps, F&& launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CHECK(hipEventRecord(e0));
    launch(i);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms * 1e3f);
  }
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return best;
}

int main() {
  const int G = 256;
  const int64_t total = 12ll << 30;
  u32x4* buf;
  uint32_t* sink;
  unsigned *ctr, *flags, *ticks;
  CHECK(hipMalloc(&buf, total));
  CHECK(hipMalloc(&sink, G * kThreads * 4));
  CHECK(hipMalloc(&ctr, 4096));
  CHECK(hipMalloc(&flags, 64));
  CHECK(hipMalloc(&ticks, G * 4));
  CHECK(hipMemset(buf, 1, total));
  CHECK(hipMemset(flags, 0, 64));
  CHECK(hipDeviceSynchronize());
  CHECK(hipFuncSetAttribute((const void*)stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsKeepAlone));
  CHECK(hipFuncSetAttribute((const void*)stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsKeepAlone));
  CHECK(hipFuncSetAttribute((const void*)barrier_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsKeepAlone));
  CHECK(hipFuncSetAttribute((const void*)stream_with_barriers_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsKeepAlone));
  int64_t cursor = 0;  // byte offset of the next cold region
  auto cold = [&](int64_t bytes) {
    if (cursor + bytes > total) cursor = 0;
    const u32x4* p = buf + cursor / 16;
    cursor += bytes;
    return p;
  };

  // ---- A ----
  {
    const int64_t window = 1310720;  // 1.31 MB per workgroup = one 128-row tile of a K = 5120 bf16 matrix
    printf("A. 335 MB (256 workgroups x 1.31 MB, 16 loads of 16 B in flight per lane) read by P back-to-back launches\n");
    float one = 0;
    for (int P : {1, 2, 4, 8, 16}) {
      const int64_t packs = window / 16 / P;
      const float us = best_us(5, [&](int) {
        const u32x4* p = cold(window * G);
        for (int i = 0; i < P; ++i)
          stream_kernel<true><<<G, kThreads, kLdsKeepAlone>>>(p, sink, window / 16, i * packs, packs);
      });
      if (P == 1) one = us;
      printf("   P = %2d: %7.1f us  %5.2f TB/s   + %5.1f us  = %5.2f us per extra boundary\n", P, us, window * G / us / 1e6, us - one,
             P > 1 ? (us - one) / (P - 1) : 0.f);
      fflush(stdout);
    }
  }

  // ---- B ----
  {
    printf("B. grid-wide barrier, 256 co-resident workgroups (one per CU), 200 rounds in one launch\n");
    const char* names[] = {"flat: one device-scope counter", "two-level: per-XCD counter (ids i, i+8, ...), then 8", "two-level: 32 consecutive ids, then 8"};
    for (int mode = 0; mode < 3; ++mode) {
      const int rounds = 200;
      float us = best_us(3, [&](int) {
        CHECK(hipMemsetAsync(ctr, 0, 4096));
        barrier_kernel<<<G, kThreads, kLdsKeepAlone>>>(ctr, flags, rounds, G, mode);
      });
      float us0 = best_us(3, [&](int) {
        CHECK(hipMemsetAsync(ctr, 0, 4096));
        barrier_kernel<<<G, kThreads, kLdsKeepAlone>>>(ctr, flags, 0, G, mode);
      });
      unsigned to;
      CHECK(hipMemcpy(&to, flags, 4, hipMemcpyDeviceToHost));
      printf("   %-56s %6.2f us per barrier (launch of 0 rounds: %.1f us)%s\n", names[mode], (us - us0) / rounds, us0,
             to ? "  [TIMED OUT: not all workgroups co-resident?]" : "");
      fflush(stdout);
      if (to) CHECK(hipMemset(flags, 0, 64));
    }
  }

  // ---- C ----
  {
    const int64_t window = 8ll << 20;  // 8 MiB per workgroup: ~300 us of streaming
    printf("C. three waves per workgroup stream 8 MiB while the fourth joins a grid barrier every `gap`\n");
    const int64_t streamed = (window / 16 / (192 * kU)) * (192 * kU) * 16 * G;
    for (double gap_us : {0.0, 40.0, 20.0, 10.0, 5.0}) {
      const int max_rounds = gap_us > 0 ? (int)(250.0 / gap_us) : 0;
      float best_stream = 1e30f;
      const float us = best_us(3, [&](int) {
        CHECK(hipMemsetAsync(ctr, 0, 4096));
        CHECK(hipMemsetAsync(ticks, 0, G * 4));
        const u32x4* p = cold(window * G);
        stream_with_barriers_kernel<<<G, kThreads, kLdsKeepAlone>>>(p, sink, window / 16, ctr, flags, ticks, G,
                                                                   (long long)(gap_us * 100.0) /* 100 MHz counter */, max_rounds);
        std::vector<unsigned> h(G);
        CHECK(hipMemcpy(h.data(), ticks, G * 4, hipMemcpyDeviceToHost));
        best_stream = std::min(best_stream, *std::max_element(h.begin(), h.end()) / 100.f);
      });
      unsigned f[2];
      CHECK(hipMemcpy(f, flags, 8, hipMemcpyDeviceToHost));
      printf("   gap %5.1f us (%3d barriers): kernel %7.1f us, slowest workgroup's stream %7.1f us  %5.2f TB/s%s\n", gap_us, max_rounds,
             us, best_stream, streamed / best_stream / 1e6, f[0] ? "  [barrier timed out]" : "");
      fflush(stdout);
      CHECK(hipMemset(flags, 0, 64));
    }
  }

  // ---- D ----
  {
    printf("D. a window read four times in a row by 256 workgroups (each its own share; plain loads / nontemporal loads)\n");
    for (int64_t mib : {64, 128, 192, 512}) {
      for (int nt = 0; nt < 2; ++nt) {
        const int64_t window = (mib << 20) / G;
        const u32x4* p = cold(window * G);
        float t[4];
        for (int pass = 0; pass < 4; ++pass)
          t[pass] = best_us(1, [&](int) {
            if (nt)
              stream_kernel<true><<<G, kThreads, kLdsKeepAlone>>>(p, sink, window / 16, 0, window / 16);
            else
              stream_kernel<false><<<G, kThreads, kLdsKeepAlone>>>(p, sink, window / 16, 0, window / 16);
          });
        printf("   %3lld MiB %-5s: pass 1 %7.1f us (%5.2f TB/s)   passes 2-4: %7.1f %7.1f %7.1f us (%5.2f TB/s)\n", (long long)mib,
               nt ? "nt" : "plain", t[0], (mib << 20) / t[0] / 1e6, t[1], t[2], t[3], (mib << 20) / t[3] / 1e6);
        fflush(stdout);
      }
    }
  }
  return 0;
}
