// What ONE compute unit can pull through its vector-memory pipe, by where the bytes come from (round 4).
//
//     hipcc -O3 --offload-arch=gfx950 tools/cu_pipe_probe.hip -o tools/build/cu_pipe_probe && tools/build/cu_pipe_probe
//
// The full-batch projection kernels (csrc/gemm_g3.hip) cost  weights / ~6 TB/s  +  activation re-reads / ~21 TB/s  -- the two
// terms ADD (profiles/r03a_g3_bench_and_ablation.json), with LDS-DMA or register staging, with loads issued by one wave or
// by several (profiles/r04a_g3_store_policy_and_loader_roles.json).  This probe isolates the mechanism: G workgroups, one
// per CU (100 KB of LDS each keeps a second one off the CU), each streaming its OWN bytes with U 16-byte loads in flight per
// lane, from
//     hbm   a private cold region (every launch a new one: nothing is in L2 or the Infinity Cache),
//     l2    a 1-MiB region shared by all workgroups and re-read 16 times (L2 hits after the first touch),
//     mix   2 L2 bytes per HBM byte (the x : w ratio of a 256 x 128 projection tile).
// If a CU's rate is set by its own request queue (bytes in flight / latency) the per-CU figure is FLAT in G until the
// chip-level limit (HBM ~6.3 TB/s, L2 ~35 TB/s) is reached, differs between hbm and l2 by the latency ratio, does not move
// with U once the queue is full, and the mixed stream takes the SUM of the two times.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
      exit(1);                                                                            \
    }                                                                                     \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;

// Each lane keeps U loads in flight; `hbm_per_block` / `l2_bytes` bytes per pass.  mode bit 0: hbm stream, bit 1: l2 stream.
template <int U>
__global__ __launch_bounds__(kThreads) void probe_kernel(const u32x4* __restrict__ hbm, const u32x4* __restrict__ l2,
                                                         uint32_t* __restrict__ sink, int64_t hbm_packs_per_block,
                                                         int64_t l2_packs, int l2_passes, int mode) {
  extern __shared__ unsigned char lds[];  // only to keep ONE workgroup per CU
  const int tid = threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  const u32x4* mine = hbm + (int64_t)blockIdx.x * hbm_packs_per_block;
  const int64_t hsteps = (mode & 1) ? hbm_packs_per_block / (kThreads * U) : 0;
  const int64_t lsteps_total = (mode & 2) ? (l2_packs / (kThreads * U)) * l2_passes : 0;
  const int64_t lper = l2_packs / (kThreads * U);
  // interleave so that the two streams finish together: r = l2 steps per hbm step
  const int64_t r = hsteps > 0 ? (lsteps_total + hsteps - 1) / hsteps : 0;
  int64_t li = 0;
  const int64_t outer = hsteps > 0 ? hsteps : lsteps_total;
  for (int64_t s = 0; s < outer; ++s) {
    u32x4 v[U];
    if (hsteps > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(mine + (s * U + u) * kThreads + tid);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
      for (int64_t k = 0; k < r && li < lsteps_total; ++k, ++li) {
        const int64_t base = (li % lper) * U;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = l2[(base + u) * kThreads + tid];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
      }
    } else {
      const int64_t base = (s % lper) * U;
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = l2[(base + u) * kThreads + tid];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x * kThreads + tid] = acc.x;  // keep the loads alive
  if (tid == 0 && lds[0] == 77) sink[0] = 1;
}

// The weight stream of a 256 x 128 projection tile as csrc/gemm_g3.hip issues it: per 64-k step 128 rows x 128 B, rows
// `ld` bytes apart (K = 5120: 10240 B), next step = the next 128 B of every row -- against the same bytes laid out
// contiguously per tile (16 KB per step, steps consecutive).  CHUNK = bytes of one row read per step (128: k-step 64;
// 256 / 512: what a deeper k-step or a [tile][step] packed layout would give).  Each workgroup streams `rows` x `ld_used`
// bytes of its own rows; U loads in flight per lane.
template <int U, int CHUNK>
__global__ __launch_bounds__(kThreads) void tile_kernel(const unsigned char* __restrict__ w, uint32_t* __restrict__ sink,
                                                        int64_t ld, int rows, int64_t row_bytes, int64_t tile_stride) {
  extern __shared__ unsigned char lds[];
  const int tid = threadIdx.x;
  constexpr int kLanesPerRow = CHUNK / 16;             // lanes covering one row's chunk
  constexpr int kRowsPerLoad = kThreads / kLanesPerRow;  // rows one 256-lane load instruction covers
  const int r0 = tid / kLanesPerRow, c = (tid % kLanesPerRow) * 16;
  const unsigned char* base = w + (int64_t)blockIdx.x * tile_stride;
  u32x4 acc = {0, 0, 0, 0};
  const int loads_per_step = rows / kRowsPerLoad;  // instructions per step
  const int64_t steps = row_bytes / CHUNK;
  // flatten (step, load) and keep U in flight
  const int64_t total = steps * loads_per_step;
  for (int64_t i = 0; i < total; i += U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u < total ? i + u : total - 1;
      const int64_t st = j / loads_per_step;
      const int q = (int)(j - st * loads_per_step);
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (int64_t)(q * kRowsPerLoad + r0) * ld + st * CHUNK + c));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x * kThreads + tid] = acc.x;
  if (tid == 0 && lds[0] == 77) sink[0] = 1;
}

template <int CHUNK>
static float run_tile(int G, const unsigned char* w, int64_t total_bytes, uint32_t* sink, int64_t ld, int rows, int64_t row_bytes,
                      int64_t tile_stride, int64_t launch_span, int reps, int64_t* cursor) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_kernel<16, CHUNK>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10));
  float best = 1e30f;
  for (int i = 0; i < reps + 1; ++i) {
    if (*cursor + launch_span > total_bytes) *cursor = 0;
    const unsigned char* win = w + *cursor;
    *cursor += launch_span;
    CHECK(hipEventRecord(e0));
    tile_kernel<16, CHUNK><<<dim3(G), dim3(kThreads), 100 << 10>>>(win, sink, ld, rows, row_bytes, tile_stride);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (i > 0 && ms < best) best = ms;
  }
  return best * 1e3f;
}

template <int U>
static float run(int G, int mode, const u32x4* hbm, int64_t hbm_bytes_total, const u32x4* l2, uint32_t* sink,
                 int64_t hbm_per_block, int64_t l2_bytes, int l2_passes, int reps, int64_t* cursor) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel<U>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10));
  float best = 1e30f;
  for (int i = 0; i < reps + 1; ++i) {
    // a fresh cold window of the big buffer for every launch
    const int64_t need = (int64_t)G * hbm_per_block;
    if (*cursor + need > hbm_bytes_total) *cursor = 0;
    const u32x4* win = hbm + *cursor / 16;
    *cursor += need;
    CHECK(hipEventRecord(e0));
    probe_kernel<U><<<dim3(G), dim3(kThreads), 100 << 10>>>(win, l2, sink, hbm_per_block / 16, l2_bytes / 16, l2_passes, mode);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (i > 0 && ms < best) best = ms;
  }
  return best * 1e3f;  // us
}

int main() {
  const int64_t hbm_total = 12ll << 30;     // 12 GiB: 256 blocks x 8 MiB windows never repeat within the Infinity Cache's reach
  const int64_t hbm_per_block = 8ll << 20;  // 8 MiB per workgroup and launch
  const int64_t l2_bytes = 1ll << 20;       // shared, L2-resident
  const int l2_passes = 16;                 // 16 MiB per workgroup from L2
  u32x4 *hbm, *l2;
  uint32_t* sink;
  CHECK(hipMalloc(&hbm, hbm_total));
  CHECK(hipMalloc(&l2, l2_bytes));
  CHECK(hipMalloc(&sink, 256 * kThreads * 4));
  CHECK(hipMemset(hbm, 1, hbm_total));
  CHECK(hipMemset(l2, 2, l2_bytes));
  CHECK(hipDeviceSynchronize());
  int64_t cursor = 0;
  const int Gs[] = {1, 8, 32, 64, 128, 256};
  printf("one workgroup (256 lanes) per CU; per-CU rate in B/ns (= GB/s), aggregate in TB/s; best of 5 launches\n");
  printf("%-5s %-4s | %-28s | %-28s | %-44s\n", "G", "U", "hbm: us  B/ns/CU  TB/s", "l2: us  B/ns/CU  TB/s", "mix (8 MiB hbm + 16 MiB l2): us  sum of the two  ratio");
  for (int G : Gs) {
    for (int U : {4, 16}) {
      float th, tl, tm;
      if (U == 4) {
        th = run<4>(G, 1, hbm, hbm_total, l2, sink, hbm_per_block, l2_bytes, l2_passes, 5, &cursor);
        tl = run<4>(G, 2, hbm, hbm_total, l2, sink, hbm_per_block, l2_bytes, l2_passes, 5, &cursor);
        tm = run<4>(G, 3, hbm, hbm_total, l2, sink, hbm_per_block, l2_bytes, l2_passes, 5, &cursor);
      } else {
        th = run<16>(G, 1, hbm, hbm_total, l2, sink, hbm_per_block, l2_bytes, l2_passes, 5, &cursor);
        tl = run<16>(G, 2, hbm, hbm_total, l2, sink, hbm_per_block, l2_bytes, l2_passes, 5, &cursor);
        tm = run<16>(G, 3, hbm, hbm_total, l2, sink, hbm_per_block, l2_bytes, l2_passes, 5, &cursor);
      }
      const double hb = (double)hbm_per_block, lb = (double)l2_bytes * l2_passes;
      printf("%-5d %-4d | %8.1f %8.1f %8.2f   | %8.1f %8.1f %8.2f   | %8.1f %10.1f %8.2f\n", G, U, th, hb / th / 1e3,
             hb * G / th / 1e6, tl, lb / tl / 1e3, lb * G / tl / 1e6, tm, th + tl, tm / (th + tl));
      fflush(stdout);
    }
  }
  // ---- access pattern of the weight stream (G = 256 workgroups, one 128-row tile of a [N, K = 5120] bf16 matrix each)
  {
    const int G = 256, rows = 128;
    const int64_t K2 = 10240;  // bytes per weight row
    const int64_t tile_bytes = rows * K2, span = (int64_t)G * tile_bytes;  // 335 MB per launch
    auto line = [&](const char* what, float us) {
      printf("%-92s %8.1f us  %6.2f TB/s\n", what, us, (double)span / us / 1e6);
      fflush(stdout);
    };
    printf("\nweight-stream access pattern, 256 workgroups x one 128-row tile (1.31 MB) each, 16 loads in flight per lane:\n");
    line("row-major [N][K] as gemm_g3 reads it: 128 rows x 128 B per step, rows 10240 B apart",
         run_tile<128>(G, (const unsigned char*)hbm, hbm_total, sink, K2, rows, K2, tile_bytes, span, 5, &cursor));
    line("row-major, 256 B of every row per step (k-step 128)",
         run_tile<256>(G, (const unsigned char*)hbm, hbm_total, sink, K2, rows, K2, tile_bytes, span, 5, &cursor));
    line("row-major, 512 B of every row per step (k-step 256)",
         run_tile<512>(G, (const unsigned char*)hbm, hbm_total, sink, K2, rows, K2, tile_bytes, span, 5, &cursor));
    // packed [tile][step][128 rows][128 B]: the same bytes, contiguous per workgroup (ld = 128 B: row r of a step at + 128 r,
    // the next step 16 KB further: modelled as rows = 128 * steps of 128 B each, one step)
    line("packed [tile][k-step][128 rows][128 B]: each workgroup's 1.31 MB contiguous",
         run_tile<128>(G, (const unsigned char*)hbm, hbm_total, sink, 128, (int)(rows * (K2 / 128)), 128, tile_bytes, span, 5, &cursor));
  }
  return 0;
}
