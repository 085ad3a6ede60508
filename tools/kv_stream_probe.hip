// Probe (round 6): how fast can ONE decode-attention launch's K/V bytes be pulled out of a paged pool, by request shape,
// cache policy and landing place?  Decides whether the matrix-core decode kernel should keep its register-staged
// 16-row x 64-B requests (default policy) or move to full-line requests / LDS-DMA with the nt policy.
//
//     hipcc -O3 --offload-arch=gfx950 tools/kv_stream_probe.hip -o /tmp/kv_stream_probe && /tmp/kv_stream_probe
//
// Geometry = Qwen3-14B TP1 decode at B = 256: token row = 8 kv heads x 256 B = 2 KB, K and V in separate slabs, pages of
// 256 tokens handed out in shuffled order, 256 workgroups x 8 waves (one workgroup per CU, wave = kv head), every
// workgroup streams `tiles` 16-token tiles (58 = the bench's 929 tokens per slot; 128 = the asymptote).  Variants:
//   regM   registers, wave-load = 16 token rows x 64 B of the wave's head (the matrix-core kernel's A-operand shape)
//   regK   registers, wave-load = 4 token rows x 256 B of the wave's head (whole 128-B lines; the streaming kernel's shape)
//   regT   registers, wave-load = 1 KB contiguous of the workgroup's 32-KB tile (wave w owns KB 4w .. 4w + 3)
//   dmaK   as regK, landing in a wave-private LDS ring (global_load_lds_dwordx4), read back with ds_read_b128
//   dmaT   as regT through LDS-DMA, one s_barrier per tile (every wave then reads ITS head's rows of the tile from LDS)
// each with the default policy and with nt (aux = 2).  One tile (8 loads per lane) in flight per wave while the previous
// one is consumed -- the depth of the product kernel.  Consumption = XOR fold (nothing elided).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <random>
#include <vector>

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

typedef uint32_t V4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_t;
typedef __attribute__((address_space(3))) V4 lds_v4;
typedef __attribute__((address_space(4))) int CInt;

constexpr int kRow = 2048;       // bytes per token row (8 heads x 128 x bf16)
constexpr int kTileBytes = 16 * kRow;
constexpr int kPageTiles = 16;   // 256-token pages

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, -1, 0x00020000);
}

enum { REG_M = 0, REG_K = 1, REG_T = 2, DMA_K = 3, DMA_T = 4 };

template <int PAT, int AUX>
__global__ __launch_bounds__(512, 2) void rd(const char* __restrict__ kbase, const char* __restrict__ vbase,
                                             const int* __restrict__ page_of, int tiles_in, uint32_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int slot = blockIdx.x;
  // skew experiment (argv[2], percent in bits 16.. of tiles_in): the first wave of every SIMD (w < 4) streams (100 + skew) %
  // of the tiles, the second (100 - skew) % -- does handing the arbitration's favourite more work shorten the launch?
  const int skew = tiles_in >> 16, base_tiles = tiles_in & 0xffff;
  const int tiles = (PAT <= REG_T || PAT == DMA_K) ? base_tiles * (w < 4 ? 100 + skew : 100 - skew) / 100 : base_tiles;
  const CInt* pages = (const CInt*)page_of;
  // byte offset of the lane's piece inside a tile, per load j
  int off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (PAT == REG_M) off[j] = (lane & 15) * kRow + w * 256 + j * 64 + (lane >> 4) * 16;
    else if (PAT == REG_K || PAT == DMA_K) off[j] = (4 * j + (lane >> 4)) * kRow + w * 256 + (lane & 15) * 16;
    else off[j] = (4 * w + j) * 1024 + lane * 16;
  }
  auto tile_base = [&](int ti) -> int64_t {
    const int g = slot * base_tiles + ti;
    const int page = pages[g / kPageTiles];
    return ((int64_t)page * kPageTiles + (g % kPageTiles)) * kTileBytes;
  };
  uint32_t acc = 0;
  if constexpr (PAT <= REG_T) {
    V4 a[8], b[8];
    auto load = [&](V4* t, int ti) {
      const int64_t tb = tile_base(ti < tiles ? ti : tiles - 1);
      const __amdgpu_buffer_rsrc_t kd = rsrc_of(kbase + tb), vd = rsrc_of(vbase + tb);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[2 * j] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(kd, off[j], 0, AUX));
        t[2 * j + 1] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(vd, off[j], 0, AUX));
      }
    };
    auto fold = [&](V4* t) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("" : "+v"(t[j]));
        acc ^= t[j].x ^ t[j].y ^ t[j].z ^ t[j].w;
      }
    };
    load(a, 0);
    for (int ti = 0; ti < tiles; ti += 2) {
      load(b, ti + 1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      fold(a);
      load(a, ti + 2);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      fold(b);
    }
    fold(a);
  } else if constexpr (PAT == DMA_K) {
    // wave-private ring: 2 stages x (K 4 KB + V 4 KB)
    unsigned char* mine = lds + w * 16384;
    auto issue = [&](int stage, int ti) {
      const int64_t tb = tile_base(ti < tiles ? ti : tiles - 1);
      const __amdgpu_buffer_rsrc_t kd = rsrc_of(kbase + tb), vd = rsrc_of(vbase + tb);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(kd, (lds_void_t)(mine + stage * 8192 + j * 1024), 16, off[j], 0, 0, AUX);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vd, (lds_void_t)(mine + stage * 8192 + 4096 + j * 1024), 16, off[j], 0, 0, AUX);
      }
    };
    auto fold = [&](int stage) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const V4 t = *(lds_v4*)(mine + stage * 8192 + j * 1024 + lane * 16);
        acc ^= t.x ^ t.y ^ t.z ^ t.w;
      }
    };
    issue(0, 0);
    for (int ti = 0; ti < tiles; ti += 2) {
      issue(1, ti + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      fold(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue(0, ti + 2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      fold(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // workgroup ring: 2 stages x (K 32 KB + V 32 KB); wave w issues KB 4w .. 4w + 3 of each, reads head w's rows
    auto issue = [&](int stage, int ti) {
      const int64_t tb = tile_base(ti < tiles ? ti : tiles - 1);
      const __amdgpu_buffer_rsrc_t kd = rsrc_of(kbase + tb), vd = rsrc_of(vbase + tb);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(kd, (lds_void_t)(lds + stage * 65536 + (4 * w + j) * 1024), 16, off[j], 0, 0, AUX);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vd, (lds_void_t)(lds + stage * 65536 + 32768 + (4 * w + j) * 1024), 16, off[j], 0, 0, AUX);
      }
    };
    auto fold = [&](int stage) {
      // head w of token (lane & 15), chunk rotated by the token: a 16-lane pass covers all banks (the product applies the
      // same permutation on the DMA source address instead)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int tok = lane & 15;
        const int chunk = ((4 * j + (lane >> 4)) ^ tok) & 15;
        const V4 t = *(lds_v4*)(lds + stage * 65536 + tok * kRow + w * 256 + chunk * 16);
        const V4 u = *(lds_v4*)(lds + stage * 65536 + 32768 + tok * kRow + w * 256 + chunk * 16);
        acc ^= t.x ^ t.y ^ t.z ^ t.w ^ u.x ^ u.y ^ u.z ^ u.w;
      }
    };
    issue(0, 0);
    for (int ti = 0; ti < tiles; ti += 2) {
      issue(1, ti + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      fold(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue(0, ti + 2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      fold(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (acc == 0x12345678u) out[slot * 512 + threadIdx.x] = acc;
}

struct Case {
  const char* name;
  void (*kernel)(const char*, const char*, const int*, int, uint32_t*);
  int lds;
};

int main(int argc, char** argv) {
  const int G = 256;
  const int rounds = argc > 1 ? atoi(argv[1]) : 3;
  const int skew = argc > 2 ? atoi(argv[2]) : 0;
  const int tile_counts[2] = {58, 128};
  const int max_tiles = 128;
  const int n_pages = (G * max_tiles + max_tiles + kPageTiles - 1) / kPageTiles + 2;  // (+ one slot: a skewed wave runs past its own)
  const size_t slab = (size_t)n_pages * kPageTiles * kTileBytes;
  char *k, *v;
  int* pages;
  uint32_t* out;
  CHECK(hipMalloc(&k, slab));
  CHECK(hipMalloc(&v, slab));
  CHECK(hipMemset(k, 1, slab));
  CHECK(hipMemset(v, 2, slab));
  CHECK(hipMalloc(&pages, n_pages * sizeof(int)));
  CHECK(hipMalloc(&out, G * 512 * sizeof(uint32_t)));
  std::vector<int> perm(n_pages);
  for (int i = 0; i < n_pages; ++i) perm[i] = i;
  std::mt19937 rng(0);
  std::shuffle(perm.begin(), perm.end(), rng);
  CHECK(hipMemcpy(pages, perm.data(), n_pages * sizeof(int), hipMemcpyHostToDevice));

  const Case cases[] = {
      {"regM default", rd<REG_M, 0>, 0},     {"regM nt     ", rd<REG_M, 2>, 0},
      {"regK default", rd<REG_K, 0>, 0},     {"regK nt     ", rd<REG_K, 2>, 0},
      {"regT default", rd<REG_T, 0>, 0},     {"regT nt     ", rd<REG_T, 2>, 0},
      {"dmaK default", rd<DMA_K, 0>, 131072}, {"dmaK nt     ", rd<DMA_K, 2>, 131072},
      {"dmaT default", rd<DMA_T, 0>, 131072}, {"dmaT nt     ", rd<DMA_T, 2>, 131072},
  };
  const int n_cases = sizeof(cases) / sizeof(cases[0]);
  for (const Case& c : cases)
    if (c.lds > 65536) CHECK(hipFuncSetAttribute((const void*)c.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("K/V stream probe: 256 workgroups x 8 waves, 2-KB token rows, shuffled 256-token pages, one tile in flight per wave\n");
  for (int tc = 0; tc < 2; ++tc) {
    const int tiles = tile_counts[tc];
    const double bytes = 2.0 * G * tiles * kTileBytes;
    std::vector<std::vector<double>> us(n_cases);
    for (int r = 0; r < rounds; ++r) {
      for (int ci = 0; ci < n_cases; ++ci) {
        const Case& c = cases[ci];
        for (int i = 0; i < 5; ++i) c.kernel<<<G, 512, c.lds>>>(k, v, pages, tiles | (skew << 16), out);
        const int iters = 20;
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) c.kernel<<<G, 512, c.lds>>>(k, v, pages, tiles | (skew << 16), out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        us[ci].push_back(ms * 1e3 / iters);
      }
    }
    printf("tiles per workgroup %d (%.0f MB per launch), skew %d %%\n", tiles, bytes / 1e6, skew);
    for (int ci = 0; ci < n_cases; ++ci) {
      std::sort(us[ci].begin(), us[ci].end());
      const double med = us[ci][us[ci].size() / 2];
      printf("  %s  %7.1f us  %5.2f TB/s   (rounds:", cases[ci].name, med, bytes / med / 1e6);
      for (double t : us[ci]) printf(" %.1f", t);
      printf(")\n");
    }
  }
  // marginal rate between the two sizes = the stream without the launch's fixed cost
  return 0;
}
