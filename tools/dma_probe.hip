// Probe: what does a wave pay between issuing an LDS-DMA load and (a) an `s_waitcnt lgkmcnt(0)` behind an ordinary
// ds_read, (b) the issue of further DMA pieces, on gfx950?  The question behind it: the prefill attention kernels spend
// ~1300 cycles per key tile around their four DMA pieces (tools/prefill_trace.py) -- is that the issue of the pieces, or
// an lgkmcnt wait that also waits for the DMA data?
//   hipcc --offload-arch=gfx950 tools/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
// One wave; `src` is 256 MB that nothing has touched (every piece misses all caches).  Stamps (s_memtime):
//   variant 0  t1: after N global_load_lds_dwordx4 pieces issued (no wait)       t2: after vmcnt(0)
//   variant 1  t1: after N pieces + one ds_read_b128 + s_waitcnt lgkmcnt(0)       t2: after vmcnt(0)
//   variant 2  as 0 with buffer_load_dwordx4 ... lds (MUBUF)                      variant 3: as 1 with MUBUF
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((address_space(3))) char lds_char;
typedef int v4i __attribute__((ext_vector_type(4)));

template <int VAR, int N>
__global__ void probe(const char* src, long long* out, int salt) {
  __shared__ __attribute__((aligned(1024))) char lds[N * 1024 + 1024];
  __shared__ __attribute__((aligned(16))) char other[1024];
  const int lane = threadIdx.x;
  other[lane * 16] = (char)lane;
  __syncthreads();
  const char* p = src + ((long)salt * N * 4096) + (lane >> 4) * 2048 + (lane & 15) * 16;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if constexpr (VAR < 2)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p + i * 8192), (lds_void_t*)(lds + i * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(lds + i * 1024), 16,
                                                (int)(salt * N * 4096 + (lane >> 4) * 2048 + (lane & 15) * 16 + i * 8192), 0, 0, 0);
  }
  v4i r = {0, 0, 0, 0};
  if constexpr (VAR & 1) {
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((int)(uintptr_t)(lds_char*)other + lane * 16) : "memory");
  }
  const long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = __builtin_readcyclecounter();
  if (lane == 0) {
    out[0] = t1 - t0;
    out[1] = t2 - t0;
    out[2] = r[0] + lds[5];
  }
}

// Throughput of the CU's DMA path with all 8 waves of one workgroup issuing: each wave sends 256 pieces (1 KB each) from a
// 512 KB region that stays in L2 (second pass timed), at most 8 in flight per wave.  ROWS = 1: a piece is 1 KB contiguous;
// ROWS = 4: four 256 B rows at a 2 KB stride (the K/V pool's [token][head][128] layout seen by one kv head).
template <int ROWS>
__global__ __launch_bounds__(512) void tput(const char* src, long long* out) {
  __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long t0 = 0;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    if (pass == 1) t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 256; ++i) {
      const int piece = (i * 8 + wave) & 511;  // 512 pieces = 512 KB
      const char* p = ROWS == 1 ? src + (long)piece * 1024 + lane * 16
                                : src + (long)piece * 8192 + (lane >> 4) * 2048 + (lane & 15) * 16;  // 4 MB region for ROWS = 4
      __builtin_amdgcn_global_load_lds((glb_void_t*)p, (lds_void_t*)(lds + (wave * 8 + (i & 7)) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = lds[7]; }
}

template <int VAR, int N>
void run(const char* src, long long* d_out, int& salt, const char* name) {
  long long h[3], a = 0, b = 0;
  const int reps = 8;
  for (int i = 0; i < reps; ++i) {
    probe<VAR, N><<<1, 64>>>(src, d_out, salt++);
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    if (i) { a += h[0]; b += h[1]; }
  }
  printf("%-58s N=%2d  t1 %6lld  t2 (data landed) %6lld cycles\n", name, N, a / (reps - 1), b / (reps - 1));
}

int main() {
  char* src;
  long long* d_out;
  hipMalloc(&src, 1ll << 30);
  hipMalloc(&d_out, 64);
  int salt = 0;
  run<0, 1>(src, d_out, salt, "global_load_lds: issue only");
  run<1, 1>(src, d_out, salt, "global_load_lds: + ds_read_b128 + lgkmcnt(0)");
  run<0, 4>(src, d_out, salt, "global_load_lds: issue only");
  run<1, 4>(src, d_out, salt, "global_load_lds: + ds_read_b128 + lgkmcnt(0)");
  run<0, 16>(src, d_out, salt, "global_load_lds: issue only");
  run<1, 16>(src, d_out, salt, "global_load_lds: + ds_read_b128 + lgkmcnt(0)");
  run<2, 4>(src, d_out, salt, "buffer_load lds: issue only");
  run<3, 4>(src, d_out, salt, "buffer_load lds: + ds_read_b128 + lgkmcnt(0)");
  run<2, 16>(src, d_out, salt, "buffer_load lds: issue only");
  run<3, 16>(src, d_out, salt, "buffer_load lds: + ds_read_b128 + lgkmcnt(0)");
  for (int rows = 1; rows <= 4; rows += 3) {
    long long h[2];
    for (int i = 0; i < 3; ++i) {
      if (rows == 1) tput<1><<<1, 512>>>(src, d_out); else tput<4><<<1, 512>>>(src, d_out);
      hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    }
    printf("one CU, 8 waves x 256 pieces from L2, %s: %lld cycles = %.1f cycles per 1 KB piece = %.1f B/clk\n",
           rows == 1 ? "1 KB contiguous" : "4 rows of 256 B at 2 KB stride", h[0], h[0] / 2048.0, 2048.0 * 1024 / h[0]);
  }
  return 0;
}
