// Probe: HBM read bandwidth of MI355X as a function of bytes in flight per CU, for the two access shapes of
// the paged decode attention kernel (what ceiling can a pure streaming reader reach?).
//   hipcc --offload-arch=gfx950 -O3 tools/read_bw_probe.hip -o /tmp/read_bw && /tmp/read_bw
// pattern L: every wave streams its own contiguous region, 1 KB per wave-load.
// pattern K: KV-pool rows of 2 KB (8 heads x 256 B); the 8 waves of a block are the 8 heads of one slot and
//            each wave-load reads 4 token rows x 256 B (16 lanes x 16 B per row) -- the decode kernel's shape.
// U = 16-byte loads in flight per lane (double-buffered: U issued before the previous U are consumed),
// W = waves per CU.  2 GiB buffer (8x the Infinity Cache), result folded into a checksum so nothing is elided.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t V4 __attribute__((ext_vector_type(4)));

template <int U, bool kKv>
__global__ __launch_bounds__(512) void rd(const V4* __restrict__ base, long steps, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  // element (16-B) index of load k of step s for this lane
  auto addr = [&](long s, int k) -> long {
    if constexpr (kKv) {
      const long slot = gw >> 3, h = gw & 7;
      const long tok = (slot * steps + s) * (4 * U) + 4 * k + (lane >> 4);
      return tok * 128 + h * 16 + (lane & 15);
    } else {
      return ((gw * steps + s) * U + k) * 64 + lane;
    }
  };
  V4 a[U], b[U];
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < U; ++k) a[k] = base[addr(0, k)];
  long s = 0;
  for (; s + 2 < steps; s += 2) {
#pragma unroll
    for (int k = 0; k < U; ++k) b[k] = base[addr(s + 1, k)];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < U; ++k) acc ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
#pragma unroll
    for (int k = 0; k < U; ++k) a[k] = base[addr(s + 2, k)];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < U; ++k) acc ^= b[k].x ^ b[k].y ^ b[k].z ^ b[k].w;
  }
#pragma unroll
  for (int k = 0; k < U; ++k) acc ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
  if (acc == 0x12345678u) out[gw] = acc;  // practically never: keeps the loads alive
}

template <int U, bool kKv>
static void run(const V4* buf, size_t bytes, int cus, int W, uint32_t* out) {
  const int wpb = 8;  // waves per block
  const long waves = (long)cus * W;
  const long blocks = waves / wpb;
  const long per_step = (long)U * 1024;  // bytes per wave per step
  long steps = (long)(bytes / (waves * per_step));
  steps -= steps & 1;
  if (steps < 4) return;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0));
    rd<U, kKv><<<dim3((unsigned)blocks), dim3(64 * wpb)>>>(buf, steps, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double moved = (double)waves * steps * per_step;
  printf("pattern %c  U=%2d  W=%2d  in-flight/CU %4d KB  %7.1f GB/s  (%.0f MB in %.3f ms)\n", kKv ? 'K' : 'L', U, W,
         U * W, moved / best / 1e6, moved / 1e6, best);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t bytes = (size_t)2 << 30;
  V4* buf;
  uint32_t* out;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&out, 1 << 20));
  CHECK(hipMemset(buf, 1, bytes));
  printf("device %s, %d CUs\n", prop.name, cus);
  const int Ws[] = {4, 8, 12, 16, 24, 32};
  for (int W : Ws) {
    run<1, false>(buf, bytes, cus, W, out);
    run<2, false>(buf, bytes, cus, W, out);
    run<4, false>(buf, bytes, cus, W, out);
    run<8, false>(buf, bytes, cus, W, out);
    if (W <= 16) run<16, false>(buf, bytes, cus, W, out);
  }
  for (int W : Ws) {
    run<1, true>(buf, bytes, cus, W, out);
    run<2, true>(buf, bytes, cus, W, out);
    run<4, true>(buf, bytes, cus, W, out);
    run<8, true>(buf, bytes, cus, W, out);
    if (W <= 16) run<16, true>(buf, bytes, cus, W, out);
  }
  return 0;
}
