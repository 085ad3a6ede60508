// Probe: cycles of ds_read_b128 / ds_write_b128 on gfx950 for candidate LDS layouts of the wstream GEMM's
// activation tile (which lanes may share a bank group in one access?).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
// Lane l = 16 kg + r reads the B fragment of row r, k-group kg.  Patterns (byte address of the lane's 16 B):
//   0 linear            l*16                         (reference: no conflicts possible)
//   1 pitch 144         r*144 + kg*16                (current layout)
//   2 pitch 128         r*128 + kg*16                (worst case reference)
//   3 pitch 128 + xor   r*128 + ((kg ^ (r&7))*16)    (8-chunk XOR swizzle)
//   4 k-major planes    kg*(16*16+16) + r*16         (chunk planes of 16 rows, 16-B pad between planes)
//   5 pitch 272         r*272 + kg*16                (the 128-k step variant)
//   6 pitch 144, kg*32  r*144 + kg*32                (k-groups two chunks apart)
//   7 pitch 160         r*160 + kg*16
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t V4 __attribute__((ext_vector_type(4)));

__global__ void probe(long* cycles, uint32_t* sink, const int* addr, int iters, int do_write) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = i;
  __syncthreads();
  const int a = addr[threadIdx.x];
  V4 acc = {0, 0, 0, 0};
  const long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (do_write) {
      *reinterpret_cast<V4*>(lds + a) = acc;
      acc.x += 1;
    } else {
      const V4 v = *reinterpret_cast<const volatile V4*>(lds + a);
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  }
  const long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
  sink[threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w ^ reinterpret_cast<uint32_t*>(lds)[threadIdx.x];
}

int main() {
  long* d_cyc; uint32_t* d_sink; int* d_addr;
  hipMalloc(&d_cyc, 8); hipMalloc(&d_sink, 256); hipMalloc(&d_addr, 256);
  const char* names[] = {"linear", "pitch144", "pitch128", "pitch128xor", "kmajor_planes", "pitch272", "pitch144_kg32", "pitch160"};
  for (int pat = 0; pat < 8; ++pat) {
    int h[64];
    for (int l = 0; l < 64; ++l) {
      const int r = l & 15, kg = l >> 4;
      switch (pat) {
        case 0: h[l] = l * 16; break;
        case 1: h[l] = r * 144 + kg * 16; break;
        case 2: h[l] = r * 128 + kg * 16; break;
        case 3: h[l] = r * 128 + ((kg ^ (r & 7)) * 16); break;
        case 4: h[l] = kg * (16 * 16 + 16) + r * 16; break;
        case 5: h[l] = r * 272 + kg * 16; break;
        case 6: h[l] = r * 144 + kg * 32; break;
        default: h[l] = r * 160 + kg * 16; break;
      }
    }
    hipMemcpy(d_addr, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w = 0; w < 2; ++w) {
      long best = 1L << 60;
      for (int rep = 0; rep < 3; ++rep) {
        probe<<<1, 64>>>(d_cyc, d_sink, d_addr, 4096, w);
        long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
        if (c < best) best = c;
      }
      printf("%-14s %s  %.2f cycles per access\n", names[pat], w ? "write" : "read ", (double)best / 4096);
    }
  }
  return 0;
}
