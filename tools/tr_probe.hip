// Probe: which LDS elements does each lane receive from ds_read_b64_tr_b16 on gfx950?
//   hipcc --offload-arch=gfx950 tools/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
// LDS holds element i at 16-bit index i; every lane passes its own byte address.  Pattern 0: lane l reads
// byte l*8 (a linear [4][16] block per 16-lane group); pattern 1: row stride 256 B ([4 rows][16 cols] block of a
// row-major [R][128] image, group g at column 16 g); pattern 2: addresses permuted inside the group.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__global__ void probe(uint16_t* out, const int* addr_bytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int a = addr_bytes[threadIdx.x];
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((__attribute__((address_space(3))) char*)lds + a));
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = (uint16_t)r[i];
}

int main() {
  int h_addr[64];
  uint16_t h_out[256];
  int* d_addr;
  uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr));
  hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, j = l & 15;
      if (pat == 0) h_addr[l] = l * 8;
      else if (pat == 1) h_addr[l] = (j >> 2) * 256 + g * 32 + (j & 3) * 8;
      else h_addr[l] = ((j & 3)) * 256 + g * 32 + (j >> 2) * 8;  // lane j: row j&3, col chunk j>>2
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_out, d_addr);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d (lane: byte address -> 4 element indices received)\n", pat);
    for (int l = 0; l < 64; ++l)
      printf("  lane %2d addr %4d (elem %4d): %4d %4d %4d %4d\n", l, h_addr[l], h_addr[l] / 2, h_out[4 * l],
             h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
  }
  return 0;
}
