// Register layout of v_mfma_f32_4x4x4_16b_bf16 (16 blocks of D[4][4] = A[4][4] . B[4][4]) found by experiment: random small
// integers in, the result compared on the host against every (block, index) <- lane mapping worth considering.
//     hipcc -O3 --offload-arch=gfx950 tools/mfma444_probe.hip -o tools/build/mfma444_probe && tools/build/mfma444_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const s4* a, const s4* b, f4* d) {
  f4 c = {0, 0, 0, 0};
  d[threadIdx.x] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
}
static uint16_t bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
int main() {
  float A[64][4], B[64][4], D[64][4];
  uint16_t ha[64][4], hb[64][4];
  srand(1);
  for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
    A[l][q] = (float)(rand() % 7 - 3); B[l][q] = (float)(rand() % 5 - 2);
    ha[l][q] = bf(A[l][q]); hb[l][q] = bf(B[l][q]);
  }
  void *da, *db, *dd;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof D);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  k<<<1, 64>>>((const s4*)da, (const s4*)db, (f4*)dd);
  if (hipMemcpy(D, dd, sizeof D, hipMemcpyDeviceToHost) != hipSuccess) { printf("failed\n"); return 1; }
  // lane -> (block, idx): mode 0: (l / 4, l % 4); mode 1: (l % 16, l / 16)
  auto blk = [](int mode, int l) { return mode ? l % 16 : l / 4; };
  auto idx = [](int mode, int l) { return mode ? l / 16 : l % 4; };
  auto lane_of = [](int mode, int b, int i) { return mode ? b + 16 * i : 4 * b + i; };
  for (int ma = 0; ma < 2; ++ma) for (int mb = 0; mb < 2; ++mb) for (int md = 0; md < 2; ++md) for (int tr = 0; tr < 2; ++tr) {
    // D lane (block, j), register i  (tr = 1: lane (block, i), register j)
    bool ok = true;
    for (int l = 0; l < 64 && ok; ++l) for (int r = 0; r < 4 && ok; ++r) {
      const int b = blk(md, l), x = idx(md, l);
      const int i = tr ? x : r, j = tr ? r : x;
      float s = 0;
      for (int kk = 0; kk < 4; ++kk) s += A[lane_of(ma, b, i)][kk] * B[lane_of(mb, b, j)][kk];
      ok = s == D[l][r];
    }
    if (ok) printf("MATCH: A lane = %s, B lane = %s, D lane = %s holding %s in its 4 registers\n",
                   ma ? "block + 16 i" : "4 block + i", mb ? "block + 16 j" : "4 block + j",
                   md ? "block + 16 x" : "4 block + x", tr ? "column j (x = row i)" : "row i (x = column j)");
  }
  (void)blk; (void)idx;
  printf("done\n");
  return 0;
}
