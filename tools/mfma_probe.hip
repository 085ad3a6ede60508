// Probe: cycles per v_mfma_f32_32x32x16_bf16 of ONE wave (and of 4 / 8 waves of a workgroup) on gfx950, as a function of how
// the accumulators are chained and of the LDS reads issued between the MFMAs -- the matrix segment of the prefill attention
// kernels runs at ~55 cycles per MFMA where the pipe's rate is 32 (tools/prefill_ablate.py "MFMA + LDS reads only").
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) char lds_char;
typedef int v4i __attribute__((ext_vector_type(4)));

// ACCS: accumulators used round-robin.  READS: 0 none, 1 = one ds_read_b128 per MFMA feeding an MFMA 4 later (A operand),
// 2 = two ds_read_b64_tr_b16 per MFMA feeding an MFMA 4 later.  All waves of the block run the same thing.
template <int ACCS, int READS>
__global__ __launch_bounds__(512) void probe(long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((int*)lds)[i] = i * 2654435761u;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  v4i bfrag = {lane, lane * 3, lane * 5, lane * 7};
  v4i af[8];
  for (int i = 0; i < 8; ++i) af[i] = v4i{lane + i, lane ^ i, i, 1};
  const int base = (lane & 31) * 256 + ((lane >> 5) * 16 ^ ((lane & 15) << 4));
  // READS == 2: the prefill kernels' V^T fragment addressing (row r4 of a 4-key unit, 64-B XOR swizzle on the row);
  // READS == 3: the same without the swizzle (rows 256 B apart: 4-way bank conflict)
  const int r4 = (lane & 15) >> 2;
  const int tbase = (lane >> 5) * 1024 + r4 * 256 + (READS == 2 ? ((r4 & 3) << 6) : 0) + (lane & 3) * 8 + ((lane >> 4) & 1) * 32;
  constexpr int N = 64;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    acc[i % ACCS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i & 7]), __builtin_bit_cast(bf16x8, bfrag),
                                                             acc[i % ACCS], 0, 0, 0);
    if constexpr (READS == 1) {
      af[(i + 4) & 7] = *reinterpret_cast<const v4i*>(lds + base + (i & 15) * 2048 % 32768 + (i & 1) * 8192);
    } else if constexpr (READS >= 2) {
      const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((lds_char*)lds + tbase + (i & 15) * 2048));
      const s16x4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((lds_char*)lds + tbase + (i & 15) * 2048 + 512));
      v4i v;
      v.x = __builtin_bit_cast(int2, r1).x; v.y = __builtin_bit_cast(int2, r1).y;
      v.z = __builtin_bit_cast(int2, r2).x; v.w = __builtin_bit_cast(int2, r2).y;
      af[(i + 4) & 7] = v;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  sink[threadIdx.x] = s;
  if (lane == 0) out[wave] = (t1 - t0);
}

template <int ACCS, int READS>
void run(long long* d_out, float* sink, int threads, const char* what) {
  long long h[8];
  for (int i = 0; i < 3; ++i) {
    probe<ACCS, READS><<<1, threads>>>(d_out, sink);
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  }
  long long mx = 0;
  for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
  printf("%-64s %d waves: %5.1f cycles per MFMA (slowest wave, 64 MFMAs)\n", what, threads / 64, mx / 64.0);
}

int main() {
  long long* d_out;
  float* sink;
  hipMalloc(&d_out, 64);
  hipMalloc(&sink, 4096);
  for (int threads : {64, 256, 512}) {
    run<1, 0>(d_out, sink, threads, "1 accumulator (dependent chain), no LDS reads");
    run<2, 0>(d_out, sink, threads, "2 accumulators alternating, no LDS reads");
    run<4, 0>(d_out, sink, threads, "4 accumulators, no LDS reads");
    run<2, 1>(d_out, sink, threads, "2 accumulators, one ds_read_b128 per MFMA (used 4 later)");
    run<4, 1>(d_out, sink, threads, "4 accumulators, one ds_read_b128 per MFMA (used 4 later)");
    run<4, 2>(d_out, sink, threads, "4 accumulators, two ds_read_b64_tr_b16 per MFMA, swizzled rows");
    run<4, 3>(d_out, sink, threads, "4 accumulators, two ds_read_b64_tr_b16 per MFMA, rows 256 B apart");
  }
  return 0;
}
