// Full-batch decode projection GEMM (128 < M <= 256):
//     out[M, N] = x[M, K] . w[N, K]^T        (torch F.linear layout, bf16 / fp16, fp32 accumulate)
//
// At M = 256 a projection of a 14B-class model is 256 flop per weight byte: the weight stream (HBM) and the MFMA
// pipe are both within 2x of their limits, and the shapes (N / 128 = 272, 56, 40 tiles for 256 CUs) quantise badly
// on any fixed tile grid.  This kernel is built around exactly that:
//   * one workgroup per CU (grid = CU count), 8 waves, tile = all 256 rows of x  x  128 weight rows, k step 64;
//     wave (mi, ni) owns x rows [128 mi, +128) x weight rows [32 ni, +32) => 4 accumulators of 32x32 per wave
//     (v_mfma_f32_32x32x16: the weight fragment is the A operand and is reused across the 4 x blocks; half the LDS
//     bytes per MFMA cycle of the 16x16x32 form the M <= 128 kernel (gemm_wstream.hip) uses);
//   * both operands are staged global -> VGPR -> LDS (16-B loads, ds_write_b128) into TWO 48-KB stages (x tile
//     256 x 128 B, w tile 128 x 128 B): tile t + 2 is requested into registers while tile t is computed and tile
//     t + 1 (requested one step earlier) is written to the other stage; one __syncthreads per step.  (The LDS-DMA
//     three-stage ring this kernel started with is what csrc/gemm_g3.hip does with dedicated loader waves; issued by
//     the MFMA waves themselves it lost to register staging, profiles/r02b_m256_gemm_ablation.txt.)
//   * LDS image: 128-B rows, 16-B chunk c of row r at slot c ^ ((r >> 1) & 7), conflict-free for the ds_read_b128
//     lane groups of a 32-row fragment read (MI355X_MICROARCH.md, LDS table);
//   * work split for balance, not for tiles: the first `full` tiles are computed whole (bf16 out staged through
//     LDS for full-line stores); the remaining tiles are cut into `tail_split` k-slices spread over ALL
//     workgroups, written as fp32 slabs and added in slice order by a small second kernel (deterministic).
//     gate_up (272 tiles): 256 whole + 16 tail tiles x 16 slices = 85 steps per workgroup, every CU equal.
//     qkv / o / down (56 / 40 / 40 tiles): full = 0, k-slices so that tiles x slices ~ CU count.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace msgl {

typedef __attribute__((ext_vector_type(16))) float g2_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 g2_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 g2_f16x8;
typedef uint32_t G4 __attribute__((ext_vector_type(4)));  // 16 bytes in registers

template <typename T>
__device__ __forceinline__ g2_f32x16 g2_mfma(const U4& a, const U4& b, g2_f32x16 c) {
  if constexpr (std::is_same_v<T, BF16>)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g2_bf16x8, a), __builtin_bit_cast(g2_bf16x8, b),
                                                   c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(g2_f16x8, a), __builtin_bit_cast(g2_f16x8, b), c,
                                                  0, 0, 0);
}

constexpr int kG2Threads = 512;
constexpr int kG2TileN = 128;                  // weight rows per tile
constexpr int kG2Rows = 256;                   // x rows per tile (M <= 256, rows >= M clamped on load, masked on store)
constexpr int kG2StepK = 64;
constexpr int kG2XBytes = kG2Rows * 128;       // 32 KB
constexpr int kG2WBytes = kG2TileN * 128;      // 16 KB
constexpr int kG2Stage = kG2XBytes + kG2WBytes;        // 48 KB
constexpr int kG2LdsBytes = 2 * kG2Stage;               // double buffer (96 KB); also holds the staged output tile
constexpr int kG2OutPitch = 264;               // bytes per row of the bf16 output tile staged in LDS (256 + 8)

template <typename T, int ABL = 0>
__global__ __launch_bounds__(kG2Threads) void m256_gemm_kernel(
    uint16_t* __restrict__ out, float* __restrict__ part, const uint16_t* __restrict__ x,
    const uint16_t* __restrict__ w, int M, int nsteps, int64_t ldx, int64_t ldw, int64_t ldo, int tiles, int full,
    int tail_split, int64_t ld_part) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kG2LdsBytes];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = sgpr(tid >> 6);
  const int mi = wv >> 2, ni = wv & 3;
  const int j = lane & 31, h = lane >> 5;

  // ---- staging map.  One wave load covers 8 rows x 128 B (whole lines): lane -> (row 8 i + lane / 8, chunk lane % 8);
  // the chunk lands in LDS at slot chunk ^ ((row >> 1) & 7).  x tile: 32 wave loads (4 per wave), w tile: 16 (2 per wave).
  const int drow = lane >> 3, dchunk = lane & 7;
  uint32_t xoff[4], woff[2];   // global byte offsets (row part; + k bytes of the step)
  int xl[4], wl[2];            // LDS byte offsets within a stage
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (wv * 4 + p) * 8 + drow;
    xoff[p] = (uint32_t)(min(row, M - 1) * (int)ldx * 2 + dchunk * 16);
    xl[p] = row * 128 + ((dchunk ^ ((row >> 1) & 7)) * 16);
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = (wv * 2 + p) * 8 + drow;
    woff[p] = (uint32_t)(row * (int)ldw * 2 + dchunk * 16);
    wl[p] = kG2XBytes + row * 128 + ((dchunk ^ ((row >> 1) & 7)) * 16);
  }
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(x);

  // ---- fragment read offsets within a stage (bytes): slot of chunk 2 s + h for this lane's row
  const int swz = (lane >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) fo[s] = ((2 * s + h) ^ swz) * 16;
  const int a_row = kG2XBytes + (ni * 32 + j) * 128;   // weight fragment row (A operand)
  const int b_row = (mi * 128 + j) * 128;              // x fragment row of block 0 (B operand); block b: + b * 32 * 128

  struct Regs {
    G4 x0, x1, x2, x3, w0, w1;
  };
  auto load = [&](Regs& r, const unsigned char* wt, int step) __attribute__((always_inline)) {
    const int64_t kb = (int64_t)step * (kG2StepK * 2);
    if (!(ABL & 1)) {
      r.x0 = *reinterpret_cast<const G4*>(xb + kb + xoff[0]);
      r.x1 = *reinterpret_cast<const G4*>(xb + kb + xoff[1]);
      r.x2 = *reinterpret_cast<const G4*>(xb + kb + xoff[2]);
      r.x3 = *reinterpret_cast<const G4*>(xb + kb + xoff[3]);
    }
    if (!(ABL & 4)) {
      if (ABL & 8) {  // probe: non-temporal weight loads
        r.w0 = __builtin_nontemporal_load(reinterpret_cast<const G4*>(wt + kb + woff[0]));
        r.w1 = __builtin_nontemporal_load(reinterpret_cast<const G4*>(wt + kb + woff[1]));
      } else {
        r.w0 = *reinterpret_cast<const G4*>(wt + kb + woff[0]);
        r.w1 = *reinterpret_cast<const G4*>(wt + kb + woff[1]);
      }
    }
  };
  auto store = [&](const Regs& r, int buf) __attribute__((always_inline)) {
    unsigned char* sb = smem + buf * kG2Stage;
    if (!(ABL & 1)) {
      *reinterpret_cast<G4*>(sb + xl[0]) = r.x0;
      *reinterpret_cast<G4*>(sb + xl[1]) = r.x1;
      *reinterpret_cast<G4*>(sb + xl[2]) = r.x2;
      *reinterpret_cast<G4*>(sb + xl[3]) = r.x3;
    }
    if (!(ABL & 4)) {
      *reinterpret_cast<G4*>(sb + wl[0]) = r.w0;
      *reinterpret_cast<G4*>(sb + wl[1]) = r.w1;
    }
  };

  // one segment: tile `tile`, steps [s0, s1); partial => fp32 slab `slice`, else bf16 out
  auto segment = [&](int tile, int s0, int s1, bool partial, int slice) __attribute__((always_inline)) {
    const unsigned char* wt = reinterpret_cast<const unsigned char*>(w) + (int64_t)tile * kG2TileN * ldw * 2;
    g2_f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    U4 a[4], bc[4], bn[4];
    U4 pa = U4{0, 0, 0, 0}, pb[4] = {pa, pa, pa, pa};  // deferred sub-step: zeros add nothing the first time

    // One step.  On entry: LDS stage `buf` holds tile `step` (visible to all), `have` holds tile step + 1 (requested one
    // step ago).  Request tile step + 2 into `next`; read the first fragments of this tile; run the MFMAs of the LAST
    // sub-step of the previous tile (operands kept in registers across the barrier) behind those reads; then sub-steps
    // 0..2 with the fragments of sub-step s + 1 requested before the MFMAs of sub-step s; publish tile step + 1 to the
    // other stage (nobody reads it any more: every wave passed the barrier that ended step - 1); barrier.
    // sched_barrier pins the order: left alone the compiler sinks each read to just before its use (MFMA pipe starved).
    auto body = [&](int buf, int step, const Regs& have, Regs& next) __attribute__((always_inline)) {
      load(next, wt, min(step + 2, s1 - 1));
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & 2)) {
        const unsigned char* sb = smem + buf * kG2Stage;
#pragma unroll
        for (int b = 0; b < 4; ++b) bc[b] = *reinterpret_cast<const U4*>(sb + b_row + b * (32 * 128) + fo[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const U4*>(sb + a_row + fo[s]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = g2_mfma<T>(pa, pb[b], acc[b]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
          for (int b = 0; b < 4; ++b) bn[b] = *reinterpret_cast<const U4*>(sb + b_row + b * (32 * 128) + fo[s + 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[b] = g2_mfma<T>(a[s], bc[b], acc[b]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int b = 0; b < 4; ++b) bc[b] = bn[b];
        }
        pa = a[3];
#pragma unroll
        for (int b = 0; b < 4; ++b) pb[b] = bc[b];
      }
      store(have, buf ^ 1);
      __syncthreads();
    };

    {
      Regs ra = {}, rb = {};
      load(ra, wt, s0);
      load(rb, wt, min(s0 + 1, s1 - 1));
      store(ra, 0);
      __syncthreads();
      int t = s0;
      for (; t + 2 <= s1; t += 2) {
        body(0, t, rb, ra);
        body(1, t + 1, ra, rb);
      }
      if (t < s1) body(0, t, rb, ra);
    }
    if (!(ABL & 2)) {
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = g2_mfma<T>(pa, pb[b], acc[b]);
    }

    // ---- epilogue.  Lane holds D[n = 32 ni + 8 g + 4 h + e][m = 128 mi + 32 b + j], g = reg >> 2, e = reg & 3.
    // (every wave is past the barrier that ended the last step: both stages are free)
    if (partial) {
      const int64_t col = (int64_t)(tile - full) * kG2TileN + ni * 32 + 4 * h;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int m = mi * 128 + b * 32 + j;
        if (m < M) {
          float* dst = part + ((int64_t)slice * M + m) * ld_part + col;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 v;
            v.x = acc[b][4 * g + 0]; v.y = acc[b][4 * g + 1]; v.z = acc[b][4 * g + 2]; v.w = acc[b][4 * g + 3];
            *reinterpret_cast<float4*>(dst + 8 * g) = v;
          }
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        unsigned char* row = smem + (mi * 128 + b * 32 + j) * kG2OutPitch + (ni * 32 + 4 * h) * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 pk;
          pk.x = Elem<T>::pack(acc[b][4 * g + 0], acc[b][4 * g + 1]);
          pk.y = Elem<T>::pack(acc[b][4 * g + 2], acc[b][4 * g + 3]);
          *reinterpret_cast<uint2*>(row + 16 * g) = pk;
        }
      }
      __syncthreads();
      // 256 rows x 256 B: a wave instruction stores 2 whole rows (32 lanes x 8 B each)
      const int c8 = tid & 31;
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int m = it * 16 + (tid >> 5);
        if (m < M) {
          const uint2 v = *reinterpret_cast<const uint2*>(smem + m * kG2OutPitch + c8 * 8);
          *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + (int64_t)tile * kG2TileN + c8 * 4) = v;
        }
      }
      __syncthreads();  // the output tile is read out before the next segment stages into it
    }
  };

  const int G = gridDim.x, g = blockIdx.x;
  for (int t = g; t < full; t += G) segment(t, 0, nsteps, false, 0);
  const int units = (tiles - full) * tail_split;
  for (int u = g; u < units; u += G) {
    const int tile = full + u / tail_split, slice = u % tail_split;
    const int s0 = (int)((int64_t)slice * nsteps / tail_split), s1 = (int)((int64_t)(slice + 1) * nsteps / tail_split);
    if (tail_split == 1)
      segment(tile, s0, s1, false, 0);
    else
      segment(tile, s0, s1, true, slice);
  }
}

// out[m][c0 + c] = round(sum_s part[s][m][c]), slabs added in slice order; 8 columns per thread
template <typename T>
__global__ __launch_bounds__(256) void m256_reduce_kernel(uint16_t* __restrict__ out, const float* __restrict__ part,
                                                          int M, int width, int S, int64_t ldo, int64_t c0) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = width >> 3;
  if (gid >= (int64_t)M * per_row) return;
  const int m = (int)(gid / per_row), c = (int)(gid - (int64_t)m * per_row);
  const float* p = part + (int64_t)m * width + c * 8;
  float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  for (int s = 1; s < S; ++s) {
    const float* q = p + (int64_t)s * M * width;
    const float4 a2 = *reinterpret_cast<const float4*>(q), b2 = *reinterpret_cast<const float4*>(q + 4);
    a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
    b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
  }
  U4 u;
  u.x = Elem<T>::pack(a.x, a.y); u.y = Elem<T>::pack(a.z, a.w);
  u.z = Elem<T>::pack(b.x, b.y); u.w = Elem<T>::pack(b.z, b.w);
  stg16(out + (int64_t)m * ldo + c0 + c * 8, u);
}

template <typename T>
static int launch_m256(uint16_t* out, float* part, const uint16_t* x, const uint16_t* w, int M, int N, int K,
                       int64_t ldx, int64_t ldw, int64_t ldo, int grid, int full, int tail_split, hipStream_t s,
                       bool skip_reduce = false) {
  const int tiles = N / kG2TileN, nsteps = K / kG2StepK;
  const int64_t width = (int64_t)(tiles - full) * kG2TileN;
  static const int abl = getenv("MSGL_M256_ABLATE") ? atoi(getenv("MSGL_M256_ABLATE")) : 0;  // diagnosis only
#define MSGL_G2(A)                                                                                              \
  m256_gemm_kernel<T, A><<<dim3((unsigned)grid), dim3(kG2Threads), 0, s>>>(out, part, x, w, M, nsteps, ldx, ldw, \
                                                                          ldo, tiles, full, tail_split, width)
  switch (abl) {
    case 2: MSGL_G2(2); break;
    case 3: MSGL_G2(3); break;
    case 5: MSGL_G2(5); break;
    case 6: MSGL_G2(6); break;
    case 8: MSGL_G2(8); break;
    case 11: MSGL_G2(11); break;
    default: MSGL_G2(0); break;
  }
#undef MSGL_G2
  if (tail_split > 1 && width > 0 && !skip_reduce) {
    const int64_t threads = (int64_t)M * (width / 8);
    m256_reduce_kernel<T><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>(
        out, part, M, (int)width, tail_split, ldo, (int64_t)full * kG2TileN);
  }
  return MSGL_OK;
}

}  // namespace msgl

using namespace msgl;

// Plan: `full` tiles whole + the rest in `tail_split` k-slices, for a grid of `grid` workgroups.
// Returns the fp32 workspace the plan needs (0 if no slabs).
extern "C" int64_t msgl_m256_gemm_workspace_bytes(int M, int N, int full, int tail_split) {
  if (M < 1 || N < kG2TileN || N % kG2TileN || full < 0 || full > N / kG2TileN || tail_split < 1) return MSGL_EINVAL;
  if (tail_split == 1) return 0;
  return (int64_t)tail_split * M * (int64_t)(N / kG2TileN - full) * kG2TileN * (int64_t)sizeof(float);
}

static int m256_entry(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                      int64_t ldo, int dtype, int grid, int full, int tail_split, void* workspace,
                      int64_t workspace_bytes, void* stream, bool skip_reduce) {
  MSGL_REQUIRE(out && x && w, "m256_gemm_nt: null pointer");
  MSGL_REQUIRE(M >= 1 && M <= kG2Rows, "m256_gemm_nt: M = %d outside [1, %d]", M, kG2Rows);
  MSGL_REQUIRE(N >= kG2TileN && N % kG2TileN == 0, "m256_gemm_nt: N = %d must be a multiple of %d", N, kG2TileN);
  MSGL_REQUIRE(K >= kG2StepK && K % kG2StepK == 0, "m256_gemm_nt: K = %d must be a multiple of %d", K, kG2StepK);
  const int tiles = N / kG2TileN, nsteps = K / kG2StepK;
  MSGL_REQUIRE(grid >= 1 && grid <= 4096, "m256_gemm_nt: grid %d", grid);
  MSGL_REQUIRE(full >= 0 && full <= tiles, "m256_gemm_nt: %d whole tiles of %d", full, tiles);
  MSGL_REQUIRE(tail_split >= 1 && tail_split <= nsteps && tail_split <= 64, "m256_gemm_nt: %d k-slices (steps %d)",
               tail_split, nsteps);
  MSGL_REQUIRE(ldx >= K && ldw >= K && ldo >= N && ldx % 8 == 0 && ldw % 8 == 0 && ldo % 8 == 0,
               "m256_gemm_nt: leading dimensions (%lld, %lld, %lld)", (long long)ldx, (long long)ldw, (long long)ldo);
  MSGL_REQUIRE((int64_t)kG2Rows * ldx * 2 < (1ll << 31) && (int64_t)kG2TileN * ldw * 2 < (1ll << 31),
               "m256_gemm_nt: operand tile exceeds 32-bit offsets");
  MSGL_REQUIRE(aligned16(x) && aligned16(w) && aligned16(out), "m256_gemm_nt: pointers must be 16-byte aligned");
  const int64_t need = msgl_m256_gemm_workspace_bytes(M, N, full, tail_split);
  if (need > 0)
    MSGL_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= need,
                 "m256_gemm_nt: plan needs %lld workspace bytes", (long long)need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (dtype == MSGL_BF16)
    rc = launch_m256<BF16>((uint16_t*)out, (float*)workspace, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx, ldw,
                           ldo, grid, full, tail_split, s, skip_reduce);
  else if (dtype == MSGL_FP16)
    rc = launch_m256<FP16>((uint16_t*)out, (float*)workspace, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx, ldw,
                           ldo, grid, full, tail_split, s, skip_reduce);
  else {
    set_error("m256_gemm_nt: unsupported dtype code %d", dtype);
    return MSGL_EINVAL;
  }
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("m256_gemm_nt");
  return MSGL_OK;
}

extern "C" int msgl_m256_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                                 int64_t ldo, int dtype, int grid, int full, int tail_split, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  return m256_entry(out, x, w, M, N, K, ldx, ldw, ldo, dtype, grid, full, tail_split, workspace, workspace_bytes, stream,
                    false);
}

// Slabs only: plan must be pure k-slicing (full == 0, tail_split > 1).  `workspace` then holds tail_split fp32 slabs
// [M][N] (slab stride M * N) for the consumer to add in slab order (msgl_fused_add_rmsnorm_slabs); `out` is not written.
extern "C" int msgl_m256_gemm_slabs_nt(const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                                       int dtype, int grid, int tail_split, void* workspace, int64_t workspace_bytes,
                                       void* stream) {
  MSGL_REQUIRE(tail_split > 1, "m256_gemm_slabs_nt: needs k-slices (tail_split %d)", tail_split);
  return m256_entry(workspace, x, w, M, N, K, ldx, ldw, N, dtype, grid, 0, tail_split, workspace, workspace_bytes, stream,
                    true);
}
