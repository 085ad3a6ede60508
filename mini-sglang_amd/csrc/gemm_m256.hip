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
//   * both operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no VGPR staging) into a ring of
//     three 48-KB stages (x tile 256 x 128 B, w tile 128 x 128 B); one counted vmcnt + one raw s_barrier per step,
//     two stages in flight while the third is computed;
//   * LDS image: 128-B rows, 16-B chunk c of row r at slot c ^ ((r >> 1) & 7) -- applied on the SOURCE address of
//     the DMA (the LDS side of a DMA is lane-linear), conflict-free for the ds_read_b128 lane groups of a
//     32-row fragment read (MI355X_MICROARCH.md, LDS table);
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
typedef __attribute__((address_space(3))) void g2_lds_void;
typedef __attribute__((address_space(1))) const void g2_glb_void;

template <typename T, int ABL = 0>
__device__ __forceinline__ g2_f32x16 g2_mfma(const U4& a, const U4& b, g2_f32x16 c) {
  if constexpr ((ABL & 8) != 0) {  // ablation: consume the operands without the matrix pipe
    c[0] += __uint_as_float((a.x ^ b.x) & 0x3f800000u);
    c[1] += __uint_as_float((a.y ^ b.y ^ a.z ^ b.z ^ a.w ^ b.w) & 0x3f800000u);
    return c;
  }
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
constexpr int kG2XSlots = 2;                   // x tile of the step being read + the one landing
constexpr int kG2WSlots = 5;                   // w tile being read + four in flight (64 KB of the weight stream per CU)
constexpr int kG2WBase = kG2XSlots * kG2XBytes;
constexpr int kG2LdsBytes = kG2WBase + kG2WSlots * kG2WBytes;  // 144 KB
constexpr int kG2OutPitch = 264;               // bytes per row of the bf16 output tile staged in LDS (256 + 8)

template <typename T, int ABL = 0>
__global__ __launch_bounds__(kG2Threads) void m256_gemm_kernel(
    uint16_t* __restrict__ out, float* __restrict__ part, const uint16_t* __restrict__ x,
    const uint16_t* __restrict__ w, int M, int nsteps, int64_t ldx, int64_t ldw, int64_t ldo, int tiles, int full,
    int tail_split, int64_t ld_part, int wpat) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kG2LdsBytes];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = sgpr(tid >> 6);
  const int mi = wv >> 2, ni = wv & 3;
  const int j = lane & 31, h = lane >> 5;

  // ---- DMA source offsets (bytes).  One wave instruction fills 1 KB = 8 rows x 128 B; lane -> (row, slot):
  // row = 8 i + lane / 8, slot = lane % 8 holds global chunk slot ^ ((row >> 1) & 7).
  // Waves 0-3 request the x tile (L2-resident, 8 pieces each), waves 4-7 the weight tile (HBM, 4 pieces each): loads
  // return in order per wave, so a wave that mixed the two would see its L2 hits only after the HBM misses issued
  // before them, and the x tiles would sit in the in-flight window for a full HBM latency.  Split by wave, x needs
  // one step of lead and two slots, and the LDS that frees holds four weight tiles in flight.
  const int drow = lane >> 3, dslot = lane & 7;
  const bool x_role = wv < 4;
  uint32_t doff[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = x_role ? (wv * 8 + p) * 8 + drow : ((wv - 4) * 4 + (p & 3)) * 8 + drow;
    const int chunk = dslot ^ ((row >> 1) & 7);
    doff[p] = x_role ? (uint32_t)(min(row, M - 1) * (int)ldx * 2 + chunk * 16) : (uint32_t)(row * (int)ldw * 2 + chunk * 16);
    if ((ABL & 32) && !x_role && wpat == 1)
      doff[p] = (uint32_t)((((wv - 4) * 4 + (p & 3)) * 4 + (lane >> 4)) * (int)ldw * 2 + (lane & 15) * 16);
    if ((ABL & 32) && !x_role && wpat == 2) doff[p] = (uint32_t)(((wv - 4) * 4 + (p & 3)) * 1024 + lane * 16);
  }
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(x);

  // ---- fragment read offsets within a stage (bytes): slot of chunk 2 s + h for this lane's row
  const int swz = (lane >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) fo[s] = ((2 * s + h) ^ swz) * 16;
  const int a_row = (ni * 32 + j) * 128;               // weight fragment row (A operand), within a w slot
  const int b_row = (mi * 128 + j) * 128;              // x fragment row of block 0 (B operand); block b: + b * 32 * 128

  // LDS-DMA by inline asm: hipcc does not see these as LDS writes, so it neither drains them with a vmcnt(0) in
  // front of the next ds_read (it does for the builtin: every step would wait for the stage it just requested)
  // nor counts them; the pipeline is counted by hand below (6 DMAs per wave and step).  saddr form: uniform
  // 64-bit base in SGPRs + this lane's 32-bit byte offset; M0 = LDS byte address of the 1-KB piece.
  auto dma = [&](const unsigned char* base, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr)
                 : "memory", "m0");
  };
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  auto issue_x = [&](int step) {  // waves 0-3
    const int64_t kb = (int64_t)step * (kG2StepK * 2);
    const uint32_t sb = smem_base + (step & 1) * kG2XBytes + wv * 8192;
#pragma unroll
    for (int p = 0; p < 8; ++p)
      if (!(ABL & 1)) dma(xb + kb, doff[p], sb + p * 1024);
  };
  auto issue_w = [&](const unsigned char* wt, int step, int slot) {  // waves 4-7
    int64_t kb = (int64_t)step * (kG2StepK * 2);
    if (ABL & 32) {  // streaming-pattern probe (data unused): 1 = 4 rows x 256 B per piece, 2 = contiguous 16-KB tiles
      if (wpat == 1) kb = (int64_t)(step >> 1) * 256 + (int64_t)(step & 1) * 64 * ldw * 2;
      if (wpat == 2) kb = (int64_t)step * kG2WBytes;
    }
    const uint32_t sb = smem_base + kG2WBase + slot * kG2WBytes + (wv - 4) * 4096;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (!(ABL & 4)) dma(wt + kb, doff[p], sb + p * 1024);
  };

  // one segment: tile `tile`, steps [s0, s1); partial => fp32 slab `slice`, else bf16 out
  auto segment = [&](int tile, int s0, int s1, bool partial, int slice) {
    const unsigned char* wt = reinterpret_cast<const unsigned char*>(w) + (int64_t)tile * kG2TileN * ldw * 2;
    if ((ABL & 32) && wpat == 2) wt = reinterpret_cast<const unsigned char*>(w) + (int64_t)tile * nsteps * kG2WBytes;
    g2_f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

    // x slot of step t: t & 1 (absolute parity: consecutive segments keep alternating); w slot: ring position
    if (x_role) {
      issue_x(s0);
    } else {
#pragma unroll
      for (int d = 0; d < 4; ++d)
        if (s0 + d < s1) issue_w(wt, s0 + d, d);
    }
    int wslot = 0;
    // Software pipeline across the step boundary: the MFMAs of the LAST sub-step of step t - 1 (operands already
    // in registers) run after the barrier of step t, behind the first fragment reads of step t, so the LDS
    // latency at a step start is covered by MFMA work instead of idling the pipe; inside a step the x fragments
    // of sub-step s + 1 are requested before the MFMAs of sub-step s.  sched_barrier pins that order (the
    // compiler otherwise sinks every read to just before its use to save registers).
    U4 a[4], bc[4], bn[4];
    U4 pa = U4{0, 0, 0, 0}, pb[4] = {pa, pa, pa, pa};  // deferred sub-step: zeros => adds nothing the first time
    for (int t = s0; t < s1; ++t) {
      // step t's tiles landed (this wave's share: counted wait; everyone's: the barrier).  The barrier also says
      // every wave is done READING step t - 1's tiles, whose slots are refilled right after it.
      if (x_role) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        const int ahead = min(3, s1 - 1 - t);  // weight tiles issued after step t's: 4 pieces each
        if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ABL & 4) ? 0 : 12) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ABL & 4) ? 0 : 8) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ABL & 4) ? 0 : 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (x_role) {
        if (t + 1 < s1) issue_x(t + 1);
      } else {
        if (t + 4 < s1) issue_w(wt, t + 4, wslot == 0 ? 4 : wslot - 1);
      }
      const unsigned char* sx = smem + (t & 1) * kG2XBytes;
      const unsigned char* sw = smem + kG2WBase + wslot * kG2WBytes;
      wslot = wslot == 4 ? 0 : wslot + 1;
      if (ABL & 2) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) bc[b] = (ABL & 16) ? U4{(uint32_t)t, 1, 2, 3} : *reinterpret_cast<const U4*>(sx + b_row + b * (32 * 128) + fo[0]);
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] = (ABL & 16) ? U4{(uint32_t)t, 5, 6, 7} : *reinterpret_cast<const U4*>(sw + a_row + fo[s]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = g2_mfma<T, ABL>(pa, pb[b], acc[b]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int b = 0; b < 4; ++b) bn[b] = (ABL & 16) ? U4{(uint32_t)t, 9, 8, 7} : *reinterpret_cast<const U4*>(sx + b_row + b * (32 * 128) + fo[s + 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = g2_mfma<T, ABL>(a[s], bc[b], acc[b]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 4; ++b) bc[b] = bn[b];
      }
      pa = a[3];
#pragma unroll
      for (int b = 0; b < 4; ++b) pb[b] = bc[b];
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = g2_mfma<T, ABL>(pa, pb[b], acc[b]);

    // ---- epilogue.  Lane holds D[n = 32 ni + 8 g + 4 h + e][m = 128 mi + 32 b + j], g = reg >> 2, e = reg & 3.
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
      __builtin_amdgcn_s_barrier();  // all waves out of the k loop before the next segment refills stage 0
    } else {
      __builtin_amdgcn_s_barrier();  // every wave is done reading the last stage: LDS becomes the output tile
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
      __syncthreads();  // the output tile is read out before the next segment's DMAs land on it
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
                       int64_t ldx, int64_t ldw, int64_t ldo, int grid, int full, int tail_split, hipStream_t s) {
  const int tiles = N / kG2TileN, nsteps = K / kG2StepK;
  const int64_t width = (int64_t)(tiles - full) * kG2TileN;
  static const int abl = getenv("MSGL_M256_ABLATE") ? atoi(getenv("MSGL_M256_ABLATE")) : 0;
  static const int wpat = getenv("MSGL_M256_WPAT") ? atoi(getenv("MSGL_M256_WPAT")) : 0;
#define MSGL_G2(A)                                                                                              \
  m256_gemm_kernel<T, A><<<dim3((unsigned)grid), dim3(kG2Threads), 0, s>>>(out, part, x, w, M, nsteps, ldx, ldw, \
                                                                          ldo, tiles, full, tail_split, width, wpat)
  switch (abl) {
    case 1: MSGL_G2(1); break;
    case 2: MSGL_G2(2); break;
    case 3: MSGL_G2(3); break;
    case 4: MSGL_G2(4); break;
    case 5: MSGL_G2(5); break;
    case 6: MSGL_G2(6); break;
    case 13: MSGL_G2(13); break;
    case 21: MSGL_G2(21); break;
    case 35: MSGL_G2(35); break;
    default: MSGL_G2(0); break;
  }
#undef MSGL_G2
  if (tail_split > 1 && width > 0) {
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

extern "C" int msgl_m256_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                                 int64_t ldo, int dtype, int grid, int full, int tail_split, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
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
                           ldo, grid, full, tail_split, s);
  else if (dtype == MSGL_FP16)
    rc = launch_m256<FP16>((uint16_t*)out, (float*)workspace, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx, ldw,
                           ldo, grid, full, tail_split, s);
  else {
    set_error("m256_gemm_nt: unsupported dtype code %d", dtype);
    return MSGL_EINVAL;
  }
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("m256_gemm_nt");
  return MSGL_OK;
}
