// Decode projection GEMM, "row-owner" generation (round 5), planned for 3 <= M <= 256 (takes M >= 1):
//     out[M, N] = x[M, K] . w[N, K]^T        (torch F.linear layout, bf16 / fp16, fp32 accumulate)
//
// What gemm_g3.hip left on the table ("What bounds the full-batch projections"):
//   * its tiles are 128 weight rows wide whatever N is, so gate_up (N = 34816 = 272 tiles on 256 CUs) runs a second round
//     for 6 % of its work (a 20-us tail + a reduce launch), qkv / o / down fill 224 / 240 / 240 of 256 CUs, and at batches
//     <= 128 the accumulator budget is half empty;
//   * its matrix waves (one per SIMD) run read-all-then-multiply with nobody to hide the LDS latency: 64 us of matrix time
//     for 34 us of MFMA work, which then shows through the 88-us memory time (95.8 us).
// This kernel keeps g3's memory engine -- four LOADER waves that only issue LDS-DMA (buffer_load ... lds, 1 KB per
// instruction, nt policy on the weight stream) into a ring of three stages, one counted vmcnt + one s_barrier per 64-k step --
// and changes the decomposition and the matrix side:
//   * the unit of ownership is 16 weight rows (one v_mfma_f32_16x16x32 A tile).  N / 16 units are cut into `tiles` BALANCED
//     contiguous ranges (widths differ by at most one unit) and every range into `slices` k-slices: plan (tiles, slices),
//     item i = (tile i / slices, slice i % slices) runs on workgroup i % grid.  (256, 1) on gate_up = every CU OWNS 8 or 9
//     units = 128 / 144 consecutive rows of w for all of K: no tail, no second round, no split-K slabs, and the SiLU.mul
//     epilogue stays legal; (64, 4) on qkv / o / down = 256 equal items.  Any N % 16 == 0.
//   * EIGHT matrix waves (two per SIMD: one hides the other's LDS latency), each owning 16 MTW token columns x all the
//     tile's units: MTW = 2 up to M = 256 (<= 9 units per tile), MTW = 1 up to M = 128 (<= 18 units per tile: half the x
//     bytes per weight byte).  Token tiles past M are neither loaded nor multiplied: the x traffic follows M in steps of 8 rows.
//   * the k loop body is compiled per tile width (switch on the width, uniform): straight-line MFMA blocks, no guards inside.
// By the per-CU law of DESIGN section 3 (t = w_cu / 27.6 B/ns + x_cu / 100 B/ns) gate_up at M = 256 costs 50 + 26 us against
// g3's 88 + tail and the library's 104 + 7 (activation launch); at M = 128: 50 + 13.
//
// LDS image as g3: 128-B rows (64 k), 16-B chunk c of row r at slot c ^ ((r >> 1) & 7), applied on the SOURCE address of the
// DMA; conflict-free for the ds_read_b128 lane groups of a 16-row x 64-B fragment (lane = row l & 15, chunk 4 kh + (l >> 4)).
// Accumulation order per output element: 32-k blocks in k order inside a slice, slices added in slice order by the consumer
// (slab-input norm / qk pass, or ro_reduce_kernel): independent of M, of the tile width and of the workgroup that ran the item.
//
// RO_EPI_SILU: `w` is a gate_up matrix in ops.interleave_gate_up order (64-row groups: 32 gate rows, then the 32 up rows of the
// same features).  A unit then is 8 gate + 8 up rows of 8 consecutive features, laid into the A tile as (gate f, gate f + 1,
// up f, up f + 1) per lane group, so that a lane holds gate and up of two features: the epilogue rounds both to the 16-bit
// type, applies silu_mul_f32 and writes out[M, N / 2] -- bit for bit the plain launch followed by msgl_silu_and_mul_interleaved
// (P/layers/activation.py:9-12 after P/layers/linear.py:32).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace msgl {

typedef __attribute__((ext_vector_type(4))) float ro_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 ro_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 ro_f16x8;
typedef __attribute__((address_space(3))) void* ro_lds_t;

template <typename T>
__device__ __forceinline__ ro_f32x4 ro_mfma(const U4& a, const U4& b, ro_f32x4 c) {
  if constexpr (std::is_same_v<T, BF16>)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ro_bf16x8, a), __builtin_bit_cast(ro_bf16x8, b), c, 0,
                                                   0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ro_f16x8, a), __builtin_bit_cast(ro_f16x8, b), c, 0, 0,
                                                  0);
}

constexpr int kRoMatrixWaves = 8;
constexpr int kRoLoaders = 4;
constexpr int kRoThreads = 64 * (kRoMatrixWaves + kRoLoaders);  // 768
constexpr int kRoStepK = 64;
constexpr int kRoUnitBytes = 16 * 128;  // one unit of one stage: 16 rows x 64 k

template <int MTW>
struct RoShape {
  static constexpr int kMaxRows = 16 * MTW * kRoMatrixWaves;  // 256 / 128 token rows
  static constexpr int kUMax = MTW == 2 ? 9 : 18;             // units per tile (accumulators: 4 MTW kUMax registers)
  static constexpr int kXBytes = kMaxRows * 128;              // 32 / 16 KB
  static constexpr int kWBytes = kUMax * kRoUnitBytes;        // 18 / 36 KB
  static constexpr int kStage = kXBytes + kWBytes;            // 50 / 52 KB
  static constexpr int kLdsBytes = 3 * kStage;                // 150 / 156 KB
  static constexpr int kXPieces = kMaxRows / 8 / kRoLoaders;  // 1-KB x pieces per loader and step: 8 / 4
  static constexpr int kWPieces = (2 * kUMax + kRoLoaders - 1) / kRoLoaders;  // at most: 5 / 9
};

constexpr int kRoLdsBytes = RoShape<1>::kLdsBytes > RoShape<2>::kLdsBytes ? RoShape<1>::kLdsBytes : RoShape<2>::kLdsBytes;
constexpr int kRoMaxXPieces = 8, kRoMaxWPieces = 9, kRoMaxMTW = 2, kRoMaxUnits = 18;  // fixed array bounds (see the kernel)
static_assert(RoShape<2>::kXPieces <= kRoMaxXPieces && RoShape<1>::kWPieces <= kRoMaxWPieces && RoShape<2>::kWPieces <= kRoMaxWPieces &&
              RoShape<1>::kUMax <= kRoMaxUnits && kRoLdsBytes <= 160 * 1024, "fixed bounds");

enum { RO_EPI_OUT = 0, RO_EPI_SILU = 1 };

struct RoParams {
  uint16_t* out;      // [M, N] (RO_EPI_SILU: [M, N / 2]); unused by k-sliced plans
  float* part;        // slices > 1: fp32 slabs [slices][M][N]
  const uint16_t* x;
  const uint16_t* w;
  int M, N, nsteps, units, tiles, slices;
  int64_t ldx, ldw, ldo;
  int abl;  // diagnosis (flags bits 8-15; 0 in production): 1 = no x loads, 2 = no LDS reads / MFMAs, 4 = no w loads,
            // 8 = static wave priorities (measured neutral), 16 = read-all-then-multiply matrix body at M > 128 (A/B partner)
};

// s_waitcnt vmcnt takes an immediate: the number of DMA instructions a loader has in flight per step depends on the tile
// width, so the counted wait is a (uniform) jump table
__device__ __forceinline__ void ro_wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// item -> (first unit, units, first step, end step, slice)
struct RoItem {
  int u0, ut, s0, s1, slice;
};
__device__ __forceinline__ RoItem ro_item(const RoParams& p, int i) {
  RoItem it;
  const int tile = i / p.slices;
  it.slice = sgpr(i - tile * p.slices);
  it.u0 = sgpr((int)((int64_t)tile * p.units / p.tiles));
  it.ut = sgpr((int)((int64_t)(tile + 1) * p.units / p.tiles)) - it.u0;
  it.s0 = sgpr((int)((int64_t)it.slice * p.nsteps / p.slices));
  it.s1 = sgpr((int)((int64_t)(it.slice + 1) * p.nsteps / p.slices));
  return it;
}

// stored row of (feature f, up?) in ops.interleave_gate_up order
__device__ __forceinline__ int ro_silu_row(int f, int up) { return 64 * (f >> 5) + 32 * up + (f & 31); }

template <typename T, int MTW, int EPI>
__global__ __launch_bounds__(kRoThreads) void ro_gemm_kernel(const RoParams p) {
  using S = RoShape<MTW>;
  // (hipcc / ROCm 7.2: a local array whose BOUND depends on a template parameter makes the host pass silently drop the kernel's
  // launch stub, see gemm_g3.hip -- every array below has a fixed bound; entries past the instantiation's own size are never
  // touched after unrolling)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kRoLdsBytes];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = sgpr(tid >> 6);
  const int G = gridDim.x, g = blockIdx.x;
  const int items = p.tiles * p.slices;
  const int M = p.M;

  if (p.abl & 8) {
    // (experiment) static priorities: the loaders first, then matrix waves 0-3 over their SIMD partners 4-7, so that the two
    // matrix waves of a SIMD fall out of lockstep (one multiplies while the other reads its fragments)
    if (wv >= kRoMatrixWaves) __builtin_amdgcn_s_setprio(3);
    else if (wv < 4) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(0);
  }
  if (wv >= kRoMatrixWaves) {
    // =============================== loader wave ===============================
    const int L = wv - kRoMatrixWaves;
    const int drow = lane >> 3, dchunk = lane & 7;
    // x piece P (rows [8 P, 8 P + 8) of the token tile) goes to loader P % 4; pieces past M are not loaded
    int xvo[kRoMaxXPieces];
    int nx = 0;
#pragma unroll
    for (int i = 0; i < S::kXPieces; ++i) {
      const int P = kRoLoaders * i + L, row = 8 * P + drow;
      xvo[i] = min(row, M - 1) * (int)p.ldx * 2 + ((dchunk ^ ((row >> 1) & 7)) * 16);
      nx += (8 * P < M) ? 1 : 0;
    }
    nx = sgpr((p.abl & 1) ? 0 : nx);
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), (short)0, -1, 0x00020000);
    // w piece P = LDS rows [8 P, 8 P + 8) of the tile = half of unit P / 2; loader P % 4.  P & 1 = L & 1 for all of a loader's
    // pieces, so the lane's row inside its unit (and with it the swizzle) is fixed
    const int r = 8 * (L & 1) + drow;
    const int wswz = (dchunk ^ ((4 * (L & 1) + (drow >> 1)) & 7)) * 16;

    int total = 0;
    for (int i = g; i < items; i += G) {
      const RoItem it = ro_item(p, i);
      total += it.s1 - it.s0;
    }
    total = sgpr(total);

    int wvo[kRoMaxWPieces];
    int nw = 0;
    const uint16_t* wbase = p.w;
    int next_item = g, cstep = 0, cend = 0, stage = 0, c_last = 0;
    for (int i = 0; i < total + 2; ++i) {
      if (i >= 2) {
        // step i - 2 has landed (step i - 1 may stay in flight); barrier: stage (i - 2) % 3 is visible to the matrix waves and
        // stage i % 3 (step i - 3) is free
        ro_wait_vm(i <= total ? c_last : 0);
        __builtin_amdgcn_s_barrier();
      }
      if (i < total) {
        if (cstep == cend) {  // next item: its rows' source offsets
          const RoItem it = ro_item(p, next_item);
          next_item += G;
          cstep = sgpr(it.s0);
          cend = sgpr(it.s1);
          nw = sgpr(2 * it.ut > L && !(p.abl & 4) ? (2 * it.ut - L + kRoLoaders - 1) / kRoLoaders : 0);
          int base_row;
          if (EPI == RO_EPI_SILU)
            base_row = ro_silu_row(8 * it.u0, 0);
          else
            base_row = 16 * it.u0;
          wbase = p.w + (int64_t)sgpr(base_row) * p.ldw;
#pragma unroll
          for (int q = 0; q < S::kWPieces; ++q) {
            const int u = (kRoLoaders * q + L) >> 1;  // unit of the tile
            int row;
            if (EPI == RO_EPI_SILU) {
              const int f = 8 * (it.u0 + u) + 2 * (r >> 2) + (r & 1);
              row = ro_silu_row(f, (r >> 1) & 1) - base_row;
            } else {
              row = 16 * u + r;
            }
            wvo[q] = row * (int)p.ldw * 2 + wswz;
          }
        }
        const __amdgpu_buffer_rsrc_t wr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wbase), (short)0, -1, 0x00020000);
        const int kb = cstep * (kRoStepK * 2);
        unsigned char* sb = smem + stage * S::kStage;
        // x and w pieces interleaved: the short-latency L2 hits and the HBM stream share the queue
#pragma unroll
        for (int q = 0; q < (S::kXPieces > S::kWPieces ? S::kXPieces : S::kWPieces); ++q) {
          if (q < S::kXPieces) {
            if (q < nx)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (ro_lds_t)(sb + (kRoLoaders * q + L) * 1024), 16, xvo[q], kb, 0, 0);
          }
          if (q < S::kWPieces) {
            if (q < nw)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (ro_lds_t)(sb + S::kXBytes + (kRoLoaders * q + L) * 1024), 16,
                                                       wvo[q], kb, 0, 2);
          }
        }
        c_last = nx + nw;
        ++cstep;
        stage = stage == 2 ? 0 : stage + 1;
      }
    }
    return;
  }

  // =============================== matrix wave ===============================
  const int r16 = lane & 15, q4 = lane >> 4;
  const int mtiles = (M + 15) >> 4;
  // token tiles of this wave: wv, wv + 8 (MTW = 2); nb of them exist
  const int nb = sgpr(mtiles > wv ? min(MTW, (mtiles - wv + kRoMatrixWaves - 1) / kRoMatrixWaves) : 0);
  const int lane_row = r16 * 128;
  int fo[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) fo[kh] = lane_row + (((4 * kh + q4) ^ ((r16 >> 1) & 7)) * 16);
  const int x_off = wv * kRoUnitBytes;  // token tile b: + b * 8 units
  int stage = 0;

  auto segment = [&](auto ut_tag, const RoItem it) __attribute__((always_inline)) {
    constexpr int UT = decltype(ut_tag)::value;
    ro_f32x4 acc[kRoMaxMTW][kRoMaxUnits];
#pragma unroll
    for (int b = 0; b < MTW; ++b)
#pragma unroll
      for (int u = 0; u < UT; ++u) acc[b][u] = ro_f32x4{0.f, 0.f, 0.f, 0.f};

    if (MTW == 2 && !(p.abl & (16 | 2))) {
      // Software-pipelined body (a wave with one token tile only multiplies a second, never stored one: M > 128 here, so every
      // wave has at least one): a half step's MFMAs run unit by unit, and as soon as a unit's two
      // MFMAs have issued its fragment registers are reloaded with the NEXT half step's fragment of that unit -- the LDS reads
      // of half step h + 1 travel under the MFMAs of half step h instead of in front of them (read-all-then-multiply leaves
      // the two matrix waves of a SIMD in lockstep: both read, then both multiply: 70 us of matrix time for 40 us of MFMAs).
      // The next STEP's fragments can only be requested after its barrier, which therefore sits between the two half steps;
      // lgkmcnt(0) in front of it: every read of this stage has returned before the loaders may refill it.
      U4 x0[2], x1[2], wa[kRoMaxUnits];
      auto rx = [&](U4(&xf)[2], const unsigned char* base, int kh) __attribute__((always_inline)) {
        xf[0] = *reinterpret_cast<const U4*>(base + x_off + fo[kh]);
        xf[1] = *reinterpret_cast<const U4*>(base + x_off + kRoMatrixWaves * kRoUnitBytes + fo[kh]);
      };
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned char* sb = smem + stage * S::kStage;
      rx(x0, sb, 0);
#pragma unroll
      for (int u = 0; u < UT; ++u) wa[u] = *reinterpret_cast<const U4*>(sb + S::kXBytes + u * kRoUnitBytes + fo[0]);
      // straight-line loop body (one basic block): every step but the last, then the last step without the look-ahead
      for (int step = it.s0; step + 1 < it.s1; ++step) {
        rx(x1, sb, 1);
#pragma unroll
        for (int u = 0; u < UT; ++u) {
          acc[0][u] = ro_mfma<T>(wa[u], x0[0], acc[0][u]);
          acc[1][u] = ro_mfma<T>(wa[u], x0[1], acc[1][u]);
          wa[u] = *reinterpret_cast<const U4*>(sb + S::kXBytes + u * kRoUnitBytes + fo[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        stage = stage == 2 ? 0 : stage + 1;
        sb = smem + stage * S::kStage;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        rx(x0, sb, 0);
#pragma unroll
        for (int u = 0; u < UT; ++u) {
          acc[0][u] = ro_mfma<T>(wa[u], x1[0], acc[0][u]);
          acc[1][u] = ro_mfma<T>(wa[u], x1[1], acc[1][u]);
          wa[u] = *reinterpret_cast<const U4*>(sb + S::kXBytes + u * kRoUnitBytes + fo[0]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      rx(x1, sb, 1);
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        acc[0][u] = ro_mfma<T>(wa[u], x0[0], acc[0][u]);
        acc[1][u] = ro_mfma<T>(wa[u], x0[1], acc[1][u]);
        wa[u] = *reinterpret_cast<const U4*>(sb + S::kXBytes + u * kRoUnitBytes + fo[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        acc[0][u] = ro_mfma<T>(wa[u], x1[0], acc[0][u]);
        acc[1][u] = ro_mfma<T>(wa[u], x1[1], acc[1][u]);
      }
      stage = stage == 2 ? 0 : stage + 1;
    } else
    for (int step = it.s0; step < it.s1; ++step) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned char* sb = smem + stage * S::kStage;
      const int kh_end = (p.abl & 2) ? 0 : 2;  // (diagnosis: no LDS reads, no MFMAs)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        if (kh >= kh_end) break;
        // a half step: its fragments requested up front, the MFMAs follow as they arrive (LDS returns in order); the other
        // matrix wave of this SIMD multiplies meanwhile
        U4 xa[kRoMaxMTW], wa[kRoMaxUnits];
#pragma unroll
        for (int b = 0; b < MTW; ++b)
          if (b < nb) xa[b] = *reinterpret_cast<const U4*>(sb + x_off + b * (kRoMatrixWaves * kRoUnitBytes) + fo[kh]);
#pragma unroll
        for (int u = 0; u < UT; ++u) wa[u] = *reinterpret_cast<const U4*>(sb + S::kXBytes + u * kRoUnitBytes + fo[kh]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < MTW; ++b)
          if (b < nb) {
#pragma unroll
            for (int u = 0; u < UT; ++u) acc[b][u] = ro_mfma<T>(wa[u], xa[b], acc[b][u]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      // every fragment of this stage is in registers once its last MFMA has issued, i.e. before this wave can arrive at the
      // next barrier; a wave without token tiles reads the weight fragments it never uses -- wait for them explicitly
      if (nb == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" ::: "memory");
      stage = stage == 2 ? 0 : stage + 1;
    }

    // ---- epilogue.  Lane holds D[n = 16 u + 4 q4 + e][m = 16 (wv + 8 b) + r16], e = register index.
#pragma unroll
    for (int b = 0; b < MTW; ++b) {
      const int m = 16 * (wv + kRoMatrixWaves * b) + r16;
      if (b < nb && m < M) {
        if (p.slices > 1) {
          float* dst = p.part + ((int64_t)it.slice * M + m) * p.N + 16 * it.u0 + 4 * q4;
#pragma unroll
          for (int u = 0; u < UT; ++u) {
            float4 v;
            v.x = acc[b][u][0]; v.y = acc[b][u][1]; v.z = acc[b][u][2]; v.w = acc[b][u][3];
            *reinterpret_cast<float4*>(dst + 16 * u) = v;
          }
        } else if (EPI == RO_EPI_SILU) {
          // registers 0, 1 = gate of features 8 (u0 + u) + 2 q4 + {0, 1}; registers 2, 3 = their up rows
          uint16_t* dst = p.out + (int64_t)m * p.ldo + 8 * it.u0 + 2 * q4;
#pragma unroll
          for (int u = 0; u < UT; ++u) {
            const uint32_t g01 = Elem<T>::pack(acc[b][u][0], acc[b][u][1]), u01 = Elem<T>::pack(acc[b][u][2], acc[b][u][3]);
            *reinterpret_cast<uint32_t*>(dst + 8 * u) = Elem<T>::pack(silu_mul_f32(Elem<T>::lo(g01), Elem<T>::lo(u01)),
                                                                       silu_mul_f32(Elem<T>::hi(g01), Elem<T>::hi(u01)));
          }
        } else {
          uint16_t* dst = p.out + (int64_t)m * p.ldo + 16 * it.u0 + 4 * q4;
#pragma unroll
          for (int u = 0; u < UT; ++u) {
            uint2 pk;
            pk.x = Elem<T>::pack(acc[b][u][0], acc[b][u][1]);
            pk.y = Elem<T>::pack(acc[b][u][2], acc[b][u][3]);
            *reinterpret_cast<uint2*>(dst + 16 * u) = pk;
          }
        }
      }
    }
  };

  for (int i = g; i < items; i += G) {
    const RoItem it = ro_item(p, i);
    const int ut = sgpr(it.ut);
#define RO_CASE(n) \
  case n:          \
    if constexpr (n <= S::kUMax) segment(std::integral_constant<int, n>{}, it); \
    break;
    switch (ut) {
      RO_CASE(1) RO_CASE(2) RO_CASE(3) RO_CASE(4) RO_CASE(5) RO_CASE(6) RO_CASE(7) RO_CASE(8) RO_CASE(9)
      RO_CASE(10) RO_CASE(11) RO_CASE(12) RO_CASE(13) RO_CASE(14) RO_CASE(15) RO_CASE(16) RO_CASE(17) RO_CASE(18)
      default: break;
    }
#undef RO_CASE
  }
}

// out[m][c] = round(sum_s part[s][m][c]), slabs added in slice order; 8 columns per thread (what a slab consumer does, as a
// launch of its own: `linear` on a k-sliced plan whose output does not go into a norm)
template <typename T>
__global__ __launch_bounds__(256) void ro_reduce_kernel(uint16_t* __restrict__ out, const float* __restrict__ part, int M,
                                                        int N, int S, int64_t ldo) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = N >> 3;
  if (gid >= (int64_t)M * per_row) return;
  const int m = (int)(gid / per_row), c = (int)(gid - (int64_t)m * per_row);
  const float* q = part + (int64_t)m * N + c * 8;
  float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
  for (int s = 1; s < S; ++s) {
    const float* q2 = q + (int64_t)s * M * N;
    const float4 a2 = *reinterpret_cast<const float4*>(q2), b2 = *reinterpret_cast<const float4*>(q2 + 4);
    a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
    b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
  }
  U4 u;
  u.x = Elem<T>::pack(a.x, a.y); u.y = Elem<T>::pack(a.z, a.w);
  u.z = Elem<T>::pack(b.x, b.y); u.w = Elem<T>::pack(b.z, b.w);
  stg16(out + (int64_t)m * ldo + c * 8, u);
}

template <typename T>
static void launch_ro(const RoParams& p, int grid, bool silu, bool slabs_only, hipStream_t s) {
  const dim3 g((unsigned)grid), b(kRoThreads);
  if (p.M > RoShape<1>::kMaxRows) {
    if (silu) ro_gemm_kernel<T, 2, RO_EPI_SILU><<<g, b, 0, s>>>(p);
    else ro_gemm_kernel<T, 2, RO_EPI_OUT><<<g, b, 0, s>>>(p);
  } else {
    if (silu) ro_gemm_kernel<T, 1, RO_EPI_SILU><<<g, b, 0, s>>>(p);
    else ro_gemm_kernel<T, 1, RO_EPI_OUT><<<g, b, 0, s>>>(p);
  }
  if (p.slices > 1 && !slabs_only) {
    const int64_t threads = (int64_t)p.M * (p.N / 8);
    ro_reduce_kernel<T><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>(p.out, p.part, p.M, p.N, p.slices, p.ldo);
  }
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_ro_gemm_max_units(int M) { return M > RoShape<1>::kMaxRows ? RoShape<2>::kUMax : RoShape<1>::kUMax; }

extern "C" int64_t msgl_ro_gemm_workspace_bytes(int M, int N, int slices) {
  return slices > 1 ? (int64_t)slices * M * N * 4 : 0;
}

extern "C" int msgl_ro_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                               int64_t ldo, int dtype, int tiles, int slices, int flags, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  const bool silu = flags & MSGL_RO_SILU, slabs_only = flags & MSGL_RO_SLABS_ONLY;
  MSGL_REQUIRE(x && w && (out || slabs_only), "ro_gemm_nt: null pointer");
  MSGL_REQUIRE(M >= 1 && M <= RoShape<2>::kMaxRows, "ro_gemm_nt: M = %d outside [1, %d]", M, RoShape<2>::kMaxRows);
  MSGL_REQUIRE(N >= 16 && N % 16 == 0, "ro_gemm_nt: N = %d must be a multiple of 16", N);
  MSGL_REQUIRE(K >= kRoStepK && K % kRoStepK == 0, "ro_gemm_nt: K = %d must be a multiple of %d", K, kRoStepK);
  const int units = N / 16, nsteps = K / kRoStepK, umax = msgl_ro_gemm_max_units(M);
  MSGL_REQUIRE(tiles >= 1 && tiles <= units, "ro_gemm_nt: %d tiles for %d 16-row units", tiles, units);
  MSGL_REQUIRE((units + tiles - 1) / tiles <= umax, "ro_gemm_nt: %d tiles leave %d units per tile, at most %d at M = %d", tiles,
               (units + tiles - 1) / tiles, umax, M);
  MSGL_REQUIRE(slices >= 1 && slices <= nsteps && slices <= 64, "ro_gemm_nt: %d k-slices (steps %d)", slices, nsteps);
  MSGL_REQUIRE((int64_t)tiles * slices < (1ll << 30), "ro_gemm_nt: %d x %d items", tiles, slices);
  MSGL_REQUIRE(!(silu && slices > 1), "ro_gemm_nt: the fused activation needs whole-K items (slices == 1)");
  MSGL_REQUIRE(!silu || N % 64 == 0, "ro_gemm_nt: an interleaved gate_up matrix has N %% 64 == 0 (N = %d)", N);
  MSGL_REQUIRE(!slabs_only || slices > 1, "ro_gemm_nt: slabs only needs k-slicing");
  const int64_t out_cols = silu ? N / 2 : N;
  MSGL_REQUIRE(ldx >= K && ldw >= K && (slabs_only || ldo >= out_cols) && ldx % 8 == 0 && ldw % 8 == 0 &&
                   (slabs_only || ldo % (silu ? 2 : 8) == 0),
               "ro_gemm_nt: leading dimensions (%lld, %lld, %lld)", (long long)ldx, (long long)ldw, (long long)ldo);
  // 32-bit source offsets: x rows from the matrix base, w rows from the tile's first row (a tile spans < 4 * 64 + 64 stored rows
  // more than its units in interleaved order)
  MSGL_REQUIRE((int64_t)RoShape<2>::kMaxRows * ldx * 2 < (1ll << 31) && (int64_t)(16 * umax + 160) * ldw * 2 < (1ll << 31),
               "ro_gemm_nt: operand tile exceeds 32-bit offsets");
  MSGL_REQUIRE(aligned16(x) && aligned16(w) && (slabs_only || (reinterpret_cast<uintptr_t>(out) & (silu ? 3u : 15u)) == 0),
               "ro_gemm_nt: pointers must be 16-byte aligned");
  const int64_t need = msgl_ro_gemm_workspace_bytes(M, N, slices);
  if (need > 0)
    MSGL_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= need, "ro_gemm_nt: plan needs %lld workspace bytes",
                 (long long)need);
  RoParams p;
  p.out = (uint16_t*)out; p.part = (float*)workspace; p.x = (const uint16_t*)x; p.w = (const uint16_t*)w;
  p.M = M; p.N = N; p.nsteps = nsteps; p.units = units; p.tiles = tiles; p.slices = slices;
  p.ldx = ldx; p.ldw = ldw; p.ldo = ldo;
  p.abl = (flags >> 8) & 0xff;
  const int items = tiles * slices, cus = device_cu_count();
  const int grid = items < cus ? items : cus;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == MSGL_BF16)
    launch_ro<BF16>(p, grid, silu, slabs_only, s);
  else if (dtype == MSGL_FP16)
    launch_ro<FP16>(p, grid, silu, slabs_only, s);
  else {
    set_error("ro_gemm_nt: unsupported dtype code %d", dtype);
    return MSGL_EINVAL;
  }
  MSGL_CHECK_LAUNCH("ro_gemm_nt");
  return MSGL_OK;
}
