// Weight-streaming projection GEMM for small decode batches (M <= 64):
//     out[M, N] = x[M, K] . w[N, K]^T        (torch F.linear layout, bf16 / fp16, fp32 accumulate)
//
// At M <= 64 a decode-step projection is a pure stream of the weight matrix (<= 64 flop per weight byte,
// ridge ~312), yet the BLAS library's kernels for these shapes reach only 1.4-4.4 TB/s on gfx950
// (profiles/r01d_gemm_sweep_small_batch.txt: down-projection 118 us for 178 MB at M = 1 .. 64).  This
// kernel is built around the stream instead of around a tile:
//   * a workgroup owns 16 consecutive weight rows; its SL waves split K between them (split-K inside the
//     workgroup: N/16 x SL waves fill the chip even for N = 5120), partial sums meet in LDS once, in a
//     fixed order (deterministic);
//   * weights go HBM -> VGPR directly (no LDS: nothing is shared between waves), 16 B per lane, in exactly
//     the v_mfma_f32_16x16x32 A-operand layout (lane = row l&15, k-group l>>4): a wave load covers
//     16 rows x 64 contiguous bytes and the two loads of a 64-k block complete the 128-B lines;
//   * activations (M x K, L2-resident, read by every workgroup) are the B operand in the same layout,
//     rows clamped to M-1 for the padding columns;
//   * a wave may take NT = 1, 2 or 4 row tiles at once: the activation fragment is then reused NT times
//     from registers (at M = 16, NT = 1 the activation re-reads through L2 -> L1 equal the weight stream);
//   * up to 4 x 64 k per step with all loads issued ahead of the MFMAs; the resident waves hide each
//     other's waits.
// The tuner (model.tune_gemms) times it against the library's best per (M, N, K) and records SL.
//
// SILU (msgl_skinny_gemm_silu_nt): `w` is a gate_up weight whose rows are interleaved in blocks of 32 (ops.interleave_gate_up:
// rows 64 j .. + 31 gate, + 32 .. + 63 up) and out[M, N/2] = silu(gate) * up: a workgroup takes NT/2 gate tiles and the NT/2
// up tiles 32 rows further (or, NT = 1, half of its waves the gate tile and half the up tile), so gate and up of one output
// column meet in one lane's accumulators (in LDS for NT = 1); both are rounded to the
// 16-bit type first, then silu_mul_f32 (common.h) -- bit for bit what msgl_skinny_gemm_nt + msgl_silu_and_mul_interleaved give,
// without the activation launch (~4.7 us of a decode step's ~200 per layer at small batch) and its round trip.
#include <type_traits>

#include "common.h"

namespace msgl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t W4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ f32x4 mfma16(const W4& a, const W4& b, const f32x4& c) {
  if constexpr (std::is_same_v<T, BF16>)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                   0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                  0, 0);
}

constexpr int kSkinnyMaxSlices = 16;
// more column / row tiles per wave = more operand registers: the workgroup (hence the register budget per
// wave) shrinks
__host__ __device__ constexpr int skinny_max_slices(int MT, int NT) { return MT * NT <= 2 ? 16 : MT * NT <= 8 ? 8 : 4; }

// MT = 16-token column tiles (M <= 16 MT), NT = 16-row weight tiles per wave: an activation fragment, read
// from L2 by every wave, is reused for NT weight tiles (at NT = 1 and M = 16 the activation traffic through
// L2 -> L1 equals the weight stream and halves the rate).  blockDim = 64 SL, grid = N / (16 NT),
// dynamic LDS = SL * MT * NT * 64 * 16 bytes.
template <typename T, int MT, int NT, bool SILU = false>
__global__ __launch_bounds__(64 * skinny_max_slices(MT, NT)) void skinny_gemm_kernel(
    uint16_t* __restrict__ out, const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, int M, int nblk,
    int64_t ldx, int64_t ldw, int64_t ldo) {
  extern __shared__ f32x4 red[];  // [SL][NT][MT][64]
  const int lane = threadIdx.x & 63;
  const int s = threadIdx.x >> 6;
  const int SL = blockDim.x >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const int blk = (int)blockIdx.x;
  // SILU with NT = 1: the first half of the workgroup's waves streams a gate tile, the second half its up tile (each half
  // splits K among its waves): one tile per wave, as the plain kernel at its fastest setting, and the pair still meets in LDS
  constexpr bool kSplitPair = SILU && NT == 1;
  const int SLk = kSplitPair ? SL >> 1 : SL;          // waves sharing one k range split
  const int up_half = kSplitPair ? (s >= SLk) : 0;
  const int sk = kSplitPair ? s - up_half * SLk : s;  // this wave's k slice
  // first weight row of the wave; SILU: NT <= 2 -> one 16-row gate tile and its up tile per workgroup, NT = 4 -> a whole 64-row group
  const int64_t n0 = !SILU ? (int64_t)blk * (16 * NT)
                     : NT <= 2 ? (int64_t)(blk >> 1) * 64 + (blk & 1) * 16 + up_half * 32 : (int64_t)blk * 64;
  // this wave's 64-k blocks
  const int b0 = (int)((int64_t)sk * nblk / SLk), b1 = (int)((int64_t)(sk + 1) * nblk / SLk);
  const uint16_t* wp = w + (n0 + r) * ldw + kg * 8;  // row tile i is 16 i rows further
  // row tile i starts 16 i rows further; SILU: the gate tiles first, the up tiles 32 rows behind their gate tile
  auto tile_off = [&](int i) { return (int64_t)(!SILU || NT == 1 ? 16 * i : (i % (NT / 2)) * 16 + (i / (NT / 2)) * 32) * ldw; };
  const uint16_t* xp[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) xp[t] = x + (int64_t)min(t * 16 + r, M - 1) * ldx + kg * 8;

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // U 64-k blocks per step: all their loads are issued back to back (2 U NT weight + 2 U MT activation
  // loads of 16 B per lane), the MFMAs follow as the data arrives.  No second register set: the other
  // resident waves' loads cover this wave's wait, and straight-line steps keep the compiler's vmcnt
  // bookkeeping exact (a conditional prefetch made it drain the queue every step).
  auto step = [&](int kb, auto u_tag) {
    constexpr int U = decltype(u_tag)::value;
    W4 a[U][NT][2], b[U][MT][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = (int64_t)(kb + u) * 64;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        a[u][i][0] = *reinterpret_cast<const W4*>(wp + tile_off(i) + k);
        a[u][i][1] = *reinterpret_cast<const W4*>(wp + tile_off(i) + k + 32);
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        b[u][t][0] = *reinterpret_cast<const W4*>(xp[t] + k);
        b[u][t][1] = *reinterpret_cast<const W4*>(xp[t] + k + 32);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep every load of the step ahead of its MFMAs (the scheduler
                                        // otherwise interleaves them to save registers: 5 loads in flight)
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          acc[i][t] = mfma16<T>(a[u][i][0], b[u][t][0], acc[i][t]);
          acc[i][t] = mfma16<T>(a[u][i][1], b[u][t][1], acc[i][t]);
        }
      }
    }
  };
  constexpr int kU = (NT + MT) <= 2 ? 4 : (NT + MT) <= 5 ? 2 : 1;  // 16-B loads in flight per lane: 2 U (NT + MT)
  int kb = b0;
  for (; kb + kU <= b1; kb += kU) step(kb, std::integral_constant<int, kU>{});
  for (; kb < b1; ++kb) step(kb, std::integral_constant<int, 1>{});

  // split-K partials meet in LDS; wave 0 adds them in slice order and writes 4 consecutive columns per lane
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int t = 0; t < MT; ++t) red[((s * NT + i) * MT + t) * 64 + lane] = acc[i][t];
  __syncthreads();
  auto slice_sum = [&](int i, int t, int j0 = 0, int j1 = -1) {
    if (j1 < 0) j1 = SL;
    f32x4 v = red[((j0 * NT + i) * MT + t) * 64 + lane];
    for (int j = j0 + 1; j < j1; ++j) {
      const f32x4 o = red[((j * NT + i) * MT + t) * 64 + lane];
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    return v;
  };
  if constexpr (SILU) {
    if (s == 0) {
      const int64_t c0 = NT <= 2 ? (int64_t)(blk >> 1) * 32 + (blk & 1) * 16 : (int64_t)blk * 32;
#pragma unroll
      for (int pr = 0; pr < (NT + 1) / 2; ++pr) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const f32x4 g = kSplitPair ? slice_sum(0, t, 0, SLk) : slice_sum(pr, t);
          const f32x4 u = kSplitPair ? slice_sum(0, t, SLk, SL) : slice_sum(pr + NT / 2, t);
          const int m = t * 16 + r;
          if (m < M) {
            // gate and up as the unfused path stores them (rounded to T), then the activation kernel's arithmetic
            const uint32_t g01 = Elem<T>::pack(g.x, g.y), g23 = Elem<T>::pack(g.z, g.w);
            const uint32_t u01 = Elem<T>::pack(u.x, u.y), u23 = Elem<T>::pack(u.z, u.w);
            uint2 pk;
            pk.x = Elem<T>::pack(silu_mul_f32(Elem<T>::lo(g01), Elem<T>::lo(u01)), silu_mul_f32(Elem<T>::hi(g01), Elem<T>::hi(u01)));
            pk.y = Elem<T>::pack(silu_mul_f32(Elem<T>::lo(g23), Elem<T>::lo(u23)), silu_mul_f32(Elem<T>::hi(g23), Elem<T>::hi(u23)));
            *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + c0 + pr * 16 + kg * 4) = pk;
          }
        }
      }
    }
    return;
  }
  if (s == 0) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const f32x4 v = slice_sum(i, t);
        const int m = t * 16 + r;
        if (m < M) {
          uint2 pk;
          pk.x = Elem<T>::pack(v.x, v.y);
          pk.y = Elem<T>::pack(v.z, v.w);
          *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + n0 + i * 16 + kg * 4) = pk;
        }
      }
    }
  }
}

template <typename T, int MT, int NT, bool SILU = false>
static int launch_skinny_t(uint16_t* out, const uint16_t* x, const uint16_t* w, int M, int N, int K, int64_t ldx,
                           int64_t ldw, int64_t ldo, int slices, hipStream_t s) {
  if (slices > skinny_max_slices(MT, NT)) {
    set_error("skinny_gemm_nt: at most %d k-slices for M = %d with %d row tiles per wave", skinny_max_slices(MT, NT),
              M, NT);
    return MSGL_EINVAL;
  }
  if (SILU && NT == 1 && (slices & 1)) {
    set_error("skinny_gemm_silu_nt: row_tiles 1 splits the waves between the gate and the up tile: %d k-slices is odd", slices);
    return MSGL_EINVAL;
  }
  const dim3 grid((unsigned)(N / (16 * (SILU && NT == 1 ? 2 : NT)))), block(64 * slices);
  const size_t lds = (size_t)slices * MT * NT * 64 * sizeof(f32x4);
  skinny_gemm_kernel<T, MT, NT, SILU><<<grid, block, lds, s>>>(out, x, w, M, K / 64, ldx, ldw, ldo);
  return MSGL_OK;
}

template <typename T, bool SILU = false>
static int launch_skinny(uint16_t* out, const uint16_t* x, const uint16_t* w, int M, int N, int K, int64_t ldx,
                         int64_t ldw, int64_t ldo, int slices, int row_tiles, hipStream_t s) {
  const int MT = M <= 16 ? 1 : M <= 32 ? 2 : 4;
#define MSGL_SKINNY(MT_, NT_) \
  if (MT == MT_ && row_tiles == NT_) return launch_skinny_t<T, MT_, NT_, SILU>(out, x, w, M, N, K, ldx, ldw, ldo, slices, s)
  MSGL_SKINNY(1, 1); MSGL_SKINNY(2, 1); MSGL_SKINNY(4, 1);
  MSGL_SKINNY(1, 2); MSGL_SKINNY(1, 4);
  MSGL_SKINNY(2, 2); MSGL_SKINNY(2, 4);
  MSGL_SKINNY(4, 2); MSGL_SKINNY(4, 4);
#undef MSGL_SKINNY
  set_error("skinny_gemm_nt: row_tiles %d unsupported (1, 2, 4)", row_tiles);
  return MSGL_EINVAL;
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_skinny_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx,
                                   int64_t ldw, int64_t ldo, int dtype, int slices, int row_tiles, void* stream) {
  MSGL_REQUIRE(out && x && w, "skinny_gemm_nt: null pointer");
  MSGL_REQUIRE(M >= 1 && M <= 64, "skinny_gemm_nt: M = %d outside [1, 64]", M);
  MSGL_REQUIRE(row_tiles == 1 || row_tiles == 2 || row_tiles == 4, "skinny_gemm_nt: row_tiles %d (1, 2, 4)", row_tiles);
  MSGL_REQUIRE(N >= 16 * row_tiles && N % (16 * row_tiles) == 0,
               "skinny_gemm_nt: N = %d must be a multiple of %d", N, 16 * row_tiles);
  MSGL_REQUIRE(K >= 64 && K % 64 == 0, "skinny_gemm_nt: K = %d must be a multiple of 64", K);
  MSGL_REQUIRE(slices >= 1 && slices <= kSkinnyMaxSlices && slices <= K / 64,
               "skinny_gemm_nt: %d k-slices outside [1, min(%d, K/64)]", slices, kSkinnyMaxSlices);
  MSGL_REQUIRE(ldx >= K && ldw >= K && ldo >= N && ldx % 8 == 0 && ldw % 8 == 0 && ldo % 4 == 0,
               "skinny_gemm_nt: leading dimensions (%lld, %lld, %lld)", (long long)ldx, (long long)ldw,
               (long long)ldo);
  MSGL_REQUIRE(aligned16(x) && aligned16(w) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0,
               "skinny_gemm_nt: x, w must be 16-byte and out 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (dtype == MSGL_BF16)
    rc = launch_skinny<BF16>((uint16_t*)out, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx, ldw, ldo, slices, row_tiles, s);
  else if (dtype == MSGL_FP16)
    rc = launch_skinny<FP16>((uint16_t*)out, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx, ldw, ldo, slices, row_tiles, s);
  else {
    set_error("skinny_gemm_nt: unsupported dtype code %d", dtype);
    return MSGL_EINVAL;
  }
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("skinny_gemm_nt");
  return MSGL_OK;
}

// gate_up projection + SiLU.mul in one launch: w [N, K] with rows in ops.interleave_gate_up order (blocks of 32: gate, up),
// out [M, N / 2].  row_tiles 2 or 4: a wave holds gate and up tiles; row_tiles 1 (slices even): half of the waves stream the
// gate tile, half the up tile, slices / 2 k-slices each.  Same bits as msgl_skinny_gemm_nt (with slices / 2 k-slices for
// row_tiles 1) followed by msgl_silu_and_mul_interleaved.
extern "C" int msgl_skinny_gemm_silu_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx,
                                        int64_t ldw, int64_t ldo, int dtype, int slices, int row_tiles, void* stream) {
  MSGL_REQUIRE(out && x && w, "skinny_gemm_silu_nt: null pointer");
  MSGL_REQUIRE(M >= 1 && M <= 64, "skinny_gemm_silu_nt: M = %d outside [1, 64]", M);
  MSGL_REQUIRE(row_tiles == 1 || row_tiles == 2 || row_tiles == 4, "skinny_gemm_silu_nt: row_tiles %d (1, 2, 4)", row_tiles);
  MSGL_REQUIRE(N >= 64 && N % 64 == 0, "skinny_gemm_silu_nt: N = %d must be a multiple of 64 (gate / up blocks of 32 rows)", N);
  MSGL_REQUIRE(K >= 64 && K % 64 == 0, "skinny_gemm_silu_nt: K = %d must be a multiple of 64", K);
  MSGL_REQUIRE(slices >= 1 && slices <= kSkinnyMaxSlices && (row_tiles == 1 ? slices / 2 : slices) <= K / 64,
               "skinny_gemm_silu_nt: %d k-slices outside [1, min(%d, K/64)]", slices, kSkinnyMaxSlices);
  MSGL_REQUIRE(ldx >= K && ldw >= K && ldo >= N / 2 && ldx % 8 == 0 && ldw % 8 == 0 && ldo % 4 == 0,
               "skinny_gemm_silu_nt: leading dimensions (%lld, %lld, %lld)", (long long)ldx, (long long)ldw, (long long)ldo);
  MSGL_REQUIRE(aligned16(x) && aligned16(w) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0,
               "skinny_gemm_silu_nt: x, w must be 16-byte and out 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (dtype == MSGL_BF16)
    rc = launch_skinny<BF16, true>((uint16_t*)out, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx, ldw, ldo, slices, row_tiles, s);
  else if (dtype == MSGL_FP16)
    rc = launch_skinny<FP16, true>((uint16_t*)out, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx, ldw, ldo, slices, row_tiles, s);
  else {
    set_error("skinny_gemm_silu_nt: unsupported dtype code %d", dtype);
    return MSGL_EINVAL;
  }
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("skinny_gemm_silu_nt");
  return MSGL_OK;
}
