// Full-batch decode projection GEMM, generation 3 (128 < M <= 256): loader waves + matrix waves.
//     out[M, N] = x[M, K] . w[N, K]^T        (torch F.linear layout, bf16 / fp16, fp32 accumulate)
//
// What round 2 measured on the register-staged kernel (gemm_m256.hip, profiles/r02b_m256_gemm_ablation.txt): weight
// stream alone 62 us, activation re-reads through L2 alone 29 us, both 89 us, both + MFMA 110 us -- the three are
// ADDITIVE because every wave issues its own loads: a vector-memory instruction that meets a full request queue
// blocks the in-order wave that issued it, and with it that wave's ds_reads and MFMAs.  This kernel separates the roles:
//   * workgroup = 8 waves, one per CU (grid = CU count).  Waves 4-7 (one per SIMD) are LOADERS: they only issue
//     LDS-DMA (buffer_load ... lds, 16 B per lane, 1 KB per instruction, no VGPR staging, no ds_write pass) into a
//     ring of THREE 48-KB stages (x tile 256 rows x 128 B, w tile 128 rows x 128 B), two stages in flight while the
//     third is computed, one counted vmcnt + one s_barrier per 64-k step.  A loader blocked on the memory pipe costs
//     nothing: the matrix waves never touch vector memory inside the k loop.
//   * waves 0-3 (one per SIMD) are MATRIX waves: wave (mi, ni) owns x rows [128 mi, +128) x weight rows [64 ni, +64)
//     = 4 x 2 accumulators of 32x32 (v_mfma_f32_32x32x16, weight fragment = A operand): 6 fragment reads per 8 MFMAs.
//   * LDS image: 128-B rows, 16-B chunk c of row r at slot c ^ ((r >> 1) & 7), applied on the SOURCE address of the
//     DMA (its LDS side is lane-linear); conflict-free for the ds_read_b128 lane groups of a 32-row fragment.
//   * the ring runs ACROSS segment boundaries: while the matrix waves store a finished tile the loaders already
//     stream the next segment's first two steps.
//   * work split as in gemm_m256.hip: `full` tiles whole, the rest cut into `tail_split` k-slices spread over all
//     workgroups (fp32 slabs, added in slice order by a second kernel or by the consumer of the projection).
//   * tried in round 4 and dropped (profiles/r04a_g3_store_policy_and_loader_roles.json, same-box, bit-identical outputs):
//     sc1 (write-through) or nt slab stores -- 32.1 -> 38.8 / 36.6 us (qkv), 25.2 -> 34.7 / 31.9 (o), and the pair with the
//     slab-consuming norm loses as much; two loaders issuing only x pieces and two only w pieces instead of a quarter of both
//     each -- within 0.5 us everywhere (the x and w terms add per CU whichever wave issues them).
//   * EPI_SILU: the weight is the gate_up matrix with rows interleaved in blocks of 32 (tile t of 128 rows =
//     gate[64t .. +32), up[64t .. +32), gate[64t+32 .. +32), up[64t+32 .. +32)): a matrix wave then holds gate and
//     up of the same output column in the same lane / register slot, and the epilogue writes
//     silu(round(gate)) * round(up) -- bit-identical to rounding the projection to 16 bits and running
//     msgl_silu_and_mul_interleaved on it (P/layers/activation.py:9-12 after P/layers/linear.py:32).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace msgl {

typedef __attribute__((ext_vector_type(16))) float g3_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 g3_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 g3_f16x8;
typedef __attribute__((address_space(3))) void* g3_lds_t;

template <typename T>
__device__ __forceinline__ g3_f32x16 g3_mfma(const U4& a, const U4& b, g3_f32x16 c) {
  if constexpr (std::is_same_v<T, BF16>)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g3_bf16x8, a), __builtin_bit_cast(g3_bf16x8, b),
                                                   c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(g3_f16x8, a), __builtin_bit_cast(g3_f16x8, b), c,
                                                  0, 0, 0);
}

constexpr int kG3Threads = 512;
constexpr int kG3TileN = 128;
constexpr int kG3Rows = 256;
constexpr int kG3StepK = 64;
constexpr int kG3XBytes = kG3Rows * 128;          // 32 KB
constexpr int kG3WBytes = kG3TileN * 128;         // 16 KB
constexpr int kG3Stage = kG3XBytes + kG3WBytes;   // 48 KB
constexpr int kG3Stages = 3;
constexpr int kG3LdsBytes = kG3Stages * kG3Stage;  // 144 KB

enum { G3_EPI_OUT = 0, G3_EPI_SILU = 1 };

// ABL (diagnosis): 1 = no x loads, 2 = no compute, 4 = no w loads, 8 = no epilogue stores.  WPOL = cache policy bits of the weight DMA
// (0 default, 2 = nt).
template <typename T, int EPI, int WPOL, int ABL>
__global__ __launch_bounds__(kG3Threads) void g3_gemm_kernel(
    uint16_t* __restrict__ out, float* __restrict__ part, const uint16_t* __restrict__ x,
    const uint16_t* __restrict__ w, int M, int nsteps, int64_t ldx, int64_t ldw, int64_t ldo, int tiles, int full,
    int tail_split, int64_t ld_part) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kG3LdsBytes];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = sgpr(tid >> 6);
  const int G = gridDim.x, g = blockIdx.x;
  const int units = (tiles - full) * tail_split;

  if (wv >= 4) {
    // =============================== loader wave ===============================
    const int L = wv - 4;
    const int drow = lane >> 3, dchunk = lane & 7;
    // piece p of the x tile = rows [8 p, 8 p + 8): this loader takes p = 4 i + L (i < 8); of the w tile p = 4 i + L (i < 4)
    int xvo[8], wvo[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (4 * i + L) * 8 + drow;
      xvo[i] = min(row, M - 1) * (int)ldx * 2 + ((dchunk ^ ((row >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (4 * i + L) * 8 + drow;
      wvo[i] = row * (int)ldw * 2 + ((dchunk ^ ((row >> 1) & 7)) * 16);
    }
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), (short)0, -1, 0x00020000);

    // segments of this workgroup: n_full whole tiles (g, g + G, ...) then n_unit k-slices (units g, g + G, ...)
    const int n_full = full > g ? (full - g + G - 1) / G : 0;
    const int n_unit = units > g ? (units - g + G - 1) / G : 0;
    int total = n_full * nsteps;
    for (int i = 0; i < n_unit; ++i) {
      const int slice = (g + i * G) % tail_split;
      total += (int)((int64_t)(slice + 1) * nsteps / tail_split) - (int)((int64_t)slice * nsteps / tail_split);
    }
    total = sgpr(total);
    constexpr int kPer = ((ABL & 1) ? 0 : 8) + ((ABL & 4) ? 0 : 4);  // DMA instructions per step and loader
    // Iteration i: [i >= 2: step i - 2 has landed (step i - 1 may stay in flight); barrier B_{i-2}: stage (i - 2) % 3 is
    // visible to the matrix waves and stage i % 3 (step i - 3) is free]  then  [i < total: issue step i into stage i % 3].
    int seg = 0, cstep = 0, cend = 0, ctile = 0, stage = 0;  // issue cursor (all wave-uniform)
    for (int i = 0; i < total + 2; ++i) {
      if (i >= 2) {
        if (i <= total) {
          if constexpr (kPer == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          else if constexpr (kPer == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else if constexpr (kPer == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      }
      if (i < total) {
        if (cstep == cend) {  // next segment
          if (seg < n_full) {
            ctile = g + seg * G; cstep = 0; cend = nsteps;
          } else {
            const int u = g + (seg - n_full) * G, slice = u % tail_split;
            ctile = full + u / tail_split;
            cstep = (int)((int64_t)slice * nsteps / tail_split);
            cend = (int)((int64_t)(slice + 1) * nsteps / tail_split);
          }
          ++seg;
          ctile = sgpr(ctile); cstep = sgpr(cstep); cend = sgpr(cend);
        }
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(w) + (int64_t)ctile * kG3TileN * ldw, (short)0, -1, 0x00020000);
        const int kb = cstep * (kG3StepK * 2);
        unsigned char* sb = smem + stage * kG3Stage;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // order x, x, w: the short-latency L2 hits and the HBM stream interleaved
          if (!(ABL & 1)) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (g3_lds_t)(sb + (4 * (2 * q) + L) * 1024), 16, xvo[2 * q], kb,
                                                     0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (g3_lds_t)(sb + (4 * (2 * q + 1) + L) * 1024), 16,
                                                     xvo[2 * q + 1], kb, 0, 0);
          }
          if (!(ABL & 4))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (g3_lds_t)(sb + kG3XBytes + (4 * q + L) * 1024), 16, wvo[q],
                                                     kb, 0, WPOL);
        }
        ++cstep;
        stage = stage == 2 ? 0 : stage + 1;
      }
    }
    return;
  }

  // =============================== matrix wave ===============================
  const int mi = wv >> 1, ni = wv & 1;
  const int j = lane & 31, h = lane >> 5;
  const int swz = (lane >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) fo[s] = ((2 * s + h) ^ swz) * 16;
  const int a_row = kG3XBytes + (ni * 64 + j) * 128;  // weight fragment row of block 0; block 1: + 32 * 128
  const int b_row = (mi * 128 + j) * 128;             // x fragment row of block 0; block b: + b * 32 * 128
  int stage = 0;

  auto segment = [&](int tile, int s0, int s1, bool partial, int slice) __attribute__((always_inline)) {
    g3_f32x16 acc[4][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][nb][e] = 0.f;

    for (int step = s0; step < s1; ++step) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!(ABL & 2)) {
        // all 24 fragments of the stage are requested up front (one wave per SIMD: nobody else hides the LDS latency;
        // 96 + 128 registers), the MFMAs then start as the first ones arrive (LDS returns in order)
        const unsigned char* sb = smem + stage * kG3Stage;
        U4 a0[4], a1[4], bb[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          a0[s] = *reinterpret_cast<const U4*>(sb + a_row + fo[s]);
#pragma unroll
          for (int b = 0; b < 4; ++b) bb[s][b] = *reinterpret_cast<const U4*>(sb + b_row + b * (32 * 128) + fo[s]);
          a1[s] = *reinterpret_cast<const U4*>(sb + a_row + 32 * 128 + fo[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[b][0] = g3_mfma<T>(a0[s], bb[s][b], acc[b][0]);
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[b][1] = g3_mfma<T>(a1[s], bb[s][b], acc[b][1]);
        }
      }
      // every fragment of this stage is in registers once the last MFMA group has issued (its operands are the last
      // reads and LDS returns in order), i.e. before this wave can arrive at the next barrier
      asm volatile("" ::: "memory");
      stage = stage == 2 ? 0 : stage + 1;
    }

    // ---- epilogue.  Lane holds D[n = 64 ni + 32 nb + 8 g4 + 4 h + e][m = 128 mi + 32 b + j], g4 = reg >> 2, e = reg & 3.
    if ((ABL & 8) && acc[0][0][0] != 1.2345e30f) {
      // diagnosis: no stores (the comparison keeps the accumulators alive)
    } else if (partial) {
      const int64_t col = (int64_t)(tile - full) * kG3TileN + ni * 64 + 4 * h;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int m = mi * 128 + b * 32 + j;
        if (m < M) {
          float* dst = part + ((int64_t)slice * M + m) * ld_part + col;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              float4 v;
              v.x = acc[b][nb][4 * g4 + 0]; v.y = acc[b][nb][4 * g4 + 1];
              v.z = acc[b][nb][4 * g4 + 2]; v.w = acc[b][nb][4 * g4 + 3];
              *reinterpret_cast<float4*>(dst + nb * 32 + 8 * g4) = v;
            }
        }
      }
    } else if (EPI == G3_EPI_SILU) {
      // out column of (tile, ni, g4, h, e) = 64 tile + 32 ni + 8 g4 + 4 h + e; gate = block 0, up = block 1
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int m = mi * 128 + b * 32 + j;
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t gb = Elem<T>::pack(acc[b][0][4 * g4 + e], 0.f), ub = Elem<T>::pack(acc[b][1][4 * g4 + e], 0.f);
            y[e] = silu_mul_f32(Elem<T>::lo(gb), Elem<T>::lo(ub));
          }
          lo[g4] = Elem<T>::pack(y[0], y[1]);
          hi[g4] = Elem<T>::pack(y[2], y[3]);
        }
        if (m < M) {
          uint16_t* dst = out + (int64_t)m * ldo + (int64_t)tile * 64 + ni * 32 + 4 * h;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            uint2 pk; pk.x = lo[g4]; pk.y = hi[g4];
            *reinterpret_cast<uint2*>(dst + 8 * g4) = pk;
          }
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int m = mi * 128 + b * 32 + j;
        if (m < M) {
          uint16_t* dst = out + (int64_t)m * ldo + (int64_t)tile * kG3TileN + ni * 64 + 4 * h;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              uint2 pk;
              pk.x = Elem<T>::pack(acc[b][nb][4 * g4 + 0], acc[b][nb][4 * g4 + 1]);
              pk.y = Elem<T>::pack(acc[b][nb][4 * g4 + 2], acc[b][nb][4 * g4 + 3]);
              *reinterpret_cast<uint2*>(dst + nb * 32 + 8 * g4) = pk;
            }
        }
      }
    }
  };

  for (int t = g; t < full; t += G) segment(t, 0, nsteps, false, 0);
  for (int u = g; u < units; u += G) {
    const int tile = full + u / tail_split, slice = u % tail_split;
    const int s0 = (int)((int64_t)slice * nsteps / tail_split), s1 = (int)((int64_t)(slice + 1) * nsteps / tail_split);
    if (tail_split == 1)
      segment(tile, s0, s1, false, 0);
    else
      segment(tile, s0, s1, true, slice);
  }
}

// out[m][c0 + c] = round(sum_s part[s][m][c]), slabs added in slice order; 8 columns per thread.
// SILU: the slab columns are an interleaved gate_up tail (blocks of 32: gate, up, gate, up per 128): thread owns 8
// output columns, out[m][c0 / 2 + ...] = silu(round(sum gate)) * round(sum up).
template <typename T, bool SILU>
__global__ __launch_bounds__(256) void g3_reduce_kernel(uint16_t* __restrict__ out, const float* __restrict__ part,
                                                        int M, int width, int S, int64_t ldo, int64_t c0) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = SILU ? (width >> 4) : (width >> 3);
  if (gid >= (int64_t)M * per_row) return;
  const int m = (int)(gid / per_row), c = (int)(gid - (int64_t)m * per_row);
  auto sum8 = [&](int col, float* r) {
    const float* p = part + (int64_t)m * width + col;
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    for (int s = 1; s < S; ++s) {
      const float* q = p + (int64_t)s * M * width;
      const float4 a2 = *reinterpret_cast<const float4*>(q), b2 = *reinterpret_cast<const float4*>(q + 4);
      a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
      b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
    }
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
  };
  U4 u;
  if constexpr (SILU) {
    // output piece c (8 columns at 8 c): tile = c / 8, q = (c % 8) / 4, r8 = c % 4 -> gate at 128 tile + 64 q + 8 r8, up + 32
    const int col = (c >> 3) * 128 + ((c >> 2) & 1) * 64 + (c & 3) * 8;
    float gsum[8], usum[8], y[8];
    sum8(col, gsum);
    sum8(col + 32, usum);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t gb = Elem<T>::pack(gsum[e], 0.f), ub = Elem<T>::pack(usum[e], 0.f);
      y[e] = silu_mul_f32(Elem<T>::lo(gb), Elem<T>::lo(ub));
    }
    u.x = Elem<T>::pack(y[0], y[1]); u.y = Elem<T>::pack(y[2], y[3]);
    u.z = Elem<T>::pack(y[4], y[5]); u.w = Elem<T>::pack(y[6], y[7]);
    stg16(out + (int64_t)m * ldo + (c0 >> 1) + c * 8, u);
  } else {
    float r[8];
    sum8(c * 8, r);
    u.x = Elem<T>::pack(r[0], r[1]); u.y = Elem<T>::pack(r[2], r[3]);
    u.z = Elem<T>::pack(r[4], r[5]); u.w = Elem<T>::pack(r[6], r[7]);
    stg16(out + (int64_t)m * ldo + c0 + c * 8, u);
  }
}

template <typename T>
static int launch_g3(uint16_t* out, float* part, const uint16_t* x, const uint16_t* w, int M, int N, int K,
                     int64_t ldx, int64_t ldw, int64_t ldo, int grid, int full, int tail_split, int flags,
                     hipStream_t s) {
  const int tiles = N / kG3TileN, nsteps = K / kG3StepK;
  const int64_t width = (int64_t)(tiles - full) * kG3TileN;
  const bool silu = flags & MSGL_G3_SILU, skip_reduce = flags & MSGL_G3_SLABS_ONLY;
  const int variant = (flags >> 8) & 0xff;  // diagnosis: bit 0 = nt weight stream off, bits 1-3 = ablation (x, compute, w)
#define MSGL_G3(E, P, A)                                                                                         \
  g3_gemm_kernel<T, E, P, A><<<dim3((unsigned)grid), dim3(kG3Threads), 0, s>>>(out, part, x, w, M, nsteps, ldx, \
                                                                                 ldw, ldo, tiles, full, tail_split, width)
  if (silu) {
    if (variant & 1) MSGL_G3(G3_EPI_SILU, 0, 0); else MSGL_G3(G3_EPI_SILU, 2, 0);
  } else {
    switch (variant) {
      case 0: MSGL_G3(G3_EPI_OUT, 2, 0); break;
      case 1: MSGL_G3(G3_EPI_OUT, 0, 0); break;
      case 2: MSGL_G3(G3_EPI_OUT, 2, 1); break;   // no x loads
      case 4: MSGL_G3(G3_EPI_OUT, 2, 2); break;   // no compute
      case 8: MSGL_G3(G3_EPI_OUT, 2, 4); break;   // no w loads
      case 6: MSGL_G3(G3_EPI_OUT, 2, 3); break;   // w stream only
      case 12: MSGL_G3(G3_EPI_OUT, 2, 6); break;  // x re-reads only
      case 10: MSGL_G3(G3_EPI_OUT, 2, 5); break;  // compute only
      case 16: MSGL_G3(G3_EPI_OUT, 2, 8); break;  // no epilogue stores
      default: set_error("g3_gemm_nt: unknown variant %d", variant); return MSGL_EINVAL;
    }
  }
#undef MSGL_G3
  if (tail_split > 1 && width > 0 && !skip_reduce) {
    if (silu) {
      const int64_t threads = (int64_t)M * (width / 16);
      g3_reduce_kernel<T, true><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>(
          out, part, M, (int)width, tail_split, ldo, (int64_t)full * kG3TileN);
    } else {
      const int64_t threads = (int64_t)M * (width / 8);
      g3_reduce_kernel<T, false><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>(
          out, part, M, (int)width, tail_split, ldo, (int64_t)full * kG3TileN);
    }
  }
  return MSGL_OK;
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_g3_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                               int64_t ldo, int dtype, int grid, int full, int tail_split, int flags, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  const bool silu = flags & MSGL_G3_SILU, slabs_only = flags & MSGL_G3_SLABS_ONLY;
  MSGL_REQUIRE(x && w && (out || slabs_only), "g3_gemm_nt: null pointer");
  MSGL_REQUIRE(M >= 1 && M <= kG3Rows, "g3_gemm_nt: M = %d outside [1, %d]", M, kG3Rows);
  MSGL_REQUIRE(N >= kG3TileN && N % kG3TileN == 0, "g3_gemm_nt: N = %d must be a multiple of %d", N, kG3TileN);
  MSGL_REQUIRE(K >= kG3StepK && K % kG3StepK == 0, "g3_gemm_nt: K = %d must be a multiple of %d", K, kG3StepK);
  const int tiles = N / kG3TileN, nsteps = K / kG3StepK;
  MSGL_REQUIRE(grid >= 1 && grid <= 4096, "g3_gemm_nt: grid %d", grid);
  MSGL_REQUIRE(full >= 0 && full <= tiles, "g3_gemm_nt: %d whole tiles of %d", full, tiles);
  MSGL_REQUIRE(tail_split >= 1 && tail_split <= nsteps && tail_split <= 64, "g3_gemm_nt: %d k-slices (steps %d)",
               tail_split, nsteps);
  MSGL_REQUIRE(!(silu && slabs_only), "g3_gemm_nt: the fused activation has no slabs-only form");
  MSGL_REQUIRE(!slabs_only || (full == 0 && tail_split > 1), "g3_gemm_nt: slabs only needs pure k-slicing");
  const int64_t out_cols = silu ? N / 2 : N;
  MSGL_REQUIRE(ldx >= K && ldw >= K && (slabs_only || ldo >= out_cols) && ldx % 8 == 0 && ldw % 8 == 0 && ldo % 8 == 0,
               "g3_gemm_nt: leading dimensions (%lld, %lld, %lld)", (long long)ldx, (long long)ldw, (long long)ldo);
  MSGL_REQUIRE((int64_t)kG3Rows * ldx * 2 < (1ll << 31) && (int64_t)kG3TileN * ldw * 2 < (1ll << 31),
               "g3_gemm_nt: operand tile exceeds 32-bit offsets");
  MSGL_REQUIRE(aligned16(x) && aligned16(w) && (slabs_only || aligned16(out)), "g3_gemm_nt: pointers must be 16-byte aligned");
  const int64_t need = msgl_m256_gemm_workspace_bytes(M, N, full, tail_split);
  if (need > 0)
    MSGL_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= need, "g3_gemm_nt: plan needs %lld workspace bytes",
                 (long long)need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (dtype == MSGL_BF16)
    rc = launch_g3<BF16>((uint16_t*)out, (float*)workspace, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx, ldw, ldo,
                         grid, full, tail_split, flags, s);
  else if (dtype == MSGL_FP16)
    rc = launch_g3<FP16>((uint16_t*)out, (float*)workspace, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx, ldw, ldo,
                         grid, full, tail_split, flags, s);
  else {
    set_error("g3_gemm_nt: unsupported dtype code %d", dtype);
    return MSGL_EINVAL;
  }
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("g3_gemm_nt");
  return MSGL_OK;
}
