// Row-streaming projection for the smallest decode batches (M <= 8):
//     out[M, N] = f(x)[M, K] . w[N, K]^T        (torch F.linear layout, bf16 / fp16, fp32 accumulate)
//
// At M <= 8 a projection is nothing but the weight matrix read once.  tools/cu_pipe_probe.hip / tools/persist_probe.hip
// measured what that costs on gfx950: a CU whose waves keep >= 64 KB of 16-byte loads in flight on a CONTIGUOUS private
// window pulls ~38 B/ns, 256 of them 7.0-7.25 TB/s, and a launch boundary between two such kernels costs 1-2 us.  The
// matrix-core kernel of gemm_skinny.hip reaches 4.5-5.3 TB/s on the same matrices (profiles/r04_b1_kernel_breakdown.txt):
// 16-row workgroups that live for one or two load batches, K split over the waves of a workgroup, N / 16 workgroups
// quantised against 256 CUs (down_proj: 320 tiles, 160 workgroups of two).  This kernel is the probe with a dot product:
//   * ONE workgroup (8 waves) per CU; CU c owns the consecutive rows [c N / G, (c + 1) N / G) -- balanced to one row
//     whatever N is -- i.e. one contiguous window of the weight matrix;
//   * the window is cut into 1-KB units (64 lanes x 16 B of one row); wave v takes units v, v + 8, ... so the eight
//     waves walk the window together, each with D (8 / 16) units in flight in a register ring: 64 / 128 KB per CU;
//   * x (all M rows, <= 139 KB) is staged ONCE into LDS behind the first D weight loads; a unit costs M ds_read_b128 and
//     4 M v_dot2 per lane (VALU: at M <= 4 under a third of what the weight stream leaves room for);
//   * a wave keeps per-lane partial sums of the row it is in and, when its units move to the next row, reduces them over
//     its lanes and parks the sum in LDS [row][wave][m]; after the stream the eight wave sums of a row are added in
//     wave order (deterministic) and rounded once.
// The staging pass is where the neighbouring row kernels of a decode layer fold in (mode):
//   1 / 2  x is a gate_up output [M, 2 K] (halves / interleaved in blocks of 32): stages silu(gate) * up -- the bits of
//          msgl_silu_and_mul[_interleaved] -- and the activation launch disappears;
//   3      x is the previous projection's output: stages fused_add_rmsnorm(x, residual) (every workgroup computes the
//          row statistics itself, in the order of rmsnorm_wide_row_kernel: same bits), workgroup 0 writes the new
//          residual to `res_out` (a different buffer: the other workgroups are still reading `res_in`).
// M is padded to 1 / 2 / 4 / 8 staged rows (zeros).  This first form (variant 0, K % 512 == 0) is the fastest at M = 1;
// from two rows on its M LDS reads and 4 M quarter-rate v_dot2 per load are the limit (gate_up 60 -> 70 -> 106 -> 178 us at
// M = 1 / 2 / 4 / 8) and the matrix-core form further down (variant 1: v_mfma_f32_4x4x4_16b, units of four rows x 128 k)
// takes over.  Measured on Qwen3-14B (profiles/r04_rowstream_bench.json, r04_small_batch_ab.json): gate_up 64 -> 57 us
// (6.2 TB/s), down_proj 44 -> 31, lm_head 251 -> 234 (6.66 TB/s) at M = 1, the same within 4 us up to M = 8; the captured
// decode step 6.65 -> 5.79 ms at B = 1, 6.85 -> 6.23 at 2, 7.17 -> 6.67 at 4, 7.69 -> 7.42 at 8.
#include <type_traits>

#include "common.h"

namespace msgl {

typedef uint32_t RW4 __attribute__((ext_vector_type(4)));

constexpr int kRsWaves = 8;
constexpr int kRsThreads = 64 * kRsWaves;
constexpr int kRsUnit = 512;              // elements of one unit: 64 lanes x 8
constexpr int kRsLdsBudget = 159 * 1024;  // dynamic LDS a workgroup may ask for (160 KB per CU, 512 B static)

enum { kRsPlain = 0, kRsSiluHalves = 1, kRsSiluIlv = 2, kRsAddNorm = 3 };

struct RowStreamParams {
  uint16_t* out;
  const uint16_t* x;
  const uint16_t* w;
  int M, N, K;
  int64_t ldx, ldw, ldo;
  // mode 3
  const uint16_t* res_in;
  uint16_t* res_out;
  const uint16_t* gamma;
  float eps;
  int64_t ldr_in, ldr_out;
};

__host__ __device__ constexpr int rs_padded_rows(int M) { return M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : 8; }

static inline int64_t rs_lds_bytes(int M, int N, int K, int G) {
  const int mm = rs_padded_rows(M);
  const int64_t rows = ((int64_t)N + G - 1) / G;
  return (int64_t)mm * K * 2 + rows * kRsWaves * mm * 4;
}

// ---- stage f(x) into LDS rows of `pitch` elements (rows >= M of the MM staged ones are zeros) ----
template <typename T, int MM, int MODE>
__device__ __forceinline__ void rs_stage(const RowStreamParams& p, uint16_t* xs, int pitch, float (*red)[16], int c, int tid, int lane,
                                         int wv) {
  const int K = p.K;
  const int pieces = K >> 3;
  if constexpr (MODE == kRsAddNorm) {
    // rmsnorm_wide_row_kernel (norm_rope_act.hip) runs one row on `pieces` threads, thread q = piece q: per-piece fma
    // chain, wave_sum over the 64 pieces of a wave, waves added in index order.  Here thread t holds pieces t and
    // t + 512 -- lane and 64-piece group are the same as there -- and the wave sums land in red[m][group].
    const int nw = (pieces + 63) >> 6;
    float v[MM][2][8];
    U4 gq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + kRsThreads * i;
      gq[i] = q < pieces ? ldg16(p.gamma + q * 8) : U4{0, 0, 0, 0};
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = tid + kRsThreads * i;
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[m][i][e] = 0.f;
        if (q < pieces && m < p.M) {
          float rr[8];
          unpack8<T>(ldg16(p.x + (int64_t)m * p.ldx + q * 8), v[m][i]);
          unpack8<T>(ldg16(p.res_in + (int64_t)m * p.ldr_in + q * 8), rr);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[m][i][e] += rr[e];
          if (c == 0) stg16(p.res_out + (int64_t)m * p.ldr_out + q * 8, pack8<T>(v[m][i]));
#pragma unroll
          for (int e = 0; e < 8; ++e) ss = fmaf(v[m][i][e], v[m][i][e], ss);
        }
        ss = wave_sum(ss);
        if (lane == 0) red[m][wv + kRsWaves * i] = ss;
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      float tot = 0.f;
      for (int i = 0; i < nw; ++i) tot += red[m][i];
      const float inv = rsqrtf(tot / (float)K + p.eps);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = tid + kRsThreads * i;
        if (q < pieces) {
          float g[8], y[8];
          unpack8<T>(gq[i], g);
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = __fmul_rn(__fmul_rn(v[m][i][e], inv), g[e]);
          *reinterpret_cast<U4*>(xs + (size_t)m * pitch + q * 8) = m < p.M ? pack8<T>(y) : U4{0, 0, 0, 0};
        }
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      const uint16_t* xr = p.x + (int64_t)m * p.ldx;
      for (int q = tid; q < pieces; q += kRsThreads) {
        U4 o = U4{0, 0, 0, 0};
        if (m < p.M) {
          if constexpr (MODE == kRsPlain) {
            o = ldg16(xr + q * 8);
          } else {
            const int k = q * 8;
            const int gcol = MODE == kRsSiluIlv ? (k >> 5) * 64 + (k & 31) : k;
            const int ucol = MODE == kRsSiluIlv ? gcol + 32 : K + k;
            float g[8], u[8], y[8];
            unpack8<T>(ldg16(xr + gcol), g);
            unpack8<T>(ldg16(xr + ucol), u);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = silu_mul_f32(g[e], u[e]);
            o = pack8<T>(y);
          }
        }
        *reinterpret_cast<U4*>(xs + (size_t)m * pitch + q * 8) = o;
      }
    }
  }
  __syncthreads();

}

template <typename T, int MM, int D, int MODE>
__global__ __launch_bounds__(kRsThreads, 2) void rowstream_gemm_kernel(const RowStreamParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
  __shared__ float red[8][16];  // mode 3: [staged row][64-piece group]
  const int K = p.K;
  uint16_t* xs = reinterpret_cast<uint16_t*>(rs_smem);                  // [MM][K]
  float* part = reinterpret_cast<float*>(rs_smem + (size_t)MM * K * 2);  // [rows][8 waves][MM]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = sgpr(tid >> 6);
  const int G = gridDim.x, c = blockIdx.x;
  const int r0 = (int)((int64_t)c * p.N / G), r1 = (int)((int64_t)(c + 1) * p.N / G);
  const int rows = r1 - r0;
  const int UR = K >> 9;        // units per row
  const int total = rows * UR;  // units of this workgroup
  const uint16_t* wrow0 = p.w + (int64_t)r0 * p.ldw;

  // unit u of the workgroup = (row u / UR, k-chunk u % UR); a wave steps by 8 units.  Past the end a wave keeps asking
  // for the matrix's first KB (cache hits): the loads of the ring stay unconditional, the wait counts exact.
  auto unit_ptr = [&](int u, int row, int kc) {
    const uint16_t* q = u < total ? wrow0 + (int64_t)row * p.ldw + kc * kRsUnit : p.w;
    return reinterpret_cast<const RW4*>(q + lane * 8);
  };
  auto advance = [&](int& u, int& row, int& kc) {
    u += kRsWaves;
    kc += kRsWaves;
    while (kc >= UR) {
      kc -= UR;
      ++row;
    }
  };

  RW4 ring[D];
  int lu = wv, lrow = wv / UR, lkc = wv - lrow * UR;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    ring[j] = __builtin_nontemporal_load(unit_ptr(lu, lrow, lkc));
    advance(lu, lrow, lkc);
  }

  for (int i = tid; i < rows * kRsWaves * MM; i += kRsThreads) part[i] = 0.f;

  rs_stage<T, MM, MODE>(p, xs, K, red, c, tid, lane, wv);

  // ---- the stream ----
  float acc0[MM], acc1[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) acc0[m] = acc1[m] = 0.f;
  auto flush = [&](int row) {
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      const float s = wave_sum(acc0[m] + acc1[m]);
      if (lane == 0) part[(row * kRsWaves + wv) * MM + m] = s;
      acc0[m] = acc1[m] = 0.f;
    }
  };
  int cu = wv, crow = wv / UR, ckc = wv - crow * UR;
  int cur_row = crow;
  while (cu < total) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if (cu < total) {
        if (crow != cur_row) {
          flush(cur_row);
          cur_row = crow;
        }
        const RW4 wq = ring[j];
        const uint16_t* xk = xs + ckc * kRsUnit + lane * 8;
#pragma unroll
        for (int m = 0; m < MM; ++m) {
          const RW4 xv = *reinterpret_cast<const RW4*>(xk + (size_t)m * K);
          acc0[m] = Elem<T>::dot2(wq.x, xv.x, acc0[m]);
          acc1[m] = Elem<T>::dot2(wq.y, xv.y, acc1[m]);
          acc0[m] = Elem<T>::dot2(wq.z, xv.z, acc0[m]);
          acc1[m] = Elem<T>::dot2(wq.w, xv.w, acc1[m]);
        }
      }
      ring[j] = __builtin_nontemporal_load(unit_ptr(lu, lrow, lkc));
      advance(lu, lrow, lkc);
      advance(cu, crow, ckc);
    }
  }
  if (wv < total) flush(cur_row);
  __syncthreads();

  // ---- the eight wave sums of every (row, m), in wave order; consecutive threads write consecutive columns ----
  for (int i = tid; i < rows * MM; i += kRsThreads) {
    const int m = i / rows, r = i - m * rows;
    if (m < p.M) {
      float s = 0.f;
#pragma unroll
      for (int v2 = 0; v2 < kRsWaves; ++v2) s += part[(r * kRsWaves + v2) * MM + m];
      p.out[(int64_t)m * p.ldo + r0 + r] = (uint16_t)Elem<T>::bits(s);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same stream with the products on the matrix cores (variant 1), for M = 2 .. 8 where the vector-unit form runs out
// of LDS bandwidth and v_dot2 issue slots (profiles/r04_rowstream_bench.json: gate_up 61 us at M = 1, 107 us at M = 4):
// v_mfma_f32_4x4x4_16b -- 16 independent blocks of D[4][4] += A[4][4] . B[4][4], lane 4 b + i holding row i of block
// b's A, lane 4 b + j column j of its B and of its D (tools/mfma444_probe.hip) -- with
//     A = four consecutive weight rows x four consecutive k,   B = the same four k of x rows m = j (and j + 4),
// block b taking the b-th 8-k piece of a 128-k chunk.  A unit is therefore 4 rows x 128 k (1 KB per wave load: 256
// contiguous bytes of each of the four rows), one ds_read_b128 of x per four staged rows and two MFMAs per 16-byte load,
// whatever M is.  CU c owns the consecutive 4-row groups [c N/4 / G, (c + 1) N/4 / G); a wave's accumulators belong to
// the group it is in; when its units move on, the 16 blocks (k pieces) are added across lanes and the 4 x 4 sums parked in
// LDS [group][wave]; after the stream the eight wave sums are added in wave order.  x rows in LDS are K + 32 elements
// apart: the four rows a ds_read_b128 touches then sit in different banks.  N % 4 == 0, K % 128 == 0.
// ------------------------------------------------------------------------------------------------------------------
typedef short RS4 __attribute__((ext_vector_type(4)));
typedef _Float16 RH4 __attribute__((ext_vector_type(4)));
typedef float RF4 __attribute__((ext_vector_type(4)));
typedef uint32_t RW2 __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ RF4 mfma444(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, const RF4& c) {
  const RW2 a = {a0, a1}, b = {b0, b1};
  if constexpr (std::is_same_v<T, BF16>)
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(RS4, a), __builtin_bit_cast(RS4, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(RH4, a), __builtin_bit_cast(RH4, b), c, 0, 0, 0);
}

constexpr int kRs4Unit = 128;  // k per unit
constexpr int kRs4Pad = 32;    // elements between the staged x rows beyond K (64 B: one quarter of the banks)

static inline int64_t rs4_lds_bytes(int M, int N, int K, int G) {
  const int mm = rs_padded_rows(M);
  const int64_t groups = ((int64_t)(N / 4) + G - 1) / G;
  return (int64_t)mm * (K + kRs4Pad) * 2 + groups * kRsWaves * (mm > 4 ? 2 : 1) * 16 * 4;
}

template <typename T, int MM, int D, int MODE>
__global__ __launch_bounds__(kRsThreads, 2) void rowstream4_gemm_kernel(const RowStreamParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
  __shared__ float red[8][16];
  constexpr int NB = MM > 4 ? 2 : 1;  // x rows come four to a B operand
  const int K = p.K, pitch = K + kRs4Pad;
  uint16_t* xs = reinterpret_cast<uint16_t*>(rs_smem);                      // [MM][pitch]
  float* part = reinterpret_cast<float*>(rs_smem + (size_t)MM * pitch * 2);  // [groups][8 waves][NB][j][i]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = sgpr(tid >> 6);
  const int b = lane >> 2, ij = lane & 3;  // block (k piece); row of A / column of B, D
  const int G = gridDim.x, c = blockIdx.x;
  const int NG = p.N >> 2;
  const int g0 = (int)((int64_t)c * NG / G), g1 = (int)((int64_t)(c + 1) * NG / G);
  const int groups = g1 - g0;
  const int UR = K >> 7;          // units per group
  const int total = groups * UR;  // units of this workgroup
  const uint16_t* wg0 = p.w + (int64_t)g0 * 4 * p.ldw;
  const uint32_t lane_off = (uint32_t)(ij * p.ldw + b * 8);  // elements: row ij of the group, k piece b

  auto unit_ptr = [&](int u, int g, int kc) {
    const uint16_t* q = u < total ? wg0 + (int64_t)g * 4 * p.ldw + kc * kRs4Unit : p.w;  // past the end: the matrix's first rows
    return reinterpret_cast<const RW4*>(q + lane_off);
  };
  auto advance = [&](int& u, int& g, int& kc) {
    u += kRsWaves;
    kc += kRsWaves;
    while (kc >= UR) {
      kc -= UR;
      ++g;
    }
  };

  RW4 ring[D];
  int lu = wv, lg = wv / UR, lkc = wv - lg * UR;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    ring[j] = __builtin_nontemporal_load(unit_ptr(lu, lg, lkc));
    advance(lu, lg, lkc);
  }

  for (int i = tid; i < groups * kRsWaves * NB * 16; i += kRsThreads) part[i] = 0.f;

  rs_stage<T, MM, MODE>(p, xs, pitch, red, c, tid, lane, wv);

  RF4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = RF4{0.f, 0.f, 0.f, 0.f};
  // this lane's x row(s): m = column index (+ 4 for the second operand), clamped into the staged rows (MM = 1, 2: the
  // columns beyond are computed on a copy and never stored)
  const uint16_t* xrow[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) xrow[nb] = xs + (size_t)(ij + 4 * nb < MM ? ij + 4 * nb : MM - 1) * pitch + b * 8;
  auto flush = [&](int g) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      RF4 v = acc[nb];
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // add the 16 blocks: lanes 4 b + j over b
        float t = v[e];
        t += dpp_mov<0x128>(t);  // row_ror:8
        t += dpp_mov<0x124>(t);  // row_ror:4
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        v[e] = t;
      }
      if (lane < 4) *reinterpret_cast<RF4*>(part + (((size_t)g * kRsWaves + wv) * NB + nb) * 16 + lane * 4) = v;  // [j][i]
      acc[nb] = RF4{0.f, 0.f, 0.f, 0.f};
    }
  };
  int cu = wv, cg = wv / UR, ckc = wv - cg * UR;
  int cur_g = cg;
  while (cu < total) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if (cu < total) {
        if (cg != cur_g) {
          flush(cur_g);
          cur_g = cg;
        }
        const RW4 wq = ring[j];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const RW4 xv = *reinterpret_cast<const RW4*>(xrow[nb] + ckc * kRs4Unit);
          acc[nb] = mfma444<T>(wq.x, wq.y, xv.x, xv.y, acc[nb]);
          acc[nb] = mfma444<T>(wq.z, wq.w, xv.z, xv.w, acc[nb]);
        }
      }
      ring[j] = __builtin_nontemporal_load(unit_ptr(lu, lg, lkc));
      advance(lu, lg, lkc);
      advance(cu, cg, ckc);
    }
  }
  if (wv < total) flush(cur_g);
  __syncthreads();

  const int rows = groups * 4;
  for (int idx = tid; idx < rows * MM; idx += kRsThreads) {
    const int m = idx / rows, r = idx - m * rows;
    if (m < p.M) {
      const int g = r >> 2, i = r & 3, nb = m >> 2, j = m & 3;
      float s = 0.f;
#pragma unroll
      for (int v2 = 0; v2 < kRsWaves; ++v2) s += part[(((size_t)g * kRsWaves + v2) * NB + nb) * 16 + j * 4 + i];
      p.out[(int64_t)m * p.ldo + 4 * g0 + r] = (uint16_t)Elem<T>::bits(s);
    }
  }
}

template <typename T, int MM, int D, int MODE, int V>
static int launch_rowstream_t(const RowStreamParams& p, int G, size_t lds, hipStream_t s) {
  auto* kernel = V ? &rowstream4_gemm_kernel<T, MM, D, MODE> : &rowstream_gemm_kernel<T, MM, D, MODE>;
  static PerDeviceOnce lds_attr;  // per instantiation and device: more than 64 KB of dynamic LDS has to be requested
  if (!lds_attr.ensure([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kRsLdsBudget) == hipSuccess;
      })) {
    set_error("rowstream_gemm_nt: cannot reserve %d bytes of LDS: %s", kRsLdsBudget, hipGetErrorString(hipGetLastError()));
    return MSGL_ELAUNCH;
  }
  kernel<<<dim3((unsigned)G), dim3(kRsThreads), lds, s>>>(p);
  return MSGL_OK;
}

template <typename T, int V>
static int launch_rowstream(const RowStreamParams& p, int depth, int mode, int G, size_t lds, hipStream_t s) {
  const int mm = rs_padded_rows(p.M);
#define MSGL_RS(MM_, D_, MODE_) \
  if (mm == MM_ && depth == D_ && mode == MODE_) return launch_rowstream_t<T, MM_, D_, MODE_, V>(p, G, lds, s)
#define MSGL_RS_M(D_, MODE_) \
  MSGL_RS(1, D_, MODE_); MSGL_RS(2, D_, MODE_); MSGL_RS(4, D_, MODE_); MSGL_RS(8, D_, MODE_)
  MSGL_RS_M(8, kRsPlain); MSGL_RS_M(16, kRsPlain);
  MSGL_RS_M(8, kRsSiluHalves); MSGL_RS_M(16, kRsSiluHalves);
  MSGL_RS_M(8, kRsSiluIlv); MSGL_RS_M(16, kRsSiluIlv);
  MSGL_RS(1, 8, kRsAddNorm); MSGL_RS(2, 8, kRsAddNorm); MSGL_RS(4, 8, kRsAddNorm);
  MSGL_RS(1, 16, kRsAddNorm); MSGL_RS(2, 16, kRsAddNorm); MSGL_RS(4, 16, kRsAddNorm);
#undef MSGL_RS_M
#undef MSGL_RS
  set_error("rowstream_gemm_nt: no kernel for M = %d, depth %d, mode %d", p.M, depth, mode);
  return MSGL_EINVAL;
}

static int rs_unsupported_reason(int M, int N, int K, int mode, int variant, const char** why) {
  const int G = device_cu_count() > 0 ? device_cu_count() : 256;
  *why = nullptr;
  if (M < 1 || M > 8) *why = "M outside [1, 8]";
  else if (N < 1) *why = "N < 1";
  else if (variant < 0 || variant > 1) *why = "variant outside 0..1";
  else if (variant == 0 && (K < kRsUnit || K % kRsUnit)) *why = "K must be a multiple of 512";
  else if (variant == 1 && (K < kRs4Unit || K % kRs4Unit || N % 4)) *why = "variant 1 needs K a multiple of 128 and N a multiple of 4";
  else if (mode < 0 || mode > 3) *why = "mode outside 0..3";
  else if (mode == kRsAddNorm && (M > 4 || K <= 1024 || K > 8192))
    *why = "mode 3 (fused add + RMSNorm) needs M <= 4 and 1024 < K <= 8192 (the rows rmsnorm_wide_row_kernel takes)";
  else if ((variant ? rs4_lds_bytes(M, N, K, G) : rs_lds_bytes(M, N, K, G)) > kRsLdsBudget)
    *why = "x and the per-row partial sums do not fit the CU's LDS";
  return *why ? 1 : 0;
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_rowstream_gemm_supported(int M, int N, int K, int mode, int variant) {
  const char* why;
  return rs_unsupported_reason(M, N, K, mode, variant, &why) ? 0 : 1;
}

extern "C" int msgl_rowstream_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx,
                                      int64_t ldw, int64_t ldo, int dtype, int depth, int mode, int variant,
                                      const void* res_in, void* res_out, const void* gamma, float eps, int64_t ldr_in,
                                      int64_t ldr_out, void* stream) {
  MSGL_REQUIRE(out && x && w, "rowstream_gemm_nt: null pointer");
  const char* why;
  if (rs_unsupported_reason(M, N, K, mode, variant, &why)) {
    set_error("rowstream_gemm_nt: M = %d, N = %d, K = %d, mode %d, variant %d: %s", M, N, K, mode, variant, why);
    return MSGL_EINVAL;
  }
  MSGL_REQUIRE(depth == 8 || depth == 16, "rowstream_gemm_nt: depth %d (8 or 16 units in flight per wave)", depth);
  const int64_t x_cols = mode == kRsSiluHalves || mode == kRsSiluIlv ? 2ll * K : K;
  MSGL_REQUIRE(ldx >= x_cols && ldw >= K && ldw < (1ll << 28) && ldo >= N && ldx % 8 == 0 && ldw % 8 == 0,
               "rowstream_gemm_nt: leading dimensions (%lld, %lld, %lld)", (long long)ldx, (long long)ldw, (long long)ldo);
  MSGL_REQUIRE(aligned16(x) && aligned16(w) && (reinterpret_cast<uintptr_t>(out) & 1u) == 0,
               "rowstream_gemm_nt: x and w must be 16-byte aligned");
  RowStreamParams p{};
  p.out = (uint16_t*)out;
  p.x = (const uint16_t*)x;
  p.w = (const uint16_t*)w;
  p.M = M; p.N = N; p.K = K;
  p.ldx = ldx; p.ldw = ldw; p.ldo = ldo;
  if (mode == kRsAddNorm) {
    MSGL_REQUIRE(res_in && res_out && gamma, "rowstream_gemm_nt: mode 3 needs res_in, res_out and the norm weight");
    MSGL_REQUIRE(res_in != res_out && res_out != x && out != x && out != res_in,
                 "rowstream_gemm_nt: mode 3: res_out must not alias res_in or x (the other workgroups are still reading them)");
    MSGL_REQUIRE(aligned16(res_in) && aligned16(res_out) && aligned16(gamma) && ldr_in >= K && ldr_out >= K &&
                     ldr_in % 8 == 0 && ldr_out % 8 == 0,
                 "rowstream_gemm_nt: mode 3: residual rows must be 16-byte aligned, strides >= K");
    p.res_in = (const uint16_t*)res_in;
    p.res_out = (uint16_t*)res_out;
    p.gamma = (const uint16_t*)gamma;
    p.eps = eps;
    p.ldr_in = ldr_in;
    p.ldr_out = ldr_out;
  }
  const int G = device_cu_count() > 0 ? device_cu_count() : 256;
  const size_t lds = (size_t)(variant ? rs4_lds_bytes(M, N, K, G) : rs_lds_bytes(M, N, K, G));
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (dtype == MSGL_BF16)
    rc = variant ? launch_rowstream<BF16, 1>(p, depth, mode, G, lds, s) : launch_rowstream<BF16, 0>(p, depth, mode, G, lds, s);
  else if (dtype == MSGL_FP16)
    rc = variant ? launch_rowstream<FP16, 1>(p, depth, mode, G, lds, s) : launch_rowstream<FP16, 0>(p, depth, mode, G, lds, s);
  else {
    set_error("rowstream_gemm_nt: unsupported dtype code %d", dtype);
    return MSGL_EINVAL;
  }
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("rowstream_gemm_nt");
  return MSGL_OK;
}
