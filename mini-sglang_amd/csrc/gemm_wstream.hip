// Weight-streaming projection GEMM for mid-size decode batches (32 < M <= 256):
//     out[M, N] = x[M, K] . w[N, K]^T        (torch F.linear layout, bf16 / fp16, fp32 accumulate)
//
// Same idea as gemm_skinny.hip -- the weight matrix goes HBM -> VGPR exactly once, already in the
// v_mfma_f32_16x16x32 A-operand layout, and is never shared -- but with M up to 256 the activation re-reads
// (every wave needs all M rows of x for its k range) would swamp L2 -> L1.  Here the 8 waves of a workgroup
// walk the SAME k range over DIFFERENT weight rows, so the activation tile of a step is staged once per
// workgroup in LDS and read by all 8 waves as the B operand:
//   * workgroup = 8 waves x NT row tiles x 16 rows; step = 64 k; x tile [16 MT rows][64 k] double-buffered
//     in LDS, 128-B rows with the 16-B chunks XOR-swizzled by the row (conflict-free reads and writes, measured);
//     x goes global -> VGPR -> LDS, requested one step ahead and written after the step's MFMAs;
//   * each wave keeps the weight fragments of the next three steps in flight (a ring of four register sets,
//     NT x 2 loads of 16 B per lane per step); one barrier per step;
//   * grid = (N / (128 NT), k splits): with k splits > 1 every workgroup writes an fp32 slab and a second tiny
//     kernel adds the slabs in split order (deterministic) and rounds once.
// (NT, k splits) and the kernel-vs-library choice are timed per shape by model.tune_gemms.
#include <type_traits>

#include "common.h"

namespace msgl {

typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ws_f16x8 __attribute__((ext_vector_type(8)));
typedef float ws_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t WS4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ ws_f32x4 ws_mfma16(const WS4& a, const WS4& b, const ws_f32x4& c) {
  if constexpr (std::is_same_v<T, BF16>)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ws_bf16x8, a), __builtin_bit_cast(ws_bf16x8, b),
                                                   c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ws_f16x8, a), __builtin_bit_cast(ws_f16x8, b), c,
                                                  0, 0, 0);
}

constexpr int kWsWaves = 8;
constexpr int kWsStepK = 64;                     // k per step
constexpr int kWsPitch = kWsStepK * 2;           // LDS row pitch in bytes (128); 16-B chunk c of row r sits at c ^ (r & 7)

// MT = 16-token column tiles (M <= 16 MT; 4, 8 or 16), NT = 16-row weight tiles per wave (1 or 2).
// PARTIAL: write fp32 slabs part[blockIdx.y][M][N] instead of the rounded output.
template <typename T, int MT, int NT, bool PARTIAL>
__global__ __launch_bounds__(64 * kWsWaves) void wstream_gemm_kernel(
    uint16_t* __restrict__ out, float* __restrict__ part, const uint16_t* __restrict__ x,
    const uint16_t* __restrict__ w, int M, int N, int nsteps, int64_t ldx, int64_t ldw, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xs[];  // [2][16 MT][kWsPitch]
  constexpr int kRows = 16 * MT;
  constexpr int kTileBytes = kRows * kWsPitch;
  constexpr int kChunks = kRows * (kWsStepK / 8);           // 16-B chunks of one x tile
  constexpr int kPerThread = kChunks / (64 * kWsWaves);     // MT / 4
  static_assert(kChunks % (64 * kWsWaves) == 0, "x tile must divide over the workgroup");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const int KS = gridDim.y;
  const int s0 = (int)((int64_t)blockIdx.y * nsteps / KS), s1 = (int)((int64_t)(blockIdx.y + 1) * nsteps / KS);
  const int64_t n0 = ((int64_t)blockIdx.x * kWsWaves + wv) * (16 * NT);
  const uint16_t* wp = w + (n0 + r) * ldw + kg * 8;  // row tile i: + 16 i ldw; k32 group j of a step: + 32 j

  // x tile loader: chunk q -> (row q / 8, 16-B piece q % 8 of the row's 128 B)
  const uint16_t* xg[kPerThread];
  int xl[kPerThread];
#pragma unroll
  for (int p = 0; p < kPerThread; ++p) {
    const int q = tid + p * (64 * kWsWaves);
    const int row = q / (kWsStepK / 8), c = q % (kWsStepK / 8);
    xg[p] = x + (int64_t)min(row, M - 1) * ldx + c * 8;
    xl[p] = row * kWsPitch + ((c ^ (row & 7)) * 16);  // XOR swizzle: see compute()
  }

  ws_f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[i][t] = ws_f32x4{0.f, 0.f, 0.f, 0.f};

  struct Frag {
    WS4 a[NT][2];  // [row tile][k32 group of the step]
  };
  auto load_w = [&](Frag& f, int step) {
    const int64_t k = (int64_t)step * kWsStepK;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) f.a[i][j] = *reinterpret_cast<const WS4*>(wp + (int64_t)i * 16 * ldw + k + j * 32);
  };
  struct XRegs {
    WS4 v[kPerThread];
  };
  auto load_x = [&](XRegs& xr, int step) {
    const int64_t k = (int64_t)step * kWsStepK;
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) xr.v[p] = *reinterpret_cast<const WS4*>(xg[p] + k);
  };
  auto store_x = [&](const XRegs& xr, int buf) {
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) *reinterpret_cast<WS4*>(xs + buf * kTileBytes + xl[p]) = xr.v[p];
  };
  // B fragment of (column tile t, k32 group j): lane (r, kg) reads 16 B of row 16 t + r, chunk 4 j + kg, stored at
  // chunk position (4 j + kg) ^ (r & 7).  Measured with tools/lds_probe.hip: this layout reads and writes at
  // the conflict-free rate; a padded 144-B pitch costs +50 % on the reads (PMC: 44 % of the LDS cycles of the
  // first version were bank-conflict cycles) because the LDS services lanes of two k-groups in the same cycle.
  const int sw0 = ((kg ^ (r & 7)) * 16), sw1 = (((4 + kg) ^ (r & 7)) * 16);
  auto compute = [&](const Frag& f, int buf) {
    const unsigned char* base = xs + buf * kTileBytes + r * kWsPitch;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const WS4 b = *reinterpret_cast<const WS4*>(base + t * 16 * kWsPitch + (j ? sw1 : sw0));
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i][t] = ws_mfma16<T>(f.a[i][j], b, acc[i][t]);
      }
    }
  };
  // one step: request the x tile of step+2 and the weight fragments of step+3 (clamped: the last steps
  // re-request the last one), compute `cur`, publish the x tile of step+1 (requested one step earlier).
  // vmcnt retires in order, so the x tile that store_x waits for must be OLDER than the weight loads meant to
  // stay in flight: requested a step early it only forces W(step+1) complete and leaves W(step+2), W(step+3)
  // outstanding.  (One step of prefetch left a wave with loads outstanding only for the ~2 us after each issue,
  // idle until the next step: 3.9 TB/s at M = 64.)
  auto body = [&](const Frag& cur, Frag& far, int buf, int step, const XRegs& x_have, XRegs& x_next) {
    load_x(x_next, min(step + 2, s1 - 1));
    load_w(far, min(step + 3, s1 - 1));
    // pin: the compiler otherwise sinks these requests below the MFMAs (fewer live registers)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    compute(cur, buf);
    store_x(x_have, buf ^ 1);
    __syncthreads();
  };

  if (s0 < s1) {
    Frag f0, f1, f2, f3;
    XRegs xa, xb;
    load_x(xb, s0);
    load_w(f0, s0);
    load_w(f1, min(s0 + 1, s1 - 1));
    load_w(f2, min(s0 + 2, s1 - 1));
    load_x(xa, min(s0 + 1, s1 - 1));
    store_x(xb, 0);
    __syncthreads();
    int step = s0;
    for (; step + 4 <= s1; step += 4) {
      body(f0, f3, 0, step, xa, xb);
      body(f1, f0, 1, step + 1, xb, xa);
      body(f2, f1, 0, step + 2, xa, xb);
      body(f3, f2, 1, step + 3, xb, xa);
    }
    if (step < s1) body(f0, f3, 0, step, xa, xb);
    if (step + 1 < s1) body(f1, f0, 1, step + 1, xb, xa);
    if (step + 2 < s1) body(f2, f1, 0, step + 2, xa, xb);
  }

  // epilogue: lane holds D[n = tile + kg 4 + e][m = 16 t + r]
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = t * 16 + r;
      if (m < M) {
        const int64_t n = n0 + i * 16 + kg * 4;
        const ws_f32x4 v = acc[i][t];
        if constexpr (PARTIAL) {
          *reinterpret_cast<ws_f32x4*>(part + ((int64_t)blockIdx.y * M + m) * N + n) = v;
        } else {
          uint2 pk;
          pk.x = Elem<T>::pack(v.x, v.y);
          pk.y = Elem<T>::pack(v.z, v.w);
          *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + n) = pk;
        }
      }
    }
  }
}

// out[m][n] = round(sum_s part[s][m][n]), slabs added in split order; 8 columns per thread
template <typename T>
__global__ __launch_bounds__(256) void wstream_reduce_kernel(uint16_t* __restrict__ out, const float* __restrict__ part,
                                                             int M, int N, int KS, int64_t ldo) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = N >> 3;
  if (gid >= (int64_t)M * per_row) return;
  const int m = (int)(gid / per_row), c = (int)(gid - (int64_t)m * per_row);
  const float* p = part + (int64_t)m * N + c * 8;
  float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  for (int s = 1; s < KS; ++s) {
    const float* q = p + (int64_t)s * M * N;
    const float4 a2 = *reinterpret_cast<const float4*>(q), b2 = *reinterpret_cast<const float4*>(q + 4);
    a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
    b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
  }
  U4 u;
  u.x = Elem<T>::pack(a.x, a.y); u.y = Elem<T>::pack(a.z, a.w);
  u.z = Elem<T>::pack(b.x, b.y); u.w = Elem<T>::pack(b.z, b.w);
  stg16(out + (int64_t)m * ldo + c * 8, u);
}

template <typename T, int MT, int NT>
static int launch_wstream_t(uint16_t* out, float* part, const uint16_t* x, const uint16_t* w, int M, int N, int K,
                            int64_t ldx, int64_t ldw, int64_t ldo, int k_splits, hipStream_t s, bool slabs_only) {
  const dim3 grid((unsigned)(N / (16 * NT * kWsWaves)), (unsigned)k_splits), block(64 * kWsWaves);
  const size_t lds = 2u * 16 * MT * kWsPitch;
  const int nsteps = K / kWsStepK;
  static PerDeviceOnce lds_attr;  // per instantiation and device: more than 64 KB of dynamic LDS has to be requested
  if (!lds_attr.ensure([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&wstream_gemm_kernel<T, MT, NT, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(&wstream_gemm_kernel<T, MT, NT, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      })) {
    set_error("wstream_gemm_nt: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(hipGetLastError()));
    return MSGL_ELAUNCH;
  }
  if (k_splits == 1) {
    wstream_gemm_kernel<T, MT, NT, false><<<grid, block, lds, s>>>(out, nullptr, x, w, M, N, nsteps, ldx, ldw, ldo);
  } else {
    wstream_gemm_kernel<T, MT, NT, true><<<grid, block, lds, s>>>(out, part, x, w, M, N, nsteps, ldx, ldw, ldo);
    if (slabs_only) return MSGL_OK;  // the consumer (a slab-input norm / qk pass) adds the slabs, in split order
    const int64_t threads = (int64_t)M * (N / 8);
    wstream_reduce_kernel<T><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>(out, part, M, N, k_splits,
                                                                                          ldo);
  }
  return MSGL_OK;
}

template <typename T>
static int launch_wstream(uint16_t* out, float* part, const uint16_t* x, const uint16_t* w, int M, int N, int K,
                          int64_t ldx, int64_t ldw, int64_t ldo, int row_tiles, int k_splits, hipStream_t s,
                          bool slabs_only = false) {
  const int MT = M <= 64 ? 4 : M <= 128 ? 8 : 16;
#define MSGL_WS(MT_, NT_) \
  if (MT == MT_ && row_tiles == NT_) \
    return launch_wstream_t<T, MT_, NT_>(out, part, x, w, M, N, K, ldx, ldw, ldo, k_splits, s, slabs_only)
  // (16, 2) -- 256 rows x two weight tiles per wave -- needs 128 accumulator registers on top of the four-deep weight
  // ring and spilled 91-96 VGPRs: not built; M > 128 takes one row tile (or the full-batch kernels of gemm_g3 / m256)
  MSGL_WS(4, 1); MSGL_WS(4, 2); MSGL_WS(8, 1); MSGL_WS(8, 2); MSGL_WS(16, 1);
#undef MSGL_WS
  set_error("wstream_gemm_nt: row_tiles %d unsupported at M = %d (1, or 2 up to M = 128)", row_tiles, M);
  return MSGL_EINVAL;
}

}  // namespace msgl

using namespace msgl;

extern "C" int64_t msgl_wstream_gemm_workspace_bytes(int M, int N, int k_splits) {
  if (M < 1 || N < 1 || k_splits < 1) return MSGL_EINVAL;
  return k_splits > 1 ? (int64_t)k_splits * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int msgl_wstream_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx,
                                    int64_t ldw, int64_t ldo, int dtype, int row_tiles, int k_splits,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  MSGL_REQUIRE(out && x && w, "wstream_gemm_nt: null pointer");
  MSGL_REQUIRE(M >= 1 && M <= 256, "wstream_gemm_nt: M = %d outside [1, 256]", M);
  MSGL_REQUIRE(row_tiles == 1 || row_tiles == 2, "wstream_gemm_nt: row_tiles %d (1, 2)", row_tiles);
  MSGL_REQUIRE(N >= 128 * row_tiles && N % (128 * row_tiles) == 0, "wstream_gemm_nt: N = %d must be a multiple of %d",
               N, 128 * row_tiles);
  MSGL_REQUIRE(K >= kWsStepK && K % kWsStepK == 0, "wstream_gemm_nt: K = %d must be a multiple of %d", K, kWsStepK);
  MSGL_REQUIRE(k_splits >= 1 && k_splits <= K / kWsStepK && k_splits <= 64, "wstream_gemm_nt: %d k splits", k_splits);
  MSGL_REQUIRE(ldx >= K && ldw >= K && ldo >= N && ldx % 8 == 0 && ldw % 8 == 0 && ldo % 8 == 0,
               "wstream_gemm_nt: leading dimensions (%lld, %lld, %lld)", (long long)ldx, (long long)ldw,
               (long long)ldo);
  MSGL_REQUIRE(aligned16(x) && aligned16(w) && aligned16(out), "wstream_gemm_nt: pointers must be 16-byte aligned");
  if (k_splits > 1) {
    MSGL_REQUIRE(workspace && aligned16(workspace) &&
                     workspace_bytes >= msgl_wstream_gemm_workspace_bytes(M, N, k_splits),
                 "wstream_gemm_nt: %d k splits need %lld workspace bytes", k_splits,
                 (long long)msgl_wstream_gemm_workspace_bytes(M, N, k_splits));
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (dtype == MSGL_BF16)
    rc = launch_wstream<BF16>((uint16_t*)out, (float*)workspace, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx,
                              ldw, ldo, row_tiles, k_splits, s);
  else if (dtype == MSGL_FP16)
    rc = launch_wstream<FP16>((uint16_t*)out, (float*)workspace, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx,
                              ldw, ldo, row_tiles, k_splits, s);
  else {
    set_error("wstream_gemm_nt: unsupported dtype code %d", dtype);
    return MSGL_EINVAL;
  }
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("wstream_gemm_nt");
  return MSGL_OK;
}

// The k-split product WITHOUT its reduce: fp32 slabs part[k_splits][M][N] in `workspace`, to be added in split order and
// rounded by the consumer (msgl_fused_add_rmsnorm_slabs / msgl_qk_norm_rope_store_slabs), exactly what
// msgl_wstream_gemm_nt's own reduce launch would have stored.  k_splits >= 2.
extern "C" int msgl_wstream_gemm_slabs_nt(const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                                          int dtype, int row_tiles, int k_splits, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  MSGL_REQUIRE(x && w && workspace, "wstream_gemm_slabs_nt: null pointer");
  MSGL_REQUIRE(M >= 1 && M <= 256, "wstream_gemm_slabs_nt: M = %d outside [1, 256]", M);
  MSGL_REQUIRE(row_tiles == 1 || row_tiles == 2, "wstream_gemm_slabs_nt: row_tiles %d (1, 2)", row_tiles);
  MSGL_REQUIRE(N >= 128 * row_tiles && N % (128 * row_tiles) == 0, "wstream_gemm_slabs_nt: N = %d must be a multiple of %d",
               N, 128 * row_tiles);
  MSGL_REQUIRE(K >= kWsStepK && K % kWsStepK == 0, "wstream_gemm_slabs_nt: K = %d must be a multiple of %d", K, kWsStepK);
  MSGL_REQUIRE(k_splits >= 2 && k_splits <= K / kWsStepK && k_splits <= 64, "wstream_gemm_slabs_nt: %d k splits", k_splits);
  MSGL_REQUIRE(ldx >= K && ldw >= K && ldx % 8 == 0 && ldw % 8 == 0, "wstream_gemm_slabs_nt: leading dimensions (%lld, %lld)",
               (long long)ldx, (long long)ldw);
  MSGL_REQUIRE(aligned16(x) && aligned16(w) && aligned16(workspace) &&
                   workspace_bytes >= msgl_wstream_gemm_workspace_bytes(M, N, k_splits),
               "wstream_gemm_slabs_nt: %d k splits need %lld aligned workspace bytes", k_splits,
               (long long)msgl_wstream_gemm_workspace_bytes(M, N, k_splits));
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (dtype == MSGL_BF16)
    rc = launch_wstream<BF16>(nullptr, (float*)workspace, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx, ldw, N,
                              row_tiles, k_splits, s, true);
  else if (dtype == MSGL_FP16)
    rc = launch_wstream<FP16>(nullptr, (float*)workspace, (const uint16_t*)x, (const uint16_t*)w, M, N, K, ldx, ldw, N,
                              row_tiles, k_splits, s, true);
  else {
    set_error("wstream_gemm_slabs_nt: unsupported dtype code %d", dtype);
    return MSGL_EINVAL;
  }
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("wstream_gemm_slabs_nt");
  return MSGL_OK;
}
