// Element-wise / per-row kernels of the decoder layer: RMSNorm, fused-add RMSNorm,
// NeoX RoPE, SiLU*mul, and the fused qk-norm + RoPE + KV-store pass.
//
// All are HBM-bound: every global access is a 16-byte (8 x 16-bit) lane access on
// contiguous row segments, math is fp32, rows are reduced with DPP adds (no LDS)
// whenever a row fits in <= 16 lanes, wave shuffles up to 64 lanes, LDS only for the
// 256-thread hidden-size rows.
#include "common.h"

namespace msgl {

// sum over the TPR threads that share a row (TPR = 8, 16, 64 or 256)
template <int TPR>
__device__ __forceinline__ float row_sum(float x, float* lds) {
  if constexpr (TPR == 8) return row8_sum(x);
  if constexpr (TPR == 16) return row16_sum(x);
  if constexpr (TPR == 64) return wave_sum(x);
  if constexpr (TPR == 256) {
    x = wave_sum(x);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[w] = x;
    __syncthreads();
    return lds[0] + lds[1] + lds[2] + lds[3];
  }
  return x;
}

// ------------------------------------------------------------------------------
// RMSNorm over rows of a logical [n0, n1, dim] tensor.  TPR threads per row, each
// thread keeps up to KMAX 8-element pieces in registers (dim <= TPR*KMAX*8).
// FUSED: x += residual (fp32), residual <- rounded sum, x <- norm(fp32 sum) * w.
// SLABS (with FUSED): x is not read from memory but assembled from `n_slabs` fp32 partial-sum slabs of a split-K
// projection (slab s at slabs + s * slab_stride, row stride slab_ld), added in slab order and rounded to the
// 16-bit type first -- exactly what the projection's own reduce kernel would have stored -- so that the reduce
// launch disappears without changing a bit of the result.
// ------------------------------------------------------------------------------
template <typename T, int TPR, int KMAX, bool FUSED, bool SLABS = false>
__global__ __launch_bounds__(256) void rmsnorm_kernel(uint16_t* out, const uint16_t* x,
                                                      uint16_t* residual,
                                                      const uint16_t* __restrict__ weight,
                                                      float eps, int64_t rows, int64_t n1, int dim,
                                                      int64_t xs0, int64_t xs1, int64_t os0,
                                                      int64_t os1, int64_t rs0, const float* __restrict__ slabs = nullptr,
                                                      int n_slabs = 0, int64_t slab_stride = 0, int64_t slab_ld = 0) {
  __shared__ float lds[4];
  constexpr int kRowsPerBlock = 256 / TPR;
  const int tir = threadIdx.x % TPR;  // thread in row
  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x / TPR;
  const bool live = row < rows;  // dead rows still take part in the reductions
  const int64_t r = live ? row : 0;
  const int64_t i0 = r / n1, i1 = r - i0 * n1;
  const uint16_t* xp = x + i0 * xs0 + i1 * xs1;
  uint16_t* op = out + i0 * os0 + i1 * os1;
  uint16_t* rp = FUSED ? residual + r * rs0 : nullptr;
  const int pieces = dim >> 3;

  float v[KMAX][8];
  U4 wq[KMAX];  // weight pieces, requested before the reduction so their latency hides behind it
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int p = tir + k * TPR;
    if (p < pieces) wq[k] = ldg16(weight + p * 8);
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int p = tir + k * TPR;
    if (p < pieces) {
      if constexpr (SLABS) {
        const float* sp = slabs + r * slab_ld + p * 8;
        float4 a = *reinterpret_cast<const float4*>(sp), b = *reinterpret_cast<const float4*>(sp + 4);
        for (int sl = 1; sl < n_slabs; ++sl) {
          const float* q = sp + (int64_t)sl * slab_stride;
          const float4 a2 = *reinterpret_cast<const float4*>(q), b2 = *reinterpret_cast<const float4*>(q + 4);
          a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
          b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unpack8<T>(pack8<T>(f), v[k]);  // the rounded projection output, as the reduce kernel stores it
      } else {
        unpack8<T>(ldg16(xp + p * 8), v[k]);
      }
      if constexpr (FUSED) {
        float rr[8];
        unpack8<T>(ldg16(rp + p * 8), rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[k][e] += rr[e];
        if (live) stg16(rp + p * 8, pack8<T>(v[k]));
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = fmaf(v[k][e], v[k][e], ss);
    }
  }
  ss = row_sum<TPR>(ss, lds);
  const float inv = rsqrtf(ss / (float)dim + eps);
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int p = tir + k * TPR;
    if (p < pieces) {
      float w[8], y[8];
      unpack8<T>(wq[k], w);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = __fmul_rn(__fmul_rn(v[k][e], inv), w[e]);
      if (live) stg16(op + p * 8, pack8<T>(y));
    }
  }
}

// ------------------------------------------------------------------------------
// Fused-add RMSNorm of WIDE rows (hidden sizes: more than 128 pieces): ONE ROW PER WORKGROUP, one 8-element piece per
// thread (blockDim = pieces rounded up to whole waves, <= 1024); with slab input the slab loop is unrolled NS-fold so
// that every load of a thread -- NS x 2 float4 of partial sums, residual, weight -- is in flight at once.
// rmsnorm_kernel above would run such a row on 256 threads x 3 pieces with a runtime slab loop: 4 waves per CU and one
// slab in flight per piece, 3.6 TB/s on the 39 MB a 256 x 5120 decode batch with 6 slabs moves.  EVERY fused-add norm
// of a wide row takes this kernel, with or without slabs and at any row count, so that the slab hand-off stays
// bit-identical to reduce-then-norm and a row's result does not depend on the batch it is in (the per-row sum of
// squares is added in one fixed order: lanes by wave_sum, waves in index order).
// ------------------------------------------------------------------------------
template <typename T, int NS, bool SLABS>
__global__ __launch_bounds__(1024) void rmsnorm_wide_row_kernel(uint16_t* x, uint16_t* residual,
                                                                const uint16_t* __restrict__ weight, float eps, int dim,
                                                                int64_t xs, int64_t rs, const float* __restrict__ slabs,
                                                                int n_slabs, int64_t slab_stride, int64_t slab_ld) {
  __shared__ float lds[16];
  const int64_t r = blockIdx.x;
  const int p = threadIdx.x;
  const int pieces = dim >> 3;
  const bool has = p < pieces;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  U4 wq = U4{0, 0, 0, 0};
  float ss = 0.f;
  if (has) {
    U4 rq;
    if constexpr (SLABS) {
      const float* sp = slabs + r * slab_ld + p * 8;
      float4 a[NS], b[NS];
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        const float* q = sp + (int64_t)(sl < n_slabs ? sl : 0) * slab_stride;  // clamped: loads stay unconditional
        a[sl] = *reinterpret_cast<const float4*>(q);
        b[sl] = *reinterpret_cast<const float4*>(q + 4);
      }
      rq = ldg16(residual + r * rs + p * 8);
      wq = ldg16(weight + p * 8);
      float4 sa = a[0], sb = b[0];
#pragma unroll
      for (int sl = 1; sl < NS; ++sl)
        if (sl < n_slabs) {
          sa.x += a[sl].x; sa.y += a[sl].y; sa.z += a[sl].z; sa.w += a[sl].w;
          sb.x += b[sl].x; sb.y += b[sl].y; sb.z += b[sl].z; sb.w += b[sl].w;
        }
      for (int sl = NS; sl < n_slabs; ++sl) {  // more slabs than the unrolled part
        const float* q = sp + (int64_t)sl * slab_stride;
        const float4 a2 = *reinterpret_cast<const float4*>(q), b2 = *reinterpret_cast<const float4*>(q + 4);
        sa.x += a2.x; sa.y += a2.y; sa.z += a2.z; sa.w += a2.w;
        sb.x += b2.x; sb.y += b2.y; sb.z += b2.z; sb.w += b2.w;
      }
      const float f[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
      unpack8<T>(pack8<T>(f), v);  // the rounded projection output, as the reduce kernel stores it
    } else {
      const U4 xq = ldg16(x + r * xs + p * 8);
      rq = ldg16(residual + r * rs + p * 8);
      wq = ldg16(weight + p * 8);
      unpack8<T>(xq, v);
    }
    float rr[8];
    unpack8<T>(rq, rr);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += rr[e];
    stg16(residual + r * rs + p * 8, pack8<T>(v));
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
  }
  ss = wave_sum(ss);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) lds[w] = ss;
  __syncthreads();
  float tot = 0.f;
  for (int i = 0; i < nw; ++i) tot += lds[i];
  const float inv = rsqrtf(tot / (float)dim + eps);
  if (has) {
    float wv[8], y[8];
    unpack8<T>(wq, wv);
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = __fmul_rn(__fmul_rn(v[e], inv), wv[e]);
    stg16(x + r * xs + p * 8, pack8<T>(y));
  }
}

// ------------------------------------------------------------------------------
// NeoX RoPE in place.  One thread rotates 8 (low half) + 8 (high half) elements.
// ------------------------------------------------------------------------------
template <typename T, typename PosT>
__global__ __launch_bounds__(256) void rope_kernel(uint16_t* __restrict__ q, uint16_t* __restrict__ k,
                                                   const PosT* __restrict__ positions,
                                                   const float* __restrict__ cache, int64_t total,
                                                   int hq, int hk, int dim, int64_t qs, int64_t ks) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int half_pieces = dim >> 4;  // 8-element pieces in half a head
  const int per_tok = (hq + hk) * half_pieces;
  const int64_t t = gid / per_tok;
  const int rem = (int)(gid - t * per_tok);
  const int head = rem / half_pieces;
  const int j = rem - head * half_pieces;
  uint16_t* base = head < hq ? q + t * qs + (int64_t)head * dim : k + t * ks + (int64_t)(head - hq) * dim;
  const int half = dim >> 1;
  const float* cs = cache + (int64_t)positions[t] * dim + j * 8;
  float a[8], b[8], c[8], s[8];
  unpack8<T>(ldg16(base + j * 8), a);
  unpack8<T>(ldg16(base + half + j * 8), b);
  *reinterpret_cast<float4*>(c) = *reinterpret_cast<const float4*>(cs);
  *reinterpret_cast<float4*>(c + 4) = *reinterpret_cast<const float4*>(cs + 4);
  *reinterpret_cast<float4*>(s) = *reinterpret_cast<const float4*>(cs + half);
  *reinterpret_cast<float4*>(s + 4) = *reinterpret_cast<const float4*>(cs + half + 4);
  float ra[8], rb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    // explicit rounding points: the fused kernel below must produce the same bits
    ra[e] = fmaf(a[e], c[e], -__fmul_rn(b[e], s[e]));
    rb[e] = fmaf(b[e], c[e], __fmul_rn(a[e], s[e]));
  }
  stg16(base + j * 8, pack8<T>(ra));
  stg16(base + half + j * 8, pack8<T>(rb));
}

// ------------------------------------------------------------------------------
// out[t, j] = act(x[t, j]) * x[t, d + j];  act = silu, or (GELU) the exact erf GELU 0.5 g (1 + erf(g / sqrt 2))
// ------------------------------------------------------------------------------
// ILV: the columns of x are a gate_up row interleaved in blocks of 32 (gate, up, gate, up per 128 columns -- the layout
// csrc/gemm_g3.hip wants for its fused epilogue): output piece j (8 columns) reads gate at 128 (j / 8) + 64 ((j % 8) / 4)
// + 8 (j % 4) and up 32 columns further.
template <typename T, bool GELU, bool ILV>
__global__ __launch_bounds__(256) void silu_mul_kernel(uint16_t* __restrict__ out,
                                                       const uint16_t* __restrict__ x, int64_t total,
                                                       int pieces, int64_t d, int64_t xs, int64_t os) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int64_t t = gid / pieces;
  const int64_t j = gid - t * pieces;
  float g[8], u[8], y[8];
  if constexpr (ILV) {
    const int64_t col = (j >> 3) * 128 + ((j >> 2) & 1) * 64 + (j & 3) * 8;
    unpack8<T>(ldg16(x + t * xs + col), g);
    unpack8<T>(ldg16(x + t * xs + col + 32), u);
  } else {
    unpack8<T>(ldg16(x + t * xs + j * 8), g);
    unpack8<T>(ldg16(x + t * xs + d + j * 8), u);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if constexpr (GELU)
      y[e] = 0.5f * g[e] * (1.0f + erff(g[e] * 0.70710678118654752f)) * u[e];
    else
      y[e] = silu_mul_f32(g[e], u[e]);
  }
  stg16(out + t * os + j * 8, pack8<T>(y));
}

// ------------------------------------------------------------------------------
// Fused pass over one token's qkv row: [rmsnorm per head] -> NeoX RoPE -> write q,k in
// place, scatter k,v to the pool.  LPR = head_dim/8 lanes own one (token, head).
// ------------------------------------------------------------------------------
// SLABS: the qkv row has not been written yet -- the qkv projection left `n_slabs` fp32 k-slice sums of it (slab s at
// slabs + s * slab_stride, row stride slab_ld, columns in qkv order).  A piece is their sum in slab order rounded to
// the 16-bit type, i.e. exactly what the projection's reduce kernel would have stored; v is then written in place as
// well, so that the qkv buffer ends up bit-identical to the two-kernel path.
template <typename T, int LPR, typename PosT, typename LocT, bool SLABS = false>
__global__ __launch_bounds__(256) void qk_norm_rope_store_kernel(
    uint16_t* __restrict__ q, uint16_t* __restrict__ k, uint16_t* __restrict__ v,
    const uint16_t* __restrict__ qw, const uint16_t* __restrict__ kw, float eps,
    const PosT* __restrict__ positions, const float* __restrict__ cache, uint16_t* __restrict__ kc,
    uint16_t* __restrict__ vc, const LocT* __restrict__ out_loc, int64_t groups, int hq, int hk,
    int64_t qs, int64_t ks, int64_t vs, int64_t cs, const float* __restrict__ slabs = nullptr, int n_slabs = 0,
    int64_t slab_stride = 0, int64_t slab_ld = 0) {
  constexpr int D = LPR * 8;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t grp = gid / LPR;
  const int c = (int)(gid % LPR);
  const bool live = grp < groups;
  const int64_t g = live ? grp : 0;
  const int per_tok = hq + 2 * hk;
  const int64_t t = g / per_tok;
  const int head = (int)(g - t * per_tok);
  auto piece = [&](const uint16_t* src) -> U4 {  // this lane's 8 elements of the projected row
    if constexpr (SLABS) {
      const float* sp = slabs + t * slab_ld + (int64_t)head * D + c * 8;
      float4 a = *reinterpret_cast<const float4*>(sp), b = *reinterpret_cast<const float4*>(sp + 4);
      for (int sl = 1; sl < n_slabs; ++sl) {
        const float* r = sp + (int64_t)sl * slab_stride;
        const float4 a2 = *reinterpret_cast<const float4*>(r), b2 = *reinterpret_cast<const float4*>(r + 4);
        a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
        b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
      }
      const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      return pack8<T>(f);
    } else {
      return ldg16(src);
    }
  };
  if (head >= hq + hk) {  // V head: plain copy into the pool
    const int h = head - hq - hk;
    if (live) {
      uint16_t* vp = v + t * vs + (int64_t)h * D + c * 8;
      const U4 val = piece(vp);
      if constexpr (SLABS) stg16(vp, val);
      stg16(vc + (int64_t)out_loc[t] * cs + (int64_t)h * D + c * 8, val);
    }
    return;  // whole LPR-lane group leaves together; DPP below only spans such groups
  }
  const bool is_q = head < hq;
  uint16_t* p = is_q ? q + t * qs + (int64_t)head * D + c * 8
                     : k + t * ks + (int64_t)(head - hq) * D + c * 8;
  const uint16_t* w = is_q ? qw : kw;
  float x[8];
  unpack8<T>(piece(p), x);
  if (w != nullptr) {
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(x[e], x[e], ss);
    ss = (LPR == 16) ? row16_sum(ss) : row8_sum(ss);
    const float inv = rsqrtf(ss / (float)D + eps);
    float wf[8];
    unpack8<T>(ldg16(w + c * 8), wf);
    // round through the storage type: the reference writes the normed value back
    // before RoPE reads it (P/layers/attention.py:50-54)
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = __fmul_rn(__fmul_rn(x[e], inv), wf[e]);
    unpack8<T>(pack8<T>(y), x);
  }
  // partner lane holds the other half of the head (i <-> i + D/2)
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = __shfl_xor(x[e], LPR / 2, 64);
  const int j = c & (LPR / 2 - 1);
  const float* csr = cache + (int64_t)positions[t] * D + j * 8;
  float co[8], si[8];
  *reinterpret_cast<float4*>(co) = *reinterpret_cast<const float4*>(csr);
  *reinterpret_cast<float4*>(co + 4) = *reinterpret_cast<const float4*>(csr + 4);
  *reinterpret_cast<float4*>(si) = *reinterpret_cast<const float4*>(csr + D / 2);
  *reinterpret_cast<float4*>(si + 4) = *reinterpret_cast<const float4*>(csr + D / 2 + 4);
  const float sgn = c < LPR / 2 ? -1.f : 1.f;
  float y[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = fmaf(x[e], co[e], sgn * __fmul_rn(o[e], si[e]));
  const U4 outv = pack8<T>(y);
  if (live) {
    stg16(p, outv);
    if (!is_q) stg16(kc + (int64_t)out_loc[t] * cs + (int64_t)(head - hq) * D + c * 8, outv);
  }
}

template <typename F>
static int dispatch_dtype(int dtype, F&& f) {
  if (dtype == MSGL_BF16) return f(BF16{});
  if (dtype == MSGL_FP16) return f(FP16{});
  set_error("unsupported dtype code %d (want MSGL_BF16 or MSGL_FP16)", dtype);
  return MSGL_EINVAL;
}

template <typename T, bool FUSED, bool SLABS = false>
static int launch_rmsnorm(uint16_t* out, const uint16_t* x, uint16_t* res, const uint16_t* w, float eps,
                          int64_t rows, int64_t n1, int64_t dim, int64_t xs0, int64_t xs1, int64_t os0,
                          int64_t os1, int64_t rs0, hipStream_t s, const float* slabs = nullptr, int n_slabs = 0,
                          int64_t slab_stride = 0, int64_t slab_ld = 0) {
  const int pieces = (int)(dim / 8);
  if constexpr (FUSED) {
    if (pieces > 128 && pieces <= 1024 && n1 == 1 && x == out) {  // hidden-size rows: one row per workgroup (see the kernel)
      const unsigned threads = (unsigned)((pieces + 63) / 64 * 64);
      MSGL_REQUIRE(rows < (1ll << 31), "fused_add_rmsnorm: too many rows");
      if constexpr (SLABS) {
        if (n_slabs <= 4)
          rmsnorm_wide_row_kernel<T, 4, true><<<dim3((unsigned)rows), dim3(threads), 0, s>>>(
              out, res, w, eps, (int)dim, xs0, rs0, slabs, n_slabs, slab_stride, slab_ld);
        else
          rmsnorm_wide_row_kernel<T, 8, true><<<dim3((unsigned)rows), dim3(threads), 0, s>>>(
              out, res, w, eps, (int)dim, xs0, rs0, slabs, n_slabs, slab_stride, slab_ld);
      } else {
        rmsnorm_wide_row_kernel<T, 1, false><<<dim3((unsigned)rows), dim3(threads), 0, s>>>(
            out, res, w, eps, (int)dim, xs0, rs0, nullptr, 0, 0, 0);
      }
      return MSGL_OK;
    }
  }
#define MSGL_NORM(TPR, KMAX)                                                                        \
  do {                                                                                              \
    const int64_t rpb = 256 / TPR;                                                                  \
    const int64_t blocks = (rows + rpb - 1) / rpb;                                                  \
    rmsnorm_kernel<T, TPR, KMAX, FUSED, SLABS><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(       \
        out, x, res, w, eps, rows, n1, (int)dim, xs0, xs1, os0, os1, rs0, slabs, n_slabs,           \
        slab_stride, slab_ld);                                                                      \
  } while (0)
  if (pieces <= 8) MSGL_NORM(8, 1);
  else if (pieces <= 16) MSGL_NORM(16, 1);
  else if (pieces <= 128) MSGL_NORM(64, 2);
  else if (pieces <= 1024) MSGL_NORM(256, 4);
  else MSGL_NORM(256, 8);
#undef MSGL_NORM
  return MSGL_OK;
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_rmsnorm(void* out, const void* x, const void* weight, float eps, int64_t n0,
                            int64_t n1, int64_t dim, int64_t x_stride0, int64_t x_stride1,
                            int64_t out_stride0, int64_t out_stride1, int dtype, void* stream) {
  MSGL_REQUIRE(n0 >= 0 && n1 >= 1, "rmsnorm: bad row counts");
  if (n0 == 0) return MSGL_OK;
  MSGL_REQUIRE(out && x && weight, "rmsnorm: null pointer");
  MSGL_REQUIRE(dim > 0 && dim % 8 == 0 && dim <= 16384, "rmsnorm: dim %lld must be a multiple of 8, <= 16384",
               (long long)dim);
  MSGL_REQUIRE(x_stride0 % 8 == 0 && x_stride1 % 8 == 0 && out_stride0 % 8 == 0 && out_stride1 % 8 == 0,
               "rmsnorm: strides must be multiples of 8 elements");
  MSGL_REQUIRE(aligned16(out) && aligned16(x) && aligned16(weight), "rmsnorm: pointers must be 16-byte aligned");
  const int64_t rows = n0 * n1;
  MSGL_REQUIRE(rows < (1ll << 31), "rmsnorm: too many rows");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_rmsnorm<T, false>((uint16_t*)out, (const uint16_t*)x, nullptr, (const uint16_t*)weight, eps,
                                    rows, n1, dim, x_stride0, x_stride1, out_stride0, out_stride1, 0, s);
  });
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("rmsnorm");
  return MSGL_OK;
}

extern "C" int msgl_fused_add_rmsnorm(void* x, void* residual, const void* weight, float eps,
                                      int64_t rows, int64_t dim, int64_t x_stride, int64_t res_stride,
                                      int dtype, void* stream) {
  MSGL_REQUIRE(rows >= 0, "fused_add_rmsnorm: negative rows");
  if (rows == 0) return MSGL_OK;
  MSGL_REQUIRE(x && residual && weight, "fused_add_rmsnorm: null pointer");
  MSGL_REQUIRE(dim > 0 && dim % 8 == 0 && dim <= 16384,
               "fused_add_rmsnorm: dim %lld must be a multiple of 8, <= 16384", (long long)dim);
  MSGL_REQUIRE(x_stride % 8 == 0 && res_stride % 8 == 0, "fused_add_rmsnorm: strides must be multiples of 8");
  MSGL_REQUIRE(aligned16(x) && aligned16(residual) && aligned16(weight),
               "fused_add_rmsnorm: pointers must be 16-byte aligned");
  MSGL_REQUIRE(rows < (1ll << 31), "fused_add_rmsnorm: too many rows");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_rmsnorm<T, true>((uint16_t*)x, (const uint16_t*)x, (uint16_t*)residual,
                                   (const uint16_t*)weight, eps, rows, 1, dim, x_stride, 0, x_stride, 0,
                                   res_stride, s);
  });
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("fused_add_rmsnorm");
  return MSGL_OK;
}

extern "C" int msgl_rope_neox_inplace(void* q, void* k, const void* positions, int positions_is_i64,
                                      const float* cos_sin_cache, int64_t num_tokens, int num_q_heads,
                                      int num_k_heads, int head_dim, int64_t q_stride, int64_t k_stride,
                                      int dtype, void* stream) {
  MSGL_REQUIRE(num_tokens >= 0, "rope: negative token count");
  if (num_tokens == 0) return MSGL_OK;
  MSGL_REQUIRE(q && k && positions && cos_sin_cache, "rope: null pointer");
  MSGL_REQUIRE(head_dim == 64 || head_dim == 128 || head_dim == 256 || head_dim == 512,
               "rope: head_dim %d not in {64,128,256,512}", head_dim);
  MSGL_REQUIRE(num_q_heads >= 0 && num_k_heads >= 0 && num_q_heads + num_k_heads > 0, "rope: bad head counts");
  MSGL_REQUIRE(q_stride % 8 == 0 && k_stride % 8 == 0, "rope: strides must be multiples of 8 elements");
  MSGL_REQUIRE(aligned16(q) && aligned16(k) && aligned16(cos_sin_cache), "rope: pointers must be 16-byte aligned");
  const int64_t total = num_tokens * (num_q_heads + num_k_heads) * (head_dim / 16);
  const int64_t blocks = (total + 255) / 256;
  MSGL_REQUIRE(blocks < (1ll << 31), "rope: too many tokens");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    if (positions_is_i64)
      rope_kernel<T, int64_t><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
          (uint16_t*)q, (uint16_t*)k, (const int64_t*)positions, cos_sin_cache, total, num_q_heads,
          num_k_heads, head_dim, q_stride, k_stride);
    else
      rope_kernel<T, int32_t><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
          (uint16_t*)q, (uint16_t*)k, (const int32_t*)positions, cos_sin_cache, total, num_q_heads,
          num_k_heads, head_dim, q_stride, k_stride);
    return MSGL_OK;
  });
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("rope_neox_inplace");
  return MSGL_OK;
}

extern "C" int msgl_fused_add_rmsnorm_slabs(void* x, void* residual, const void* weight, float eps, int64_t rows,
                                            int64_t dim, int64_t x_stride, int64_t res_stride, const float* slabs,
                                            int num_slabs, int64_t slab_stride, int64_t slab_ld, int dtype,
                                            void* stream) {
  MSGL_REQUIRE(rows >= 0, "fused_add_rmsnorm_slabs: negative rows");
  if (rows == 0) return MSGL_OK;
  MSGL_REQUIRE(x && residual && weight && slabs, "fused_add_rmsnorm_slabs: null pointer");
  MSGL_REQUIRE(dim > 0 && dim % 8 == 0 && dim <= 16384,
               "fused_add_rmsnorm_slabs: dim %lld must be a multiple of 8, <= 16384", (long long)dim);
  MSGL_REQUIRE(num_slabs >= 1 && num_slabs <= 64 && slab_ld >= dim && slab_ld % 4 == 0 && slab_stride % 4 == 0,
               "fused_add_rmsnorm_slabs: %d slabs, ld %lld, stride %lld", num_slabs, (long long)slab_ld,
               (long long)slab_stride);
  MSGL_REQUIRE(x_stride % 8 == 0 && res_stride % 8 == 0, "fused_add_rmsnorm_slabs: strides must be multiples of 8");
  MSGL_REQUIRE(aligned16(x) && aligned16(residual) && aligned16(weight) && aligned16(slabs),
               "fused_add_rmsnorm_slabs: pointers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_rmsnorm<T, true, true>((uint16_t*)x, (const uint16_t*)x, (uint16_t*)residual, (const uint16_t*)weight,
                                         eps, rows, 1, dim, x_stride, 0, x_stride, 0, res_stride, s, slabs, num_slabs,
                                         slab_stride, slab_ld);
  });
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("fused_add_rmsnorm_slabs");
  return MSGL_OK;
}

template <bool GELU, bool ILV = false>
static int act_and_mul(const char* what, void* out, const void* x, int64_t num_tokens, int64_t d, int64_t x_stride,
                       int64_t out_stride, int dtype, void* stream) {
  MSGL_REQUIRE(num_tokens >= 0, "%s: negative token count", what);
  if (num_tokens == 0) return MSGL_OK;
  MSGL_REQUIRE(out && x, "%s: null pointer", what);
  MSGL_REQUIRE(d > 0 && d % (ILV ? 64 : 8) == 0, "%s: d %lld must be a multiple of %d", what, (long long)d, ILV ? 64 : 8);
  MSGL_REQUIRE(x_stride % 8 == 0 && out_stride % 8 == 0, "%s: strides must be multiples of 8", what);
  MSGL_REQUIRE(aligned16(out) && aligned16(x), "%s: pointers must be 16-byte aligned", what);
  const int pieces = (int)(d / 8);
  const int64_t total = num_tokens * pieces;
  const int64_t blocks = (total + 255) / 256;
  MSGL_REQUIRE(blocks < (1ll << 31), "%s: too many tokens", what);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    silu_mul_kernel<T, GELU, ILV><<<dim3((unsigned)blocks), dim3(256), 0, s>>>((uint16_t*)out, (const uint16_t*)x, total,
                                                                           pieces, d, x_stride, out_stride);
    return MSGL_OK;
  });
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH(what);
  return MSGL_OK;
}

extern "C" int msgl_silu_and_mul(void* out, const void* x, int64_t num_tokens, int64_t d, int64_t x_stride,
                                 int64_t out_stride, int dtype, void* stream) {
  return act_and_mul<false>("silu_and_mul", out, x, num_tokens, d, x_stride, out_stride, dtype, stream);
}

extern "C" int msgl_silu_and_mul_interleaved(void* out, const void* x, int64_t num_tokens, int64_t d, int64_t x_stride,
                                             int64_t out_stride, int dtype, void* stream) {
  return act_and_mul<false, true>("silu_and_mul_interleaved", out, x, num_tokens, d, x_stride, out_stride, dtype, stream);
}

extern "C" int msgl_gelu_and_mul(void* out, const void* x, int64_t num_tokens, int64_t d, int64_t x_stride,
                                 int64_t out_stride, int dtype, void* stream) {
  return act_and_mul<true>("gelu_and_mul", out, x, num_tokens, d, x_stride, out_stride, dtype, stream);
}

static int qk_norm_rope_store_impl(void* q, void* k, void* v, const void* q_norm_w, const void* k_norm_w, float eps,
                                   const void* positions, int positions_is_i64, const float* cos_sin_cache,
                                   void* k_cache, void* v_cache, const void* out_loc, int out_loc_is_i64,
                                   int64_t num_tokens, int num_q_heads, int num_k_heads, int head_dim,
                                   int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t cache_stride,
                                   int dtype, void* stream, const float* slabs, int n_slabs, int64_t slab_stride,
                                   int64_t slab_ld) {
  MSGL_REQUIRE(num_tokens >= 0, "qk_norm_rope_store: negative token count");
  if (num_tokens == 0) return MSGL_OK;
  MSGL_REQUIRE(q && k && v && positions && cos_sin_cache && k_cache && v_cache && out_loc,
               "qk_norm_rope_store: null pointer");
  MSGL_REQUIRE(head_dim == 64 || head_dim == 128, "qk_norm_rope_store: head_dim %d not in {64,128}", head_dim);
  MSGL_REQUIRE((q_norm_w == nullptr) == (k_norm_w == nullptr), "qk_norm_rope_store: need both or no norm weights");
  MSGL_REQUIRE(q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0 && cache_stride % 8 == 0,
               "qk_norm_rope_store: strides must be multiples of 8 elements");
  MSGL_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(k_cache) && aligned16(v_cache) &&
                   aligned16(cos_sin_cache),
               "qk_norm_rope_store: pointers must be 16-byte aligned");
  MSGL_REQUIRE(positions_is_i64 == 0 || positions_is_i64 == 1, "qk_norm_rope_store: bad positions flag");
  const int lpr = head_dim / 8;
  const int64_t groups = num_tokens * (num_q_heads + 2 * num_k_heads);
  const int64_t blocks = (groups * lpr + 255) / 256;
  MSGL_REQUIRE(blocks < (1ll << 31), "qk_norm_rope_store: too many tokens");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
#define MSGL_QKRS_S(LPR, PT, LT, SL)                                                                   \
  qk_norm_rope_store_kernel<T, LPR, PT, LT, SL><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(         \
      (uint16_t*)q, (uint16_t*)k, (uint16_t*)v, (const uint16_t*)q_norm_w, (const uint16_t*)k_norm_w, \
      eps, (const PT*)positions, cos_sin_cache, (uint16_t*)k_cache, (uint16_t*)v_cache, (const LT*)out_loc, \
      groups, num_q_heads, num_k_heads, q_stride, k_stride, v_stride, cache_stride, slabs, n_slabs,    \
      slab_stride, slab_ld)
#define MSGL_QKRS(LPR, PT, LT)                 \
  do {                                         \
    if (slabs) MSGL_QKRS_S(LPR, PT, LT, true); \
    else MSGL_QKRS_S(LPR, PT, LT, false);      \
  } while (0)
#define MSGL_QKRS_L(LPR)                                                  \
  do {                                                                    \
    if (positions_is_i64) {                                               \
      if (out_loc_is_i64) MSGL_QKRS(LPR, int64_t, int64_t);               \
      else MSGL_QKRS(LPR, int64_t, int32_t);                              \
    } else {                                                              \
      if (out_loc_is_i64) MSGL_QKRS(LPR, int32_t, int64_t);               \
      else MSGL_QKRS(LPR, int32_t, int32_t);                              \
    }                                                                     \
  } while (0)
    if (lpr == 16) MSGL_QKRS_L(16); else MSGL_QKRS_L(8);
#undef MSGL_QKRS_L
#undef MSGL_QKRS
#undef MSGL_QKRS_S
    return MSGL_OK;
  });
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("qk_norm_rope_store");
  return MSGL_OK;
}

extern "C" int msgl_qk_norm_rope_store(void* q, void* k, const void* v, const void* q_norm_w,
                                       const void* k_norm_w, float eps, const void* positions,
                                       int positions_is_i64, const float* cos_sin_cache, void* k_cache,
                                       void* v_cache, const void* out_loc, int out_loc_is_i64,
                                       int64_t num_tokens, int num_q_heads, int num_k_heads, int head_dim,
                                       int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                       int64_t cache_stride, int dtype, void* stream) {
  return qk_norm_rope_store_impl(q, k, const_cast<void*>(v), q_norm_w, k_norm_w, eps, positions, positions_is_i64,
                                 cos_sin_cache, k_cache, v_cache, out_loc, out_loc_is_i64, num_tokens, num_q_heads,
                                 num_k_heads, head_dim, q_stride, k_stride, v_stride, cache_stride, dtype, stream, nullptr,
                                 0, 0, 0);
}

extern "C" int msgl_qk_norm_rope_store_slabs(void* qkv, int64_t qkv_stride, const float* slabs, int num_slabs,
                                             int64_t slab_stride, int64_t slab_ld, const void* q_norm_w,
                                             const void* k_norm_w, float eps, const void* positions,
                                             int positions_is_i64, const float* cos_sin_cache, void* k_cache,
                                             void* v_cache, const void* out_loc, int out_loc_is_i64,
                                             int64_t num_tokens, int num_q_heads, int num_k_heads, int head_dim,
                                             int64_t cache_stride, int dtype, void* stream) {
  MSGL_REQUIRE(qkv && slabs, "qk_norm_rope_store_slabs: null pointer");
  const int64_t width = (int64_t)(num_q_heads + 2 * num_k_heads) * head_dim;
  MSGL_REQUIRE(num_slabs >= 1 && num_slabs <= 64 && slab_ld >= width && slab_ld % 4 == 0 && slab_stride % 4 == 0 &&
                   aligned16(slabs) && qkv_stride >= width,
               "qk_norm_rope_store_slabs: %d slabs, ld %lld, stride %lld, row %lld", num_slabs, (long long)slab_ld,
               (long long)slab_stride, (long long)qkv_stride);
  uint16_t* base = static_cast<uint16_t*>(qkv);
  return qk_norm_rope_store_impl(base, base + (int64_t)num_q_heads * head_dim,
                                 base + (int64_t)(num_q_heads + num_k_heads) * head_dim, q_norm_w, k_norm_w, eps, positions,
                                 positions_is_i64, cos_sin_cache, k_cache, v_cache, out_loc, out_loc_is_i64, num_tokens,
                                 num_q_heads, num_k_heads, head_dim, qkv_stride, qkv_stride, qkv_stride, cache_stride, dtype,
                                 stream, slabs, num_slabs, slab_stride, slab_ld);
}
