// Sampling kernels: greedy argmax, temperature softmax, exact top-k / top-p sampling.
//
// All three walk [rows, vocab] once or a few times with one workgroup per row; the row
// (<= 608 KB fp32) stays L2-resident between passes.  Top-k / top-p uses no sort: the k-th
// largest probability and the top-p mass cut are found by a 3-level radix select over the
// fp32 bit pattern (monotone for non-negative floats) with LDS histograms.  Every sum that
// decides a threshold or the sampled index is an INTEGER (fixed-point mass, counts), so the
// result is independent of atomic ordering: all TP ranks, which hold identical logits and
// the same Philox (seed, offset), draw the same token (reference: every rank seeds 42,
// P/engine/engine.py:37).
#include "common.h"

namespace msgl {

typedef unsigned long long u64;

// ------------------------------------------------------------------------------------------
// argmax: first index of the row maximum
// ------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ float load_logit(const void* row, int64_t i) {
  if constexpr (DT == MSGL_F32) return static_cast<const float*>(row)[i];
  if constexpr (DT == MSGL_BF16) return __uint_as_float((uint32_t) static_cast<const uint16_t*>(row)[i] << 16);
  return (float)static_cast<const _Float16*>(row)[i];
}

// 8 consecutive logits starting at an 8-aligned index (16-byte loads)
template <int DT>
__device__ __forceinline__ void load_logits8(const void* row, int64_t i, float (&f)[8]) {
  if constexpr (DT == MSGL_F32) {
    const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(row) + i);
    const float4 b = *reinterpret_cast<const float4*>(static_cast<const float*>(row) + i + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    const U4 u = ldg16(static_cast<const uint16_t*>(row) + i);
    using E = Elem<std::conditional_t<DT == MSGL_BF16, BF16, FP16>>;
    f[0] = E::lo(u.x); f[1] = E::hi(u.x); f[2] = E::lo(u.y); f[3] = E::hi(u.y);
    f[4] = E::lo(u.z); f[5] = E::hi(u.z); f[6] = E::lo(u.w); f[7] = E::hi(u.w);
  }
}

constexpr int kRowThreads = 1024;

template <int DT, bool VEC>
__global__ __launch_bounds__(kRowThreads) void argmax_kernel(int* __restrict__ out,
                                                             const char* __restrict__ logits, int64_t vocab,
                                                             int64_t row_stride_bytes) {
  __shared__ float s_val[kRowThreads / 64];
  __shared__ int s_idx[kRowThreads / 64];
  const void* row = logits + (int64_t)blockIdx.x * row_stride_bytes;
  const int tid = threadIdx.x;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  if constexpr (VEC) {
    const int64_t nvec = vocab >> 3;
    for (int64_t p = tid; p < nvec; p += kRowThreads) {
      float f[8];
      load_logits8<DT>(row, p * 8, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (f[e] > best) { best = f[e]; bidx = (int)(p * 8 + e); }
      }
    }
    for (int64_t i = (nvec << 3) + tid; i < vocab; i += kRowThreads) {
      const float x = load_logit<DT>(row, i);
      if (x > best) { best = x; bidx = (int)i; }
    }
  } else {
    for (int64_t i = tid; i < vocab; i += kRowThreads) {
      const float x = load_logit<DT>(row, i);
      if (x > best) { best = x; bidx = (int)i; }
    }
  }
  // (value desc, index asc) reduction
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bidx, off, 64);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if ((tid & 63) == 0) { s_val[tid >> 6] = best; s_idx[tid >> 6] = bidx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kRowThreads / 64; ++w) {
      if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bidx)) { best = s_val[w]; bidx = s_idx[w]; }
    }
    out[blockIdx.x] = bidx == 0x7fffffff ? 0 : bidx;
  }
}

// ------------------------------------------------------------------------------------------
// probs = softmax(logits / T)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce_max(float x, float* lds) {
  x = wave_max(x);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = x;
  __syncthreads();
  float r = lds[0];
  for (int w = 1; w < kRowThreads / 64; ++w) r = fmaxf(r, lds[w]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_reduce_sum(float x, float* lds) {
  x = wave_sum(x);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = x;
  __syncthreads();
  float r = 0.f;
  for (int w = 0; w < kRowThreads / 64; ++w) r += lds[w];  // fixed order: deterministic
  __syncthreads();
  return r;
}

template <int DT>
__global__ __launch_bounds__(kRowThreads) void softmax_kernel(float* __restrict__ probs,
                                                              const char* __restrict__ logits,
                                                              const float* __restrict__ temperatures,
                                                              int64_t vocab, int64_t logits_stride_bytes,
                                                              int64_t probs_stride) {
  __shared__ float lds[kRowThreads / 64];
  const void* row = logits + (int64_t)blockIdx.x * logits_stride_bytes;
  float* prow = probs + (int64_t)blockIdx.x * probs_stride;
  const float inv_t = 1.0f / temperatures[blockIdx.x];
  const int tid = threadIdx.x;
  float mx = -INFINITY;
  for (int64_t i = tid; i < vocab; i += kRowThreads) mx = fmaxf(mx, load_logit<DT>(row, i));
  mx = block_reduce_max(mx, lds);
  const float k = inv_t * 1.4426950408889634f;
  float sum = 0.f;
  for (int64_t i = tid; i < vocab; i += kRowThreads) {
    const float e = __builtin_amdgcn_exp2f((load_logit<DT>(row, i) - mx) * k);
    prow[i] = e;
    sum += e;
  }
  sum = block_reduce_sum(sum, lds);
  const float inv = 1.0f / sum;
  for (int64_t i = tid; i < vocab; i += kRowThreads) prow[i] *= inv;
}

// ------------------------------------------------------------------------------------------
// top-k / top-p sampling from probs
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 fixed_mass(float p) {  // 2^-40 resolution, exact scaling
  return p > 0.f ? (u64)__float2ull_rz(p * 1099511627776.0f) : 0ull;
}

__device__ __forceinline__ u64 shfl_up_u64(u64 v, int d) {
  const uint32_t lo = __shfl_up((uint32_t)v, d, 64);
  const uint32_t hi = __shfl_up((uint32_t)(v >> 32), d, 64);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
  const uint32_t lo = __shfl((uint32_t)v, src, 64);
  const uint32_t hi = __shfl((uint32_t)(v >> 32), src, 64);
  return ((u64)hi << 32) | lo;
}

// inclusive prefix sum over the 1024 threads (thread order), exact integer arithmetic
__device__ __forceinline__ u64 block_prefix_incl(u64 v, u64* lds_w) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u64 o = shfl_up_u64(v, d);
    if (lane >= d) v += o;
  }
  if (lane == 63) lds_w[w] = v;
  __syncthreads();
  u64 base = 0;
  for (int i = 0; i < w; ++i) base += lds_w[i];
  __syncthreads();
  return v + base;
}

__device__ __forceinline__ u64 block_sum_u64(u64 v, u64* lds_w) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, d, 64);
    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), d, 64);
    v += ((u64)hi << 32) | lo;
  }
  if (lane == 0) lds_w[w] = v;
  __syncthreads();
  u64 r = 0;
  for (int i = 0; i < kRowThreads / 64; ++i) r += lds_w[i];
  __syncthreads();
  return r;
}

struct Philox {
  static __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const u64 p0 = (u64)0xD2511F53u * c[0];
    const u64 p1 = (u64)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  // Philox4x32-10, first output word
  static __device__ __forceinline__ uint32_t draw(u64 seed, u64 ctr) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    return c[0];
  }
};

constexpr int kBins = 2048;  // 11-bit digits; thread t owns bins 2047-2t and 2046-2t (descending)
constexpr uint32_t kNoDigit = 0xffffffffu;

// One radix-select level.  Elements taking part: bits >= floor_bits and (bits & mask) == prefix.
// BY_MASS = false: find the digit where the descending element COUNT reaches `target`;
// BY_MASS = true : where the descending fixed-point MASS reaches `target`.
// Returns the digit; `target` is reduced by what lies above it, `mass_above` accumulates the
// mass of everything strictly above the chosen digit, `digit_mass` = mass inside the digit.
template <bool BY_MASS>
__device__ __forceinline__ uint32_t select_level(const float* __restrict__ row, int64_t vocab,
                                                 uint32_t floor_bits, uint32_t mask, uint32_t prefix,
                                                 int shift, int nbits, u64& target, u64& mass_above,
                                                 u64& digit_mass, uint32_t* h_cnt, u64* h_mass, u64* lds_w,
                                                 uint32_t* s_digit) {
  const int tid = threadIdx.x;
  const uint32_t dmask = (1u << nbits) - 1u;
  for (int b = tid; b < kBins; b += kRowThreads) {
    h_cnt[b] = 0u;
    h_mass[b] = 0ull;
  }
  if (tid == 0) *s_digit = kNoDigit;  // stays if the row holds less than `target` (fewer than k positive entries)
  __syncthreads();
  for (int64_t i = tid; i < vocab; i += kRowThreads) {
    const float p = row[i];
    const uint32_t bits = __float_as_uint(p);
    if (bits >= floor_bits && (bits & mask) == prefix && p > 0.f) {
      const uint32_t d = (bits >> shift) & dmask;
      atomicAdd(&h_cnt[d], 1u);
      atomicAdd(&h_mass[d], fixed_mass(p));
    }
  }
  __syncthreads();
  const int b_hi = kBins - 1 - 2 * tid, b_lo = b_hi - 1;  // descending ownership
  const u64 q_hi = BY_MASS ? h_mass[b_hi] : (u64)h_cnt[b_hi];
  const u64 q_lo = BY_MASS ? h_mass[b_lo] : (u64)h_cnt[b_lo];
  const u64 incl = block_prefix_incl(q_hi + q_lo, lds_w);  // quantity in bins >= b_lo
  const u64 above_hi = incl - q_hi - q_lo;                 // quantity in bins > b_hi
  const u64 above_lo = above_hi + q_hi;                    // quantity in bins > b_lo
  if (above_hi < target && target <= above_hi + q_hi) *s_digit = (uint32_t)b_hi;
  if (above_lo < target && target <= above_lo + q_lo) *s_digit = (uint32_t)b_lo;
  __syncthreads();
  const uint32_t digit = *s_digit;
  if (digit == kNoDigit) {  // uniform: every thread read the same word
    __syncthreads();
    return kNoDigit;
  }
  // mass strictly above the digit, and quantity strictly above (to reduce the target)
  u64 m_above = 0, q_above = 0;
  if ((uint32_t)b_hi > digit) { m_above += h_mass[b_hi]; q_above += q_hi; }
  if ((uint32_t)b_lo > digit) { m_above += h_mass[b_lo]; q_above += q_lo; }
  m_above = block_sum_u64(m_above, lds_w);
  q_above = block_sum_u64(q_above, lds_w);
  mass_above += m_above;
  target -= q_above;
  digit_mass = h_mass[digit];
  __syncthreads();
  return digit;
}

__global__ __launch_bounds__(kRowThreads) void sample_kernel(int* __restrict__ out,
                                                             const float* __restrict__ probs,
                                                             const int* __restrict__ top_k,
                                                             const float* __restrict__ top_p, int64_t vocab,
                                                             int64_t stride, u64 seed, u64 offset) {
  __shared__ uint32_t h_cnt[kBins];
  __shared__ u64 h_mass[kBins];
  __shared__ u64 lds_w[kRowThreads / 64];
  __shared__ uint32_t s_digit;
  __shared__ int s_pick;
  const int tid = threadIdx.x;
  const int64_t r = blockIdx.x;
  const float* row = probs + r * stride;

  uint32_t thr_bits = 0u;  // keep set = { bits >= thr_bits, p > 0 }
  u64 kept_mass = 0;
  bool have_mass = false;

  const int64_t k = top_k ? (int64_t)top_k[r] : vocab;
  if (k >= 1 && k < vocab) {
    u64 target = (u64)k, above = 0, dm = 0;
    uint32_t prefix = 0u, mask = 0u;
    const uint32_t d1 = select_level<false>(row, vocab, 0u, mask, prefix, 21, 11, target, above, dm, h_cnt,
                                            h_mass, lds_w, &s_digit);
    // fewer than k positive probabilities (a low temperature underflows the tail to 0): top-k keeps all of them,
    // as flashinfer does -- no threshold, mass computed below if top-p needs it
    if (d1 != kNoDigit) {
      prefix |= d1 << 21; mask |= 0x7ffu << 21;
      const uint32_t d2 = select_level<false>(row, vocab, 0u, mask, prefix, 10, 11, target, above, dm, h_cnt,
                                              h_mass, lds_w, &s_digit);
      prefix |= d2 << 10; mask |= 0x7ffu << 10;
      const uint32_t d3 = select_level<false>(row, vocab, 0u, mask, prefix, 0, 10, target, above, dm, h_cnt,
                                              h_mass, lds_w, &s_digit);
      prefix |= d3;
      thr_bits = prefix;  // the k-th largest value; ties at it stay in the keep set
      kept_mass = above + dm;
      have_mass = true;
    }
  }
  const float pp = top_p ? top_p[r] : 1.0f;
  if (pp < 1.0f) {
    if (!have_mass) {
      u64 loc = 0;
      for (int64_t i = tid; i < vocab; i += kRowThreads) loc += fixed_mass(row[i]);
      kept_mass = block_sum_u64(loc, lds_w);
    }
    u64 target = (u64)((double)pp * (double)kept_mass);
    if (target < 1) target = 1;
    if (target > kept_mass) target = kept_mass;
    if (kept_mass > 0) {
      u64 above = 0, dm = 0;
      uint32_t prefix = 0u, mask = 0u;
      const uint32_t floor_bits = thr_bits;
      const uint32_t d1 = select_level<true>(row, vocab, floor_bits, mask, prefix, 21, 11, target, above, dm,
                                             h_cnt, h_mass, lds_w, &s_digit);
      if (d1 != kNoDigit) {  // (target <= kept_mass, so a digit exists; the guard keeps a bad row from indexing out of range)
        prefix |= d1 << 21; mask |= 0x7ffu << 21;
        const uint32_t d2 = select_level<true>(row, vocab, floor_bits, mask, prefix, 10, 11, target, above, dm,
                                               h_cnt, h_mass, lds_w, &s_digit);
        prefix |= d2 << 10; mask |= 0x7ffu << 10;
        const uint32_t d3 = select_level<true>(row, vocab, floor_bits, mask, prefix, 0, 10, target, above, dm,
                                               h_cnt, h_mass, lds_w, &s_digit);
        prefix |= d3;
        thr_bits = prefix > thr_bits ? prefix : thr_bits;
        kept_mass = above + dm;
        have_mass = true;
      }
    }
  }

  // ---- inverse-CDF draw in index order over the keep set -------------------------------
  // wave w owns the contiguous index range [w*seg, (w+1)*seg), walked 64 indices at a time
  const int lane = tid & 63, w = tid >> 6;
  constexpr int kWaves = kRowThreads / 64;
  const int64_t seg = ((vocab + kWaves - 1) / kWaves + 63) / 64 * 64;
  const int64_t lo = (int64_t)w * seg, hi = min(vocab, lo + seg);
  u64 wave_mass = 0;
  for (int64_t i = lo + lane; i < hi; i += 64) {
    const float p = row[i];
    if (__float_as_uint(p) >= thr_bits) wave_mass += fixed_mass(p);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t l = __shfl_xor((uint32_t)wave_mass, d, 64);
    const uint32_t h = __shfl_xor((uint32_t)(wave_mass >> 32), d, 64);
    wave_mass += ((u64)h << 32) | l;
  }
  if (lane == 0) lds_w[w] = wave_mass;
  if (tid == 0) s_pick = -1;
  __syncthreads();
  u64 total = 0, before = 0;
  for (int i = 0; i < kWaves; ++i) {
    if (i < w) before += lds_w[i];
    total += lds_w[i];
  }
  const uint32_t x = Philox::draw(seed, offset + (u64)r);
  // u in [0, total): (total * x) >> 32 without 128-bit arithmetic
  const u64 u = (total >> 32) * (u64)x + (((total & 0xffffffffull) * (u64)x) >> 32);
  if (total > 0 && u >= before && u < before + wave_mass) {  // exactly one wave
    u64 run = before;
    for (int64_t base = lo; base < hi; base += 64) {
      const int64_t i = base + lane;
      u64 m = 0;
      if (i < hi) {
        const float p = row[i];
        if (__float_as_uint(p) >= thr_bits) m = fixed_mass(p);
      }
      u64 incl = m;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const u64 o = shfl_up_u64(incl, d);
        if (lane >= d) incl += o;
      }
      const u64 chunk = shfl_u64(incl, 63);
      if (u < run + chunk) {
        const bool mine = m > 0 && u >= run + incl - m && u < run + incl;
        if (mine) s_pick = (int)i;
        break;
      }
      run += chunk;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int pick = s_pick;
    if (pick < 0) {  // degenerate row (all-zero mass): fall back to the first index
      pick = 0;
    }
    out[r] = pick;
  }
}


// ------------------------------------------------------------------------------------------
// sample directly from logits (temperature only: no top-k / top-p filter)
//   index ~ softmax(logits / T).  Probabilities are never materialised: with e_i = 2^((x_i - max) k)
//   the draw is the inverse CDF over the integer masses floor(e_i 2^40) in index order, which is the
//   same distribution as normalise-then-draw up to fp32 rounding of the normalisation (and, like the
//   probs path, exact integer arithmetic from there on: identical on every TP rank).
//   Two passes over the row (max; masses per wave segment) + one wave re-walking its own segment,
//   16-byte loads throughout.  Wave w owns the contiguous index range [w seg, (w+1) seg).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 exp_mass(float x, float mx, float k) {
  return (u64)__float2ull_rz(__builtin_amdgcn_exp2f((x - mx) * k) * 1099511627776.0f);  // e in (0, 1]
}

template <int DT>
__global__ __launch_bounds__(kRowThreads) void sample_logits_kernel(int* __restrict__ out,
                                                                    const char* __restrict__ logits,
                                                                    const float* __restrict__ temperatures,
                                                                    int64_t vocab, int64_t row_stride_bytes,
                                                                    u64 seed, u64 offset) {
  __shared__ float lds_f[kRowThreads / 64];
  __shared__ u64 lds_w[kRowThreads / 64];
  __shared__ int s_pick;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t r = blockIdx.x;
  const void* row = logits + r * row_stride_bytes;
  constexpr int kWaves = kRowThreads / 64;
  constexpr int kStep = 64 * 8;  // indices per wave iteration
  const int64_t seg = ((vocab + kWaves - 1) / kWaves + kStep - 1) / kStep * kStep;
  const int64_t lo = (int64_t)w * seg, hi = min(vocab, lo + seg);

  float mx = -INFINITY;
  for (int64_t base = lo + lane * 8; base < hi; base += kStep) {
    if (base + 8 <= hi) {
      float f[8];
      load_logits8<DT>(row, base, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, f[e]);
    } else {
      for (int64_t i = base; i < hi; ++i) mx = fmaxf(mx, load_logit<DT>(row, i));
    }
  }
  mx = block_reduce_max(mx, lds_f);
  const float k = 1.4426950408889634f / temperatures[r];

  auto masses8 = [&](int64_t base, u64 (&mm)[8]) {  // masses of indices base .. base+7 (0 past hi)
    if (base + 8 <= hi) {
      float f[8];
      load_logits8<DT>(row, base, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) mm[e] = exp_mass(f[e], mx, k);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) mm[e] = base + e < hi ? exp_mass(load_logit<DT>(row, base + e), mx, k) : 0ull;
    }
  };
  u64 wave_mass = 0;
  for (int64_t base = lo + lane * 8; base < hi; base += kStep) {
    u64 mm[8];
    masses8(base, mm);
#pragma unroll
    for (int e = 0; e < 8; ++e) wave_mass += mm[e];
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t l32 = __shfl_xor((uint32_t)wave_mass, d, 64);
    const uint32_t h32 = __shfl_xor((uint32_t)(wave_mass >> 32), d, 64);
    wave_mass += ((u64)h32 << 32) | l32;
  }
  if (lane == 0) lds_w[w] = wave_mass;
  if (tid == 0) s_pick = -1;
  __syncthreads();
  u64 total = 0, before = 0;
  for (int i = 0; i < kWaves; ++i) {
    if (i < w) before += lds_w[i];
    total += lds_w[i];
  }
  const uint32_t x = Philox::draw(seed, offset + (u64)r);
  const u64 u = (total >> 32) * (u64)x + (((total & 0xffffffffull) * (u64)x) >> 32);  // [0, total)
  if (total > 0 && u >= before && u < before + wave_mass) {  // exactly one wave
    u64 run = before;
    for (int64_t base0 = lo; base0 < hi; base0 += kStep) {
      u64 mm[8];
      masses8(base0 + lane * 8, mm);
      u64 mine = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) mine += mm[e];
      u64 incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const u64 o = shfl_up_u64(incl, d);
        if (lane >= d) incl += o;
      }
      const u64 chunk = shfl_u64(incl, 63);
      if (u < run + chunk) {
        u64 start = run + incl - mine;  // mass before this lane's 8 indices
        if (mine > 0 && u >= start && u < start + mine) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (u >= start && u < start + mm[e]) s_pick = (int)(base0 + lane * 8 + e);
            start += mm[e];
          }
        }
        break;
      }
      run += chunk;
    }
  }
  __syncthreads();
  if (tid == 0) out[r] = s_pick < 0 ? 0 : s_pick;  // degenerate row (no mass): first index
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_argmax_rows(int32_t* out, const void* logits, int64_t rows, int64_t vocab,
                                int64_t row_stride, int logits_dtype, void* stream) {
  MSGL_REQUIRE(rows >= 0, "argmax_rows: negative rows");
  if (rows == 0) return MSGL_OK;
  MSGL_REQUIRE(out && logits, "argmax_rows: null pointer");
  MSGL_REQUIRE(vocab >= 1 && vocab < (1ll << 31) && rows < (1ll << 31), "argmax_rows: bad shape");
  const int esz = logits_dtype == MSGL_F32 ? 4 : 2;
  const bool vec = aligned16(logits) && (row_stride * esz) % 16 == 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 g((unsigned)rows), b(kRowThreads);
#define MSGL_ARGMAX(DT)                                                                                      \
  do {                                                                                                       \
    if (vec) argmax_kernel<DT, true><<<g, b, 0, s>>>(out, (const char*)logits, vocab, row_stride * esz);     \
    else argmax_kernel<DT, false><<<g, b, 0, s>>>(out, (const char*)logits, vocab, row_stride * esz);        \
  } while (0)
  if (logits_dtype == MSGL_F32) MSGL_ARGMAX(MSGL_F32);
  else if (logits_dtype == MSGL_BF16) MSGL_ARGMAX(MSGL_BF16);
  else if (logits_dtype == MSGL_FP16) MSGL_ARGMAX(MSGL_FP16);
  else {
    set_error("argmax_rows: unsupported dtype code %d", logits_dtype);
    return MSGL_EINVAL;
  }
#undef MSGL_ARGMAX
  MSGL_CHECK_LAUNCH("argmax_rows");
  return MSGL_OK;
}

extern "C" int msgl_softmax_temperature(float* probs, const void* logits, const float* temperatures,
                                        int64_t rows, int64_t vocab, int64_t logits_stride,
                                        int64_t probs_stride, int logits_dtype, void* stream) {
  MSGL_REQUIRE(rows >= 0, "softmax_temperature: negative rows");
  if (rows == 0) return MSGL_OK;
  MSGL_REQUIRE(probs && logits && temperatures, "softmax_temperature: null pointer");
  MSGL_REQUIRE(vocab >= 1 && rows < (1ll << 31), "softmax_temperature: bad shape");
  const int esz = logits_dtype == MSGL_F32 ? 4 : 2;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 g((unsigned)rows), b(kRowThreads);
  if (logits_dtype == MSGL_F32)
    softmax_kernel<MSGL_F32><<<g, b, 0, s>>>(probs, (const char*)logits, temperatures, vocab, logits_stride * esz, probs_stride);
  else if (logits_dtype == MSGL_BF16)
    softmax_kernel<MSGL_BF16><<<g, b, 0, s>>>(probs, (const char*)logits, temperatures, vocab, logits_stride * esz, probs_stride);
  else if (logits_dtype == MSGL_FP16)
    softmax_kernel<MSGL_FP16><<<g, b, 0, s>>>(probs, (const char*)logits, temperatures, vocab, logits_stride * esz, probs_stride);
  else {
    set_error("softmax_temperature: unsupported dtype code %d", logits_dtype);
    return MSGL_EINVAL;
  }
  MSGL_CHECK_LAUNCH("softmax_temperature");
  return MSGL_OK;
}

extern "C" int msgl_sample_top_k_top_p(int32_t* out, const float* probs, const int32_t* top_k,
                                       const float* top_p, int64_t rows, int64_t vocab,
                                       int64_t probs_stride, uint64_t seed, uint64_t offset, void* stream) {
  MSGL_REQUIRE(rows >= 0, "sample_top_k_top_p: negative rows");
  if (rows == 0) return MSGL_OK;
  MSGL_REQUIRE(out && probs, "sample_top_k_top_p: null pointer");
  MSGL_REQUIRE(vocab >= 1 && vocab < (1ll << 31) && rows < (1ll << 31), "sample_top_k_top_p: bad shape");
  sample_kernel<<<dim3((unsigned)rows), dim3(kRowThreads), 0, static_cast<hipStream_t>(stream)>>>(
      out, probs, top_k, top_p, vocab, probs_stride, (u64)seed, (u64)offset);
  MSGL_CHECK_LAUNCH("sample_top_k_top_p");
  return MSGL_OK;
}

extern "C" int msgl_sample_from_logits(int32_t* out, const void* logits, const float* temperatures, int64_t rows,
                                       int64_t vocab, int64_t logits_stride, int logits_dtype, uint64_t seed,
                                       uint64_t offset, void* stream) {
  MSGL_REQUIRE(rows >= 0, "sample_from_logits: negative rows");
  if (rows == 0) return MSGL_OK;
  MSGL_REQUIRE(out && logits && temperatures, "sample_from_logits: null pointer");
  MSGL_REQUIRE(vocab >= 1 && vocab < (1ll << 31) && rows < (1ll << 31), "sample_from_logits: bad shape");
  const int esz = logits_dtype == MSGL_F32 ? 4 : 2;
  MSGL_REQUIRE(aligned16(logits) && (logits_stride * esz) % 16 == 0,
               "sample_from_logits: rows must be 16-byte aligned (stride %lld elements)", (long long)logits_stride);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 g((unsigned)rows), b(kRowThreads);
  if (logits_dtype == MSGL_F32)
    sample_logits_kernel<MSGL_F32><<<g, b, 0, s>>>(out, (const char*)logits, temperatures, vocab, logits_stride * esz, (u64)seed, (u64)offset);
  else if (logits_dtype == MSGL_BF16)
    sample_logits_kernel<MSGL_BF16><<<g, b, 0, s>>>(out, (const char*)logits, temperatures, vocab, logits_stride * esz, (u64)seed, (u64)offset);
  else if (logits_dtype == MSGL_FP16)
    sample_logits_kernel<MSGL_FP16><<<g, b, 0, s>>>(out, (const char*)logits, temperatures, vocab, logits_stride * esz, (u64)seed, (u64)offset);
  else {
    set_error("sample_from_logits: unsupported dtype code %d", logits_dtype);
    return MSGL_EINVAL;
  }
  MSGL_CHECK_LAUNCH("sample_from_logits");
  return MSGL_OK;
}
