// Paged varlen causal prefill attention for gfx950 (MFMA 32x32x16, fp32 accumulate).
//
// One workgroup = 4 waves = one 128-row query tile of one (request, q head); each wave owns 32
// query rows.  K/V are gathered through the reference's token-granular page table in 64-key
// tiles, staged global -> registers -> LDS one tile ahead of the MFMAs (loads issued before the
// tile's compute, LDS written after it).
//
// Data flow per 64-key tile and wave (all in registers except the K/V tile):
//   S^T = K . Q^T   ("swapped" QK^T: A = K rows from LDS (XOR-swizzled, conflict-free
//                    ds_read_b128), B = the wave's Q fragments held in VGPRs)
//        => lane l owns query row (l & 31): its 32 scores are in-lane (+ lane l^32), so the
//           online softmax needs one cross-lane exchange per row statistic, no LDS;
//   O^T += V^T . P^T (A = V^T fragments read from a transposed, swizzled LDS image of V with
//                    two ds_read_b64; B = P packed to 16-bit straight from the score registers:
//                    the contraction index is permuted so each lane uses its OWN 8 keys -- no
//                    cross-lane movement of P)
//        => every O accumulator of lane l belongs to query row (l & 31): rescale and the final
//           1/l are lane-local.
// Causal mask is bottom-right aligned (query j of q_len sees keys t <= k_len - q_len + j),
// applied only on tiles that cross the diagonal; tiles fully above it are skipped.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace msgl {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int kQTile = MSGL_PREFILL_QTILE;  // 128 query rows per workgroup
constexpr int kKTile = 64;                  // keys per tile
constexpr int kD = 128;
constexpr float kNegBigP = -3.0e38f;

template <typename T>
__device__ __forceinline__ f32x16 mfma32(const U4& a, const U4& b, f32x16 c) {
  if constexpr (std::is_same_v<T, BF16>) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                   c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c,
                                                  0, 0, 0);
  }
}

struct PrefillParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  const int* page_table;
  const int* req_rows;
  const int* seq_lens;
  const int* cu_q;
  const int* tile_cu;
  const int* tile_order;  // optional [total_tiles]: launch order of the q tiles (heaviest first); tr kernel only
  uint16_t* out;
  int64_t pt_stride, q_stride, kv_stride_tok, kv_stride_head, out_stride;
  int batch, hq, group;
  float scale_log2;
};

// swizzles (see header comment): K image [64 keys][256 B], V^T image [128 d][128 B = 64 keys]
__device__ __forceinline__ int k_off(int key, int byte_in_row) { return key * 256 + (byte_in_row ^ ((key & 15) << 4)); }
__device__ __forceinline__ int vt_g(int d) { return ((d >> 1) & 15) ^ ((d >> 5) & 3); }
__device__ __forceinline__ int vt_off(int d, int unit) { return d * 128 + ((unit ^ vt_g(d)) << 3); }

template <typename T>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const PrefillParams p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * kKTile * kD * 2];
  char* lds_k = lds;
  char* lds_v = lds + kKTile * kD * 2;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hi = lane >> 5;
  const int hq = blockIdx.y;
  const int kvh = hq / p.group;

  // ---- which (request, q tile) is this workgroup ---------------------------------------
  const int tile = blockIdx.x;
  int lo = 0, hi_b = p.batch;  // find b with tile_cu[b] <= tile < tile_cu[b+1]
  while (hi_b - lo > 1) {
    const int mid = (lo + hi_b) >> 1;
    if (p.tile_cu[mid] <= tile) lo = mid; else hi_b = mid;
  }
  const int b = lo;
  const int q_begin = p.cu_q[b];
  const int q_len = p.cu_q[b + 1] - q_begin;
  const int k_len = p.seq_lens[b];
  const int q0 = (tile - p.tile_cu[b]) * kQTile;
  const int row = p.req_rows ? p.req_rows[b] : b;
  const int* pt = p.page_table + (int64_t)row * p.pt_stride;
  const int diag = k_len - q_len;  // query j sees keys t <= diag + j
  // keys needed by this tile: [0, kend)
  const int kend = min(k_len, diag + min(q0 + kQTile, q_len));
  const int ntiles = (kend + kKTile - 1) / kKTile;

  // ---- Q fragments: lane owns query row (l & 31) of its wave ---------------------------
  const int my_q = q0 + wave * 32 + (lane & 31);  // row inside the request
  const bool q_valid = my_q < q_len;
  const int64_t q_tok = q_begin + (q_valid ? my_q : q_len - 1);
  U4 qf[8];
  {
    const uint16_t* qp = p.q + q_tok * p.q_stride + (int64_t)hq * kD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = ldg16(qp + ks * 16);
  }
  const int my_qpos = diag + my_q;                  // last visible key of my row
  const int wave_min_qpos = diag + q0 + wave * 32;  // smallest in the wave (row 0)

  f32x16 o[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
  float m_run = kNegBigP, l_run = 0.f;

  // ---- K/V tile staging: thread owns 16-byte pieces c = tid + 256 i, i < 4 ---------------
  U4 kreg[4], vreg[4];
  auto issue_loads = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      const int key = c >> 4, piece = c & 15;
      const int tok = min(kt * kKTile + key, k_len - 1);  // never dereference past the sequence
      const int64_t off = (int64_t)pt[tok] * p.kv_stride_tok + (int64_t)kvh * p.kv_stride_head + piece * 8;
      kreg[i] = ldg16(p.k + off);
      vreg[i] = ldg16(p.v + off);
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      const int key = c >> 4, piece = c & 15;
      *reinterpret_cast<U4*>(lds_k + k_off(key, piece * 16)) = kreg[i];
      // V transposed: element (key, d) -> V^T[d][key]
      const uint32_t w[4] = {vreg[i].x, vreg[i].y, vreg[i].z, vreg[i].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = piece * 8 + e;
        const uint16_t val = (uint16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
        *reinterpret_cast<uint16_t*>(lds_v + vt_off(d, key >> 2) + (key & 3) * 2) = val;
      }
    }
  };

  if (ntiles > 0) issue_loads(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();  // previous tile fully consumed
    write_lds();
    __syncthreads();
    if (kt + 1 < ntiles) issue_loads(kt + 1);  // in flight during this tile's MFMAs

    const int key0 = kt * kKTile;
    if (key0 > diag + q0 + wave * 32 + 31) continue;  // whole tile above this wave's diagonal

    // ---- S^T = K . Q^T -----------------------------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
      const int key = kb * 32 + (lane & 31);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const U4 a = *reinterpret_cast<const U4*>(lds_k + k_off(key, ks * 32 + hi * 16));
        s[kb] = mfma32<T>(a, qf[ks], s[kb]);
      }
    }
    // ---- scale, mask, online softmax (lane-local row) ------------------------------------
    const bool need_mask = key0 + kKTile - 1 > wave_min_qpos;
    float tmax = kNegBigP;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float x = s[kb][r] * p.scale_log2;
        if (need_mask) {
          const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key > my_qpos) x = -INFINITY;
        }
        s[kb][r] = x;
        tmax = fmaxf(tmax, x);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float rsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
        s[kb][r] = e;
        rsum += e;
      }
    }
    rsum += __shfl_xor(rsum, 32, 64);
    l_run = fmaf(l_run, alpha, rsum);
    m_run = m_new;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[nb][r] *= alpha;

    // ---- O^T += V^T . P^T ------------------------------------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        U4 pf;
        pf.x = Elem<T>::pack(s[kb][8 * half + 0], s[kb][8 * half + 1]);
        pf.y = Elem<T>::pack(s[kb][8 * half + 2], s[kb][8 * half + 3]);
        pf.z = Elem<T>::pack(s[kb][8 * half + 4], s[kb][8 * half + 5]);
        pf.w = Elem<T>::pack(s[kb][8 * half + 6], s[kb][8 * half + 7]);
        const int u1 = kb * 8 + half * 4 + hi;  // 4-key unit of slots j = 0..3; j = 4..7 is unit u1 + 2
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int d = nb * 32 + (lane & 31);
          const uint2 v1 = *reinterpret_cast<const uint2*>(lds_v + vt_off(d, u1));
          const uint2 v2 = *reinterpret_cast<const uint2*>(lds_v + vt_off(d, u1 + 2));
          U4 vf;
          vf.x = v1.x; vf.y = v1.y; vf.z = v2.x; vf.w = v2.y;
          o[nb] = mfma32<T>(vf, pf, o[nb]);
        }
      }
    }
  }

  // ---- epilogue: O[q row][d] = O^T / l ---------------------------------------------------------
  if (q_valid) {
    const float inv = 1.0f / l_run;
    uint16_t* op = p.out + (int64_t)(q_begin + my_q) * p.out_stride + (int64_t)hq * kD;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = nb * 32 + 8 * rg + 4 * hi;
        uint2 w;
        w.x = Elem<T>::pack(o[nb][4 * rg + 0] * inv, o[nb][4 * rg + 1] * inv);
        w.y = Elem<T>::pack(o[nb][4 * rg + 2] * inv, o[nb][4 * rg + 3] * inv);
        *reinterpret_cast<uint2*>(op + d) = w;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Second-generation kernel (impl 2, the default): same math and fragment ownership as above, but
//   * V is staged ROW-major ([64 keys][256 B], like K: four ds_write_b128 per thread and tile instead of
//     thirty-two 2-byte transposing stores) and the V^T MFMA fragments come out of LDS through the gfx950
//     transposing read ds_read_b64_tr_b16: a 16-lane group reads one [4 keys][16 d] block (lane j passes the
//     address of row j>>2, columns 4(j&3)..+3) and lane j receives column j of it = its 4 consecutive keys.
//     64-B XOR swizzle on the key's low two bits => the 32 lanes of a read hit 64 distinct banks;
//   * K/V tiles are double-buffered in LDS (64 KB per workgroup): ONE barrier per 64-key tile; loads for tile
//     t+1 are issued before tile t's MFMAs and written to LDS after them; page-table slots run one more
//     tile ahead so the gather never waits for its own indices;
//   * <= 256 registers: two workgroups (8 waves) per CU, so one workgroup's softmax overlaps the other's MFMAs;
//   * 1-D grid remapped so that each XCD owns a contiguous range of (kv head, q tile, q head of the group):
//     the G query heads that share a K/V tile run back to back on ONE XCD's L2 (at Hkv = 8: kv head == XCD);
//     q tiles are taken in the caller's order (heaviest first) so the tail is made of light tiles;
//   * the O rescale is skipped (exactly) on tiles where no row's running max moved.
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ int v_off(int key, int byte_in_row) { return key * 256 + (byte_in_row ^ ((key & 3) << 6)); }

__device__ __forceinline__ uint2 tr_read_b64(const char* lds_ptr) {
  const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_char*)lds_ptr);
  return __builtin_bit_cast(uint2, r);
}

constexpr int kTileBytes = kKTile * kD * 2;  // 16 KB: one K (or V) tile image

template <typename T, bool kFold = true>
__global__ __launch_bounds__(256, 2) void attn_prefill_tr_kernel(const PrefillParams p, int total_tiles) {
  __shared__ __attribute__((aligned(16))) char lds[4 * kTileBytes];  // [buf][K | V]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hi = lane >> 5;

  // ---- block -> (kv head, q tile, head of the group): XCD-contiguous virtual index ------------------
  const int n_per = (int)(gridDim.x >> 3);
  const int vidx = (int)(blockIdx.x & 7) * n_per + (int)(blockIdx.x >> 3);
  if (vidx >= total_tiles * p.hq) return;
  const int per_kv = total_tiles * p.group;
  const int kvh = vidx / per_kv;
  const int rem = vidx - kvh * per_kv;
  const int ti = rem / p.group;
  const int hq = kvh * p.group + (rem - ti * p.group);
  const int tile = p.tile_order ? p.tile_order[ti] : ti;

  int lo = 0, hi_b = p.batch;  // find b with tile_cu[b] <= tile < tile_cu[b+1]
  while (hi_b - lo > 1) {
    const int mid = (lo + hi_b) >> 1;
    if (p.tile_cu[mid] <= tile) lo = mid; else hi_b = mid;
  }
  const int b = lo;
  const int q_begin = p.cu_q[b];
  const int q_len = p.cu_q[b + 1] - q_begin;
  const int k_len = p.seq_lens[b];
  const int q0 = (tile - p.tile_cu[b]) * kQTile;
  const int row = p.req_rows ? p.req_rows[b] : b;
  const int* pt = p.page_table + (int64_t)row * p.pt_stride;
  const int diag = k_len - q_len;
  const int kend = min(k_len, diag + min(q0 + kQTile, q_len));
  const int ntiles = (kend + kKTile - 1) / kKTile;

  // ---- Q fragments ------------------------------------------------------------------------------
  const int my_q = q0 + wave * 32 + (lane & 31);
  const bool q_valid = my_q < q_len;
  const int64_t q_tok = q_begin + (q_valid ? my_q : q_len - 1);
  U4 qf[8];
  {
    const uint16_t* qp = p.q + q_tok * p.q_stride + (int64_t)hq * kD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = ldg16(qp + ks * 16);
  }
  const int my_qpos = diag + my_q;
  const int wave_min_qpos = diag + q0 + wave * 32;

  f32x16 o[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
  float m_run = kNegBigP, l_run = 0.f;

  // ---- staging: thread owns 16-byte pieces c = tid + 256 i (key = c >> 4, piece = c & 15) -----------
  const int st_key = tid >> 4, st_piece = tid & 15;  // key of piece i = st_key + 16 i
  const int64_t head_off = (int64_t)kvh * p.kv_stride_head + st_piece * 8;
  int sl[4];
  U4 kreg[4], vreg[4];
  auto load_slots = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sl[i] = pt[min(kt * kKTile + st_key + 16 * i, k_len - 1)];  // never past the sequence
  };
  auto issue_loads = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t off = (int64_t)sl[i] * p.kv_stride_tok + head_off;
      kreg[i] = ldg16(p.k + off);
      vreg[i] = ldg16(p.v + off);
    }
  };
  auto write_lds = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = st_key + 16 * i;
      *reinterpret_cast<U4*>(buf + k_off(key, st_piece * 16)) = kreg[i];
      *reinterpret_cast<U4*>(buf + kTileBytes + v_off(key, st_piece * 16)) = vreg[i];
    }
  };
  // V^T fragment addressing: lane = 16 g4 + j; the lane's share of a [4 keys][16 d] block is row j >> 2,
  // columns 16 (g4 & 1) + 4 (j & 3) .. +3 of d block nb; the swizzle term depends on the lane's row only
  int voff[4];
  {
    const int j = lane & 15;
    const int r4 = j >> 2;  // lane j of a 16-lane tr-read group passes row j >> 2, 8-byte chunk j & 3
    const int ch = j & 3;   // (layout confirmed on hardware by tools/tr_probe.hip, pattern 1)
    const int w = (16 * ((lane >> 4) & 1) + 4 * ch) * 2;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) voff[nb] = kTileBytes + hi * 1024 + r4 * 256 + (((nb ^ r4) & 3) << 6) + w;
  }

  if (ntiles > 0) {
    load_slots(0);
    issue_loads();
    if (ntiles > 1) load_slots(1);
  }
  for (int kt = 0; kt < ntiles; ++kt) {
    char* buf = lds + (kt & 1) * (2 * kTileBytes);
    write_lds(buf);  // tile kt (its loads were issued one tile ago); the buffer was last read at tile kt-2
    if (kt + 1 < ntiles) {
      issue_loads();  // tile kt+1: in flight during this tile's MFMAs
      if (kt + 2 < ntiles) load_slots(kt + 2);
    }
    __syncthreads();

    const int key0 = kt * kKTile;
    if (key0 > diag + q0 + wave * 32 + 31) continue;  // whole tile above this wave's diagonal

    // ---- S^T = K . Q^T -----------------------------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
      const int key = kb * 32 + (lane & 31);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const U4 a = *reinterpret_cast<const U4*>(buf + k_off(key, ks * 32 + hi * 16));
        s[kb] = mfma32<T>(a, qf[ks], s[kb]);
      }
    }
    // ---- mask, online softmax (lane-local row); kFold: the softmax scale folded into the exponent's fma
    float tmax = kNegBigP;
    float m_new, alpha, rsum = 0.f;
    const bool need_mask = key0 + kKTile - 1 > wave_min_qpos;
    if constexpr (!kFold) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float x = s[kb][r] * p.scale_log2;
          if (need_mask) {
            const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key > my_qpos) x = -INFINITY;
          }
          s[kb][r] = x;
          tmax = fmaxf(tmax, x);
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      m_new = fmaxf(m_run, tmax);
      alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
          s[kb][r] = e;
          rsum += e;
        }
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float x = s[kb][r];
          if (need_mask) {
            const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key > my_qpos) x = -INFINITY;
            s[kb][r] = x;
          }
          tmax = fmaxf(tmax, x);
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * p.scale_log2;  // scale > 0: max commutes with it
      m_new = fmaxf(m_run, tmax);
      alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(fmaf(s[kb][r], p.scale_log2, -m_new));
          s[kb][r] = e;
          rsum += e;
        }
      }
    }
    rsum += __shfl_xor(rsum, 32, 64);
    l_run = fmaf(l_run, alpha, rsum);
    if (!__all(m_new == m_run)) {  // alpha == 1 on every row otherwise: the rescale would be the identity
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] *= alpha;
    }
    m_run = m_new;

    // ---- O^T += V^T . P^T ------------------------------------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        U4 pf;
        pf.x = Elem<T>::pack(s[kb][8 * half + 0], s[kb][8 * half + 1]);
        pf.y = Elem<T>::pack(s[kb][8 * half + 2], s[kb][8 * half + 3]);
        pf.z = Elem<T>::pack(s[kb][8 * half + 4], s[kb][8 * half + 5]);
        pf.w = Elem<T>::pack(s[kb][8 * half + 6], s[kb][8 * half + 7]);
        // lane's keys: 4-key unit u1 = kb 8 + half 4 + hi (k slots 0..3) and unit u1 + 2 (k slots 4..7);
        // unit u = rows 4u..4u+3 = 1 KB of the image (hi is folded into voff)
        constexpr int kUnitBytes = 4 * 256;
        const int ub = (kb * 8 + half * 4) * kUnitBytes;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const uint2 v1 = tr_read_b64(buf + voff[nb] + ub);
          const uint2 v2 = tr_read_b64(buf + voff[nb] + ub + 2 * kUnitBytes);
          U4 vf;
          vf.x = v1.x; vf.y = v1.y; vf.z = v2.x; vf.w = v2.y;
          o[nb] = mfma32<T>(vf, pf, o[nb]);
        }
      }
    }
  }

  // ---- epilogue: O[q row][d] = O^T / l ---------------------------------------------------------
  if (q_valid) {
    const float inv = 1.0f / l_run;
    uint16_t* op = p.out + (int64_t)(q_begin + my_q) * p.out_stride + (int64_t)hq * kD;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = nb * 32 + 8 * rg + 4 * hi;
        uint2 w;
        w.x = Elem<T>::pack(o[nb][4 * rg + 0] * inv, o[nb][4 * rg + 1] * inv);
        w.y = Elem<T>::pack(o[nb][4 * rg + 2] * inv, o[nb][4 * rg + 3] * inv);
        *reinterpret_cast<uint2*>(op + d) = w;
      }
    }
  }
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_attn_prefill(void* out, const void* q, const void* k_cache, const void* v_cache,
                                 const int32_t* page_table, int64_t pt_stride, const int32_t* req_rows,
                                 const int32_t* seq_lens, const int32_t* cu_seqlens_q, const int32_t* tile_cu,
                                 int batch, int total_tiles, int num_q_heads, int num_kv_heads, int head_dim,
                                 int64_t q_stride_tok, int64_t kv_stride_tok, int64_t kv_stride_head,
                                 int64_t out_stride_tok, float sm_scale, int dtype, const int32_t* tile_order,
                                 int impl, void* stream) {
  MSGL_REQUIRE(batch >= 0 && total_tiles >= 0, "attn_prefill: negative sizes");
  MSGL_REQUIRE(impl >= 0 && impl <= 3, "attn_prefill: impl %d (0 default, 1 register-transposed V, 2 tr-read, 3 tr-read unfolded scale)", impl);
  if (batch == 0 || total_tiles == 0) return MSGL_OK;
  MSGL_REQUIRE(out && q && k_cache && v_cache && page_table && seq_lens && cu_seqlens_q && tile_cu,
               "attn_prefill: null pointer");
  MSGL_REQUIRE(head_dim == 128, "attn_prefill: head_dim %d unsupported (128 only)", head_dim);
  MSGL_REQUIRE(num_kv_heads >= 1 && num_q_heads % num_kv_heads == 0 && num_q_heads <= 65535,
               "attn_prefill: %d q heads / %d kv heads", num_q_heads, num_kv_heads);
  MSGL_REQUIRE(q_stride_tok % 8 == 0 && kv_stride_tok % 8 == 0 && kv_stride_head % 8 == 0 &&
                   out_stride_tok % 4 == 0,
               "attn_prefill: strides must be multiples of 8 elements");
  MSGL_REQUIRE(aligned16(q) && aligned16(k_cache) && aligned16(v_cache) &&
                   (reinterpret_cast<uintptr_t>(out) & 7u) == 0,
               "attn_prefill: pointers must be 16-byte aligned");
  PrefillParams p;
  p.q = (const uint16_t*)q;
  p.k = (const uint16_t*)k_cache;
  p.v = (const uint16_t*)v_cache;
  p.page_table = page_table;
  p.req_rows = req_rows;
  p.seq_lens = seq_lens;
  p.cu_q = cu_seqlens_q;
  p.tile_cu = tile_cu;
  p.tile_order = tile_order;
  p.out = (uint16_t*)out;
  p.pt_stride = pt_stride;
  p.q_stride = q_stride_tok;
  p.kv_stride_tok = kv_stride_tok;
  p.kv_stride_head = kv_stride_head;
  p.out_stride = out_stride_tok;
  p.batch = batch;
  p.hq = num_q_heads;
  p.group = num_q_heads / num_kv_heads;
  p.scale_log2 = sm_scale * 1.4426950408889634f;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype != MSGL_BF16 && dtype != MSGL_FP16) {
    set_error("attn_prefill: unsupported dtype code %d", dtype);
    return MSGL_EINVAL;
  }
  if (impl == 0) impl = 2;  // the tr-read kernel; impl 1 keeps the first-generation kernel callable (cross-check in tests)
  if (impl == 1) {  // first-generation kernel: 2-D grid in natural order (tile_order unused)
    const dim3 grid((unsigned)total_tiles, (unsigned)num_q_heads), block(256);
    if (dtype == MSGL_BF16) attn_prefill_kernel<BF16><<<grid, block, 0, s>>>(p);
    else attn_prefill_kernel<FP16><<<grid, block, 0, s>>>(p);
  } else {
    const int64_t total = (int64_t)total_tiles * num_q_heads;
    MSGL_REQUIRE(total < (1ll << 30), "attn_prefill: %lld workgroups", (long long)total);
    const unsigned blocks = (unsigned)((total + 7) / 8) * 8;  // 8 XCDs x n_per
    if (impl == 3) {
      if (dtype == MSGL_BF16) attn_prefill_tr_kernel<BF16, false><<<dim3(blocks), dim3(256), 0, s>>>(p, total_tiles);
      else attn_prefill_tr_kernel<FP16, false><<<dim3(blocks), dim3(256), 0, s>>>(p, total_tiles);
    } else if (dtype == MSGL_BF16) attn_prefill_tr_kernel<BF16><<<dim3(blocks), dim3(256), 0, s>>>(p, total_tiles);
    else attn_prefill_tr_kernel<FP16><<<dim3(blocks), dim3(256), 0, s>>>(p, total_tiles);
  }
  MSGL_CHECK_LAUNCH("attn_prefill");
  return MSGL_OK;
}
