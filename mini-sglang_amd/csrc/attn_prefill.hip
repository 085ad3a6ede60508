// Paged varlen causal prefill attention for gfx950 (MFMA 32x32x16, fp32 accumulate).
//
// Two kernels with the same math per query row, bit-identical to each other (impl codes in include/msgl_hip.h): the
// default DMA-staged one (attn_prefill_dma_kernel, impl 4 = 0) and its register-staged predecessor
// (attn_prefill_tr_kernel, impl 2), kept as the A/B partner and cross-check.  One workgroup = 4 waves = one 128-row
// query tile of one (request, q head); each wave owns 32 query rows.  K/V are gathered through the reference's
// token-granular page table in 64-key tiles one tile ahead of the MFMAs: global -> registers -> LDS (impl 2), global ->
// LDS by DMA (global_load_lds_dwordx4; impl 4).  Rounds 1-3 also carried a first-generation kernel (V transposed by
// 2-byte LDS stores) and a counter-phase kernel (two wave groups of a 256-row tile, matrix segment beside softmax
// segment: bit-identical, 43 % VALU/MFMA co-execution, NOT faster -- DESIGN.md section 8); both were removed in round 4,
// their measurements stay under profiles/r03_prefill_*.  Ablation variants (ABL != 0: timing only, WRONG results) are
// compiled only with -DMSGL_PREFILL_DIAG (MSGL_PREFILL_DIAG=1 python mini-sglang_amd/build.py): a stray impl code
// cannot select one in a production build.
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

// element offset of a pool slot: slot and the token stride are both < 2^32 (checked at launch), so ONE v_mad_u64_u32
// does it; the int64 x int64 product the types ask for costs two quarter-rate 32-bit multiplies and a carry chain more
__device__ __forceinline__ int64_t slot_offset(int slot, int64_t stride_tok) {
  return (int64_t)((uint64_t)(uint32_t)slot * (uint64_t)(uint32_t)stride_tok);
}

// The request a q tile belongs to and its scalars, in ONE memory latency for batches up to 64 requests (the bisection over
// tile_cu followed by the four scalar loads paid log2(B) + 1 dependent ones: ~2.5 us of a workgroup that computes for
// ~5 us on the benchmark's prefill chunks).  Lane i of a 64-request block loads request i's tile_cu, cu_q, cu_q[+1],
// seq_len and table row together; b = the last request with tile_cu[b] <= tile is counted with a ballot and its values are
// taken from that lane.
struct TileRequest {
  int b, q_begin, q_end, k_len, row, tile_first;
};
__device__ __forceinline__ TileRequest find_request(const PrefillParams& p, int tile, int lane) {
  TileRequest r;
  int base = 0;
  for (;;) {
    const int idx = min(base + lane, p.batch - 1);
    const int tc = base + lane < p.batch ? p.tile_cu[idx] : 0x7fffffff;
    const int cq0 = p.cu_q[idx], cq1 = p.cu_q[idx + 1], kl = p.seq_lens[idx];
    const int rr = p.req_rows ? p.req_rows[idx] : idx;
    const int cnt = __popcll(__ballot(tc <= tile));
    if (cnt > 0 && (cnt < 64 || base + 64 >= p.batch)) {
      const int l = cnt - 1;
      r.b = base + l;
      r.tile_first = __builtin_amdgcn_readlane(tc, l);
      r.q_begin = __builtin_amdgcn_readlane(cq0, l);
      r.q_end = __builtin_amdgcn_readlane(cq1, l);
      r.k_len = __builtin_amdgcn_readlane(kl, l);
      r.row = __builtin_amdgcn_readlane(rr, l);
      return r;
    }
    if (cnt == 0) {  // the last request of the previous block (more than 64 requests only)
      r.b = base - 1;
      r.tile_first = p.tile_cu[r.b];
      r.q_begin = p.cu_q[r.b];
      r.q_end = p.cu_q[r.b + 1];
      r.k_len = p.seq_lens[r.b];
      r.row = p.req_rows ? p.req_rows[r.b] : r.b;
      return r;
    }
    base += 64;
  }
}

// swizzles (see header comment): K image [64 keys][256 B], V^T image [128 d][128 B = 64 keys]
__device__ __forceinline__ int k_off(int key, int byte_in_row) { return key * 256 + (byte_in_row ^ ((key & 15) << 4)); }
__device__ __forceinline__ int vt_g(int d) { return ((d >> 1) & 15) ^ ((d >> 5) & 3); }
__device__ __forceinline__ int vt_off(int d, int unit) { return d * 128 + ((unit ^ vt_g(d)) << 3); }

// ------------------------------------------------------------------------------------------------
// Second-generation kernel (impl 2; the default until the DMA-staged impl 4 below): same math and fragment ownership as above, but
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

// ABL (diagnosis only, wrong results): 1 = K/V tiles loaded once (no HBM/L2 stream, no LDS writes after the first tile),
// 2 = no QK^T MFMAs, 4 = no softmax arithmetic, 8 = no PV MFMAs, 16 = no barrier after the first tile
template <typename T, bool kFold = true, int ABL = 0>
__global__ __launch_bounds__(256, 2) void attn_prefill_tr_kernel(const PrefillParams p, int total_tiles) {
  __shared__ __attribute__((aligned(16))) char lds[4 * kTileBytes];  // [buf][K | V]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: the diagonal tests below are scalar branches
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
      const int64_t off = slot_offset(sl[i], p.kv_stride_tok) + head_off;
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
    char* buf = lds + ((ABL & 1) ? 0 : (kt & 1)) * (2 * kTileBytes);
    if (!(ABL & 1) || kt == 0) write_lds(buf);  // tile kt (its loads were issued one tile ago); the buffer was last read at tile kt-2
    if (kt + 1 < ntiles && !(ABL & 1)) {
      issue_loads();  // tile kt+1: in flight during this tile's MFMAs
      if (kt + 2 < ntiles) load_slots(kt + 2);
    }
    if (!(ABL & 16) || kt == 0) __syncthreads();

    const int key0 = kt * kKTile;
    if (key0 > diag + q0 + wave * 32 + 31) continue;  // whole tile above this wave's diagonal

    // ---- S^T = K . Q^T -----------------------------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
      const int key = kb * 32 + (lane & 31);
      if (!(ABL & 2)) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const U4 a = *reinterpret_cast<const U4*>(buf + k_off(key, ks * 32 + hi * 16));
          s[kb] = mfma32<T>(a, qf[ks], s[kb]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = __uint_as_float(qf[r & 7].x) * 1e-30f + (float)kt;
      }
    }
    // ---- mask, online softmax (lane-local row); kFold: the softmax scale folded into the exponent's fma
    if constexpr (ABL & 4) {  // diagnosis: no softmax arithmetic (p = s as it is)
      m_run = 0.f;
      l_run += s[0][0] + s[1][15];
    } else {
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
    }

    if constexpr (!(ABL & 8)) {
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
    } else {
      o[0][0] += s[0][3] + s[1][7];
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
// Third-generation kernel (impl 4, the default): the math, fragment ownership and LDS images of the tr-read kernel, with
//   * K/V tiles moved global -> LDS by the DMA path (global_load_lds_dwordx4): no staging registers, no ds_write, no
//     LDS-write bandwidth.  A wave instruction fills 1 KB = 4 key rows lane-linearly, so the XOR swizzles of the images
//     are applied on the SOURCE side: the lane that lands on 16-byte slot c of row r fetches piece c ^ swz(r);
//   * the tile loop unrolled by two: every LDS address is (a per-lane register computed once) + an immediate;
//   * the causal mask hoisted out of the per-element path (one uniform branch per tile; compare against inline constants);
//   * one barrier per tile as before: wait own DMA of tile t, barrier, issue the DMA of tile t+1 into the buffer
//     everyone has just finished reading, compute tile t;
//   * LDS reads issued >= 4 MFMAs ahead of their use (pinned with sched_barrier).
// ABL (diagnosis, wrong results): 1 = tiles DMA'd once, 2 = no QK^T, 4 = no softmax, 8 = no PV.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

template <typename T, int ABL = 0>
__global__ __launch_bounds__(256, 2) void attn_prefill_dma_kernel(const PrefillParams p, int total_tiles) {
  // two distinct LDS objects, not one array: the compiler then knows a tile's fragment reads cannot alias the DMA writes
  // of the next tile (other object) and does not wait for them (vmcnt) before the reads
  __shared__ __attribute__((aligned(1024))) char lds_a[2 * kTileBytes];  // tiles 0, 2, ..: [K | V]
  __shared__ __attribute__((aligned(1024))) char lds_b[2 * kTileBytes];  // tiles 1, 3, ..

  // ABL bit 16: MFMA blocks at s_setprio 1, bit 32: the softmax at priority 1 instead.  Neither moves the kernel outside
  // run-to-run noise once the variants are timed interleaved (tools/prefill_ablate.py), so the default sets no priority
  constexpr bool kPrioMatrix = (ABL & 16) != 0;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;

  const int n_per = (int)(gridDim.x >> 3);
  const int vidx = (int)(blockIdx.x & 7) * n_per + (int)(blockIdx.x >> 3);
  if (vidx >= total_tiles * p.hq) return;
  const int per_kv = total_tiles * p.group;
  const int kvh = vidx / per_kv;
  const int rem = vidx - kvh * per_kv;
  const int ti = rem / p.group;
  const int hq = kvh * p.group + (rem - ti * p.group);
  const int tile = p.tile_order ? p.tile_order[ti] : ti;

  const TileRequest rq = find_request(p, tile, lane);
  const int q_begin = rq.q_begin;
  const int q_len = rq.q_end - q_begin;
  const int k_len = rq.k_len;
  const int q0 = (tile - rq.tile_first) * kQTile;
  const int row = rq.row;
  const int* pt = p.page_table + (int64_t)row * p.pt_stride;
  const int diag = k_len - q_len;
  const int kend = min(k_len, diag + min(q0 + kQTile, q_len));
  const int ntiles = (kend + kKTile - 1) / kKTile;

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
  const int wave_min_qpos = diag + q0 + wave * 32;  // scalar

  f32x16 o[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
  float m_run = kNegBigP, l_run = 0.f;

  // ---- staging: DMA instruction i of this wave fills rows 16 i + 4 wave .. + 3 (chunk 4 i + wave of the image);
  //      lane -> row st_row + 16 i, 16-byte slot (lane & 15)
  const int st_row = tid >> 4;
  const int64_t k_lane = (int64_t)kvh * p.kv_stride_head + (((lane & 15) ^ (st_row & 15)) << 3);
  const int64_t v_lane = (int64_t)kvh * p.kv_stride_head + (((lane & 15) ^ ((st_row & 3) << 2)) << 3);
  int sl[4];
  auto load_slots = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sl[i] = pt[min(kt * kKTile + st_row + 16 * i, k_len - 1)];
  };
  auto issue_dma = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t off = slot_offset(sl[i], p.kv_stride_tok);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.k + off + k_lane), (lds_void_t*)(buf + (4 * i + wave) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.v + off + v_lane),
                                       (lds_void_t*)(buf + kTileBytes + (4 * i + wave) * 1024), 16, 0, 0);
    }
  };
  // per-lane LDS addresses (buffer 0): K fragment of k-step ks, V^T fragment of d block nb
  int koff[8], voff[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) koff[ks] = k_off(lane & 31, ks * 32 + hi * 16);
  {
    const int j = lane & 15;
    const int r4 = j >> 2;
    const int ch = j & 3;
    const int w = (16 * ((lane >> 4) & 1) + 4 * ch) * 2;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) voff[nb] = kTileBytes + hi * 1024 + r4 * 256 + (((nb ^ r4) & 3) << 6) + w;
  }

  auto tile_body = [&](auto BUFC, const int kt) {
    constexpr int BUF = decltype(BUFC)::value;
    char* buf = BUF ? lds_b : lds_a;
    char* nxt = BUF ? lds_a : lds_b;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt (and the slots of tile kt + 1/2)
    __syncthreads();                                  // everyone's pieces; everyone done with the other buffer
    if (!(ABL & 1) && kt + 1 < ntiles) {
      issue_dma(nxt);
      if (kt + 2 < ntiles) load_slots(kt + 2);
    }
    const int key0 = kt * kKTile;
    if (key0 > wave_min_qpos + 31) return;  // whole tile above this wave's diagonal

    // ---- S^T = K . Q^T.  Issue order is pinned (sched_barrier): the 8 fragments of key block 0 first, then one
    //      MFMA per further LDS read -- key block 1's fragments under block 0's chain, the V^T fragments of the first
    //      half of P.V (they do not depend on the softmax) under block 1's chain: no MFMA waits on a read issued just
    //      before it
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
    if constexpr (kPrioMatrix) __builtin_amdgcn_s_setprio(1);
    if constexpr (ABL & 32) __builtin_amdgcn_s_setprio(0);
    U4 vfa[8];  // V^T fragments of keys 0..31: [half][nb]
    constexpr int kUnitBytes = 4 * 256;
    auto read_vf = [&](int kb, int half, int nb) {
      const int ub = (kb * 8 + half * 4) * kUnitBytes;
      const uint2 v1 = tr_read_b64(buf + voff[nb] + ub);
      const uint2 v2 = tr_read_b64(buf + voff[nb] + ub + 2 * kUnitBytes);
      U4 vf;
      vf.x = v1.x; vf.y = v1.y; vf.z = v2.x; vf.w = v2.y;
      return vf;
    };
    if constexpr (!(ABL & 2)) {
      U4 kf0[8], kf1[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) kf0[ks] = *reinterpret_cast<const U4*>(buf + koff[ks]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {  // every read is issued >= 4 MFMAs (128 cycles) ahead of its use
        s[0] = mfma32<T>(kf0[ks], qf[ks], s[0]);
        if (ks < 4) {
          kf1[2 * ks] = *reinterpret_cast<const U4*>(buf + koff[2 * ks] + 8192);
          kf1[2 * ks + 1] = *reinterpret_cast<const U4*>(buf + koff[2 * ks + 1] + 8192);
        } else if constexpr (!(ABL & 8)) {
          vfa[ks - 4] = read_vf(0, 0, ks - 4);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        s[1] = mfma32<T>(kf1[ks], qf[ks], s[1]);
        if constexpr (!(ABL & 8)) {
          if (ks < 4) vfa[4 + ks] = read_vf(0, 1, ks);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = __uint_as_float(qf[r & 7].x) * 1e-30f + (float)kt;
      if constexpr (!(ABL & 8)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) vfa[j] = read_vf(0, j >> 2, j & 3);
      }
    }
    if constexpr (kPrioMatrix) __builtin_amdgcn_s_setprio(0);
    if constexpr (ABL & 32) __builtin_amdgcn_s_setprio(1);
    if constexpr (ABL & 4) {
      m_run = 0.f;
      l_run += s[0][0] + s[1][15];
    } else {
      // ---- causal mask (tiles crossing this wave's diagonal only): key0 + c + 4 hi > my_qpos  <=>  c > thr
      if (key0 + kKTile - 1 > wave_min_qpos) {
        const int thr = my_qpos - key0 - 4 * hi;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kb * 32 + (r & 3) + 8 * (r >> 2) > thr) s[kb][r] = -INFINITY;
      }
      // ---- online softmax, lane-local row (+ lane ^ 32); the scale is folded into the exponent's fma
      float tmax = kNegBigP;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[kb][r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * p.scale_log2;
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float rsum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(fmaf(s[kb][r], p.scale_log2, -m_new));
          s[kb][r] = e;
          rsum += e;
        }
      rsum += __shfl_xor(rsum, 32, 64);
      l_run = fmaf(l_run, alpha, rsum);
      if (!__all(m_new == m_run)) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[nb][r] *= alpha;
      }
      m_run = m_new;
    }
    // ---- O^T += V^T . P^T: keys 0..31 on the fragments read during QK^T, the fragments of keys 32..63 read under them
    if constexpr (!(ABL & 8)) {
      auto pack_p = [&](int kb, int half) {
        U4 pf;
        pf.x = Elem<T>::pack(s[kb][8 * half + 0], s[kb][8 * half + 1]);
        pf.y = Elem<T>::pack(s[kb][8 * half + 2], s[kb][8 * half + 3]);
        pf.z = Elem<T>::pack(s[kb][8 * half + 4], s[kb][8 * half + 5]);
        pf.w = Elem<T>::pack(s[kb][8 * half + 6], s[kb][8 * half + 7]);
        return pf;
      };
      U4 vfb[8];
      const U4 p00 = pack_p(0, 0), p01 = pack_p(0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kPrioMatrix) __builtin_amdgcn_s_setprio(1);
      if constexpr (ABL & 32) __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j & 3] = mfma32<T>(vfa[j], (j >> 2) ? p01 : p00, o[j & 3]);
        if (j < 4) {
          vfb[2 * j] = read_vf(1, (2 * j) >> 2, (2 * j) & 3);
          vfb[2 * j + 1] = read_vf(1, (2 * j + 1) >> 2, (2 * j + 1) & 3);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const U4 p10 = pack_p(1, 0), p11 = pack_p(1, 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j & 3] = mfma32<T>(vfb[j], (j >> 2) ? p11 : p10, o[j & 3]);
      if constexpr (kPrioMatrix) __builtin_amdgcn_s_setprio(0);
    } else {
      o[0][0] += s[0][3] + s[1][7];
    }
  };

  if (ntiles > 0) {
    load_slots(0);
    issue_dma(lds_a);
    if (ntiles > 1) load_slots(1);
  }
  for (int kt = 0; kt < ntiles; kt += 2) {
    tile_body(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < ntiles) tile_body(std::integral_constant<int, 1>{}, kt + 1);
  }

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

extern "C" int msgl_attn_prefill_q_tile(int /*impl*/) { return kQTile; }

extern "C" int msgl_attn_prefill(void* out, const void* q, const void* k_cache, const void* v_cache,
                                 const int32_t* page_table, int64_t pt_stride, const int32_t* req_rows,
                                 const int32_t* seq_lens, const int32_t* cu_seqlens_q, const int32_t* tile_cu,
                                 int batch, int total_tiles, int num_q_heads, int num_kv_heads, int head_dim,
                                 int64_t q_stride_tok, int64_t kv_stride_tok, int64_t kv_stride_head,
                                 int64_t out_stride_tok, float sm_scale, int dtype, const int32_t* tile_order,
                                 int impl, void* stream) {
  MSGL_REQUIRE(batch >= 0 && total_tiles >= 0, "attn_prefill: negative sizes");
#ifdef MSGL_PREFILL_DIAG
  MSGL_REQUIRE(impl == 0 || impl == 2 || impl == 4 || (impl >= 16 && impl < 48) || (impl >= 64 && impl < 128),
               "attn_prefill: impl %d (0 = 4 DMA-staged, 2 register-staged; diagnostic build: 16 + bits / 64 + bits = ablations)", impl);
#else
  MSGL_REQUIRE(impl == 0 || impl == 2 || impl == 4,
               "attn_prefill: impl %d (0 = 4 the DMA-staged kernel, 2 the register-staged kernel; ablation codes need a "
               "-DMSGL_PREFILL_DIAG build)", impl);
#endif
  if (batch == 0 || total_tiles == 0) return MSGL_OK;
  MSGL_REQUIRE(out && q && k_cache && v_cache && page_table && seq_lens && cu_seqlens_q && tile_cu,
               "attn_prefill: null pointer");
  MSGL_REQUIRE(head_dim == 128, "attn_prefill: head_dim %d unsupported (128 only)", head_dim);
  MSGL_REQUIRE(num_kv_heads >= 1 && num_q_heads % num_kv_heads == 0 && num_q_heads <= 65535,
               "attn_prefill: %d q heads / %d kv heads", num_q_heads, num_kv_heads);
  MSGL_REQUIRE(kv_stride_tok > 0 && kv_stride_tok < (1ll << 32), "attn_prefill: kv token stride %lld", (long long)kv_stride_tok);
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
  if (impl == 0) impl = 4;  // the DMA-staged kernel; impl 2 keeps its predecessor callable (A/B timing, cross-check in tests)
  const int64_t total = (int64_t)total_tiles * num_q_heads;
  MSGL_REQUIRE(total < (1ll << 30), "attn_prefill: %lld workgroups", (long long)total);
  const unsigned blocks = (unsigned)((total + 7) / 8) * 8;  // 8 XCDs x n_per
  if (impl == 4) {
    if (dtype == MSGL_BF16) attn_prefill_dma_kernel<BF16><<<dim3(blocks), dim3(256), 0, s>>>(p, total_tiles);
    else attn_prefill_dma_kernel<FP16><<<dim3(blocks), dim3(256), 0, s>>>(p, total_tiles);
  } else if (impl == 2) {
    if (dtype == MSGL_BF16) attn_prefill_tr_kernel<BF16><<<dim3(blocks), dim3(256), 0, s>>>(p, total_tiles);
    else attn_prefill_tr_kernel<FP16><<<dim3(blocks), dim3(256), 0, s>>>(p, total_tiles);
  }
#ifdef MSGL_PREFILL_DIAG
  else if (impl >= 64) {  // DMA-staged kernel, ablation bits = impl - 64 (timing only: WRONG results)
    MSGL_REQUIRE(dtype == MSGL_BF16, "attn_prefill: ablations are bf16 only");
#define MSGL_PF_DMA(A) case A: attn_prefill_dma_kernel<BF16, A><<<dim3(blocks), dim3(256), 0, s>>>(p, total_tiles); break
    switch (impl - 64) {
      MSGL_PF_DMA(1); MSGL_PF_DMA(2); MSGL_PF_DMA(4); MSGL_PF_DMA(8); MSGL_PF_DMA(6); MSGL_PF_DMA(12);
      MSGL_PF_DMA(10); MSGL_PF_DMA(14); MSGL_PF_DMA(15); MSGL_PF_DMA(16); MSGL_PF_DMA(32);
      default: set_error("attn_prefill: unknown ablation %d", impl - 64); return MSGL_EINVAL;
    }
#undef MSGL_PF_DMA
  } else {  // register-staged kernel, ablation bits = impl - 16
    MSGL_REQUIRE(dtype == MSGL_BF16, "attn_prefill: ablations are bf16 only");
#define MSGL_PF_ABL(A) case A: attn_prefill_tr_kernel<BF16, true, A><<<dim3(blocks), dim3(256), 0, s>>>(p, total_tiles); break
    switch (impl - 16) {
      MSGL_PF_ABL(1); MSGL_PF_ABL(2); MSGL_PF_ABL(4); MSGL_PF_ABL(8); MSGL_PF_ABL(6); MSGL_PF_ABL(10); MSGL_PF_ABL(12);
      MSGL_PF_ABL(14); MSGL_PF_ABL(15); MSGL_PF_ABL(17); MSGL_PF_ABL(31); MSGL_PF_ABL(16);
      default: set_error("attn_prefill: unknown ablation %d", impl - 16); return MSGL_EINVAL;
    }
#undef MSGL_PF_ABL
  }
#endif
  MSGL_CHECK_LAUNCH("attn_prefill");
  return MSGL_OK;
}
