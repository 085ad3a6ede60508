// Paged decode attention for gfx950: one query token per request, KV read through the
// reference's token-granular page table.
//
// HBM-bound (intensity = GQA group size, 2..8 flop/B), so the design is a streaming one:
//   * balanced flat split: the 16-token tiles of ALL requests form one line, cut into equal
//     slots (one per resident wave of a kv head) by a tiny device-side plan kernel once per
//     step; a slot is a short list of (request, tile range) pieces, so every wave streams the
//     same number of KV bytes whatever the raggedness of the batch (uniform chunks left the
//     slowest wave with 1.4x the mean at the bench shape).  The grid is fixed by
//     (capacity, max_bs) => legal inside hipGraph replay;
//   * one wave per unit, all G query heads of the GQA group packed in that wave so each
//     K/V byte is fetched from HBM exactly once;
//   * a 256-B K (or V) row of one token = one 16-lane DPP row, 16 B per lane => every
//     16-B-per-lane wave load reads 4 whole token rows (full 128-B lines); with page_size >= 16
//     the tile's table entry is ONE scalar load and K/V come through buffer loads (scalar
//     48-bit tile base in the descriptor + a lane-constant 32-bit offset): no vector address
//     arithmetic in the loop;
//   * q.k via v_dot2_f32_bf16 on the packed data (no unpack); the 16-lane reduction is a
//     reduce-scatter over a role-ordered tile (see the kernel): 5 G DPP adds per tile instead
//     of 16 G, the softmax scalar work (scale, max, exp) is done once per token instead of
//     once per lane, probabilities of the row's other three tokens return by 3 G DPP moves,
//     P.V with v_pk_fma_f32; softmax state (m, o) per 16-lane row over its token subset, l per
//     quad, merged once per unit.  (The all-reduce formulation it replaces measured the same
//     at G = 5 and 20 % slower at G = 8: profiles/r01c_microbench_decode_rs{0,1}.json.)
//   * K/V tiles double-buffered in registers one 16-token tile ahead (8 KB in flight per
//     wave), page-table slots prefetched two tiles ahead: no LDS, no barriers.
// Split-KV partials (fp32 o, m, l) go to a workspace and are merged by a second small
// kernel; single-chunk requests write their final output directly.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace msgl {

// Order pins.  hipcc on its own sinks the next tile's global loads behind the compute block (and
// floats the pure q.k dots up to the loads), which drains vmcnt(0) at the loop top and leaves nothing
// in flight while the wave computes.  MSGL_PIN_MEM: an empty asm with a memory clobber keeps the plain
// C++ loads on their side at IR level, sched_barrier(0) does the same in the machine scheduler.
// pin_tile() additionally routes a tile's registers through the asm, so nothing that consumes the
// tile can be placed before that point.  vmcnt bookkeeping stays with the compiler.
#define MSGL_PIN_MEM()                      \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)

typedef uint32_t V4 __attribute__((ext_vector_type(4)));
typedef int I4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor over "everything from base": 48-bit base, stride 0, num_records = 2^32 - 1 bytes
// (offsets used with it are < 2^31), gfx9 data-format word as used for untyped dword access
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, -1, 0x00020000);
}
typedef int I16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(4))) int CInt;                                // constant address space
typedef int I16a __attribute__((ext_vector_type(16), aligned(16)));
typedef __attribute__((address_space(4))) I16a CI16;

constexpr int kPlanHdr = 4;  // [0] n_items [1] tokens per slot [2] batch [3] n_slots
constexpr float kNegBig = -3.0e38f;

// plan buffer layout (int32 words):
//   hdr[4] | item_start[max_bs] | n_chunks[max_bs] | tile_start[max_bs] | slot_first[capacity + 1] |
//   items[4 * capacity] = (request, first tile, end tile, slot) |
//   items2[4 * capacity] = (seq_len, pieces of the request, its first piece, 0) | arrivals[max_bs * kTicketHeads]
// items2 repeats per piece what the kernels otherwise fetch by request AFTER the piece record arrived: with it a piece's
// scalars are two independent loads at an address known from the piece number alone (and are prefetched for piece i + 1
// while piece i streams).
// arrivals[b][kv head]: how many pieces of request b have published their partial sums in the running launch (matrix-core
// kernel under select code 72: the piece that arrives last combines them).  Zeroed by the plan kernel, left at zero by
// every launch.
constexpr int kTicketHeads = 64;
__host__ __device__ inline int64_t plan_off_item_start() { return kPlanHdr; }
__host__ __device__ inline int64_t plan_off_n_chunks(int max_bs) { return kPlanHdr + (int64_t)max_bs; }
__host__ __device__ inline int64_t plan_off_tile_start(int max_bs) { return kPlanHdr + 2ll * max_bs; }
__host__ __device__ inline int64_t plan_off_slot_first(int max_bs) { return kPlanHdr + 3ll * max_bs; }
__host__ __device__ inline int64_t plan_off_items(int max_bs, int capacity) {
  return ((kPlanHdr + 3ll * max_bs + capacity + 1 + 3) / 4) * 4;  // int4-aligned
}
__host__ __device__ inline int64_t plan_off_items2(int max_bs, int capacity) {
  return plan_off_items(max_bs, capacity) + 4ll * capacity;
}
__host__ __device__ inline int64_t plan_off_arrivals(int max_bs, int capacity) {
  return plan_off_items2(max_bs, capacity) + 4ll * capacity;
}

// ------------------------------------------------------------------------------
// plan: seq_lens -> balanced slots.  One block, 256 threads.
//   nt_b = ceil(S_b / 16) tiles, laid end to end (request-major); slot k owns global tiles
//   [k q, (k+1) q) with q = max(ceil(NT / target_slots), min_tiles).  A request overlapping m slots
//   contributes m pieces; pieces are numbered request-major, which also makes the pieces of one slot
//   consecutive.  Everything is closed-form from two prefix sums.
// ------------------------------------------------------------------------------
__device__ __forceinline__ int block_scan_excl_256(int mine, int* red, int* total) {
  const int tid = threadIdx.x;
  red[tid] = mine;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) {
      const int v = red[i];
      red[i] = run;
      run += v;
    }
    red[256] = run;
  }
  __syncthreads();
  const int r = red[tid];
  *total = red[256];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void decode_plan_kernel(int* __restrict__ plan,
                                                          const int* __restrict__ seq_lens, int batch,
                                                          int max_bs, int capacity, int target_slots,
                                                          int min_tiles) {
  __shared__ int red[257];
  const int tid = threadIdx.x;
  int* item_start = plan + plan_off_item_start();
  int* n_chunks = plan + plan_off_n_chunks(max_bs);
  int* tile_start = plan + plan_off_tile_start(max_bs);
  int* slot_first = plan + plan_off_slot_first(max_bs);
  int* items = plan + plan_off_items(max_bs, capacity);
  int* items2 = plan + plan_off_items2(max_bs, capacity);

  // thread t owns a contiguous request segment
  const int per = (batch + 255) / 256;
  const int b0 = min(tid * per, batch), b1 = min(b0 + per, batch);
  int mine = 0;
  for (int b = b0; b < b1; ++b) mine += (max(seq_lens[b], 0) + 15) >> 4;
  int NT;  // totals fit int32: batch * max_seq_len / 16 < 2^31 for every supported configuration
  int run = block_scan_excl_256(mine, red, &NT);
  for (int b = b0; b < b1; ++b) {
    tile_start[b] = run;
    run += (max(seq_lens[b], 0) + 15) >> 4;
  }

  // items <= n_slots + batch - 1 must fit the workspace
  int slots = min(target_slots, capacity - batch + 1);
  if (slots < 1) slots = 1;
  int q = (NT + slots - 1) / slots;
  if (q < min_tiles) q = min_tiles;
  if (q < 1) q = 1;
  const int n_slots = (NT + q - 1) / q;

  mine = 0;
  for (int b = b0; b < b1; ++b) {
    const int nt = (max(seq_lens[b], 0) + 15) >> 4;
    const int ts = tile_start[b];
    mine += nt ? ((ts + nt - 1) / q - ts / q + 1) : 0;
  }
  int n_items;
  run = block_scan_excl_256(mine, red, &n_items);
  for (int b = b0; b < b1; ++b) {
    const int nt = (max(seq_lens[b], 0) + 15) >> 4;
    const int ts = tile_start[b];
    const int k0 = ts / q;
    const int n = nt ? ((ts + nt - 1) / q - k0 + 1) : 0;
    item_start[b] = run;
    n_chunks[b] = n;
    for (int j = 0; j < n; ++j) {
      const int k = k0 + j;
      const int g0 = max(k * q, ts), g1 = min((k + 1) * q, ts + nt);
      int* it = items + 4 * (int64_t)(run + j);
      it[0] = b;
      it[1] = g0 - ts;
      it[2] = g1 - ts;
      it[3] = k;
      int* i2 = items2 + 4 * (int64_t)(run + j);
      i2[0] = seq_lens[b];
      i2[1] = n;
      i2[2] = run;
      i2[3] = 0;
      if (g0 == k * q) slot_first[k] = run + j;  // every slot starts with exactly one such piece
    }
    run += n;
  }
  if (tid == 0) {
    plan[0] = n_items;
    plan[1] = q * 16;
    plan[2] = batch;
    plan[3] = n_slots;
    slot_first[n_slots] = n_items;
  }
  int* arrivals = plan + plan_off_arrivals(max_bs, capacity);
  for (int i = tid; i < max_bs * kTicketHeads; i += 256) arrivals[i] = 0;
}

// ------------------------------------------------------------------------------
// partial attention
// ------------------------------------------------------------------------------
struct DecodeParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  const int* page_table;
  const int* req_rows;
  const int* seq_lens;
  const int* plan;
  uint16_t* out;
  float* part_o;
  float* part_ml;
  int64_t pt_stride, q_stride, kv_stride_tok, kv_stride_head, out_stride;
  int max_bs, capacity, hq, hv, group;  // hv = virtual kv heads (hq / G), group = hq / real kv heads
  int slot_run;               // aligned runs of this many positions map to consecutive slots (1: none)
  float scale_log2;
  unsigned long long* trace;  // diagnosis (msgl_attn_decode_trace): 16 clock stamps per wave, else nullptr
  int* arrivals;              // plan_off_arrivals: per (request, kv head) arrival counters of the matrix-core kernel
};

struct Tile {
  V4 k[4], v[4];
};

__device__ __forceinline__ void pin_tile(Tile& t) {
  asm volatile(""
               : "+v"(t.k[0]), "+v"(t.k[1]), "+v"(t.k[2]), "+v"(t.k[3]), "+v"(t.v[0]), "+v"(t.v[1]),
                 "+v"(t.v[2]), "+v"(t.v[3])
               :
               : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// waves per workgroup: with 2 waves/SIMD (3 <= G <= 5) one 8-wave workgroup fills a CU, and for the
// common hv = 8 it is exactly the 8 kv heads of ONE slot: they walk the same token rows in step, so a
// 2-KB token row (8 heads x 256 B) is consumed by one CU within a short window (DRAM page / TLB locality).
#ifndef MSGL_DECODE_WPB
#define MSGL_DECODE_WPB(G) (((G) >= 3 && (G) <= 5) ? 8 : 4)
#define MSGL_DECODE_MINW(G) ((G) <= 2 ? 3 : (G) <= 5 ? 2 : 1)
#endif
template <int G>
struct DecodeGeom {
  static constexpr int kWavesPerBlock = MSGL_DECODE_WPB(G);
  static constexpr int kMinWavesPerSimd = MSGL_DECODE_MINW(G);
};

template <typename T, int G, bool kRun>
__global__ __launch_bounds__(64 * DecodeGeom<G>::kWavesPerBlock, DecodeGeom<G>::kMinWavesPerSimd) void
attn_decode_kernel(const DecodeParams p) {
  constexpr int D = 128;
  constexpr int kWPB = DecodeGeom<G>::kWavesPerBlock;
  const int lane = threadIdx.x & 63;
  const int r = lane >> 4;  // DPP row = token sub-slot
  const int c = lane & 15;  // 16-byte piece of the 256-B head row
  const int gw = sgpr((int)blockIdx.x * kWPB + (int)(threadIdx.x >> 6));
  const int slot = gw / p.hv;
  const int h = gw - slot * p.hv;
  const int n_slots = p.plan[3];
  if (slot >= n_slots) return;
  const int* n_chunks = p.plan + plan_off_n_chunks(p.max_bs);
  const int* slot_first = p.plan + plan_off_slot_first(p.max_bs);
  const int4* items = reinterpret_cast<const int4*>(p.plan + plan_off_items(p.max_bs, p.capacity));
  const int item_begin = sgpr(slot_first[slot]);
  const int item_end = sgpr(slot_first[slot + 1]);

  for (int item = item_begin; item < item_end; ++item) {
    const int4 it = items[item];
    const int b = sgpr(it.x);
    const int S = sgpr(p.seq_lens[b]);
    const int t0 = sgpr(it.y) * 16;
    const int t1 = min(S, sgpr(it.z) * 16);
    const int row = p.req_rows ? sgpr(p.req_rows[b]) : b;
    const int* pt = p.page_table + (int64_t)row * p.pt_stride;
    const int hq0 = h * G;
    const int kvh = hq0 / p.group;

    uint32_t qr[G][4];
    {
      const uint16_t* qp = p.q + (int64_t)b * p.q_stride + (int64_t)hq0 * D + c * 8;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const U4 u = ldg16(qp + g * D);
        qr[g][0] = u.x; qr[g][1] = u.y; qr[g][2] = u.z; qr[g][3] = u.w;
      }
    }
    const uint16_t* kb = p.k + (int64_t)kvh * p.kv_stride_head + c * 8;
    const uint16_t* vb = p.v + (int64_t)kvh * p.kv_stride_head + c * 8;
    // role-ordered tile: the lane's quad "owns" token 4r + own of each 16-token tile; register j of
    // a tile holds the token at row offset own ^ {0, 1, 3, 2}[j], i.e. j = 0 the owned token, j = 1 the one
    // its half-mirror partner owns, j = 2 / 3 the ones its row-mirror / ror-8 partners own.  With that
    // order the 16-lane reduction of the 4 x G scores is a reduce-scatter (G * (2 + 1 + 2) DPP adds
    // instead of 4 * G * 4): every lane ends with the G scores of ITS token only, the softmax scalar
    // work is done once per token instead of once per lane, and the probabilities of the other three
    // tokens come back with three DPP moves per head.
    const int own = (c >> 2) & 3;
    const int tok_off[4] = {own, own ^ 1, own ^ 3, own ^ 2};
    uint32_t voff[4];  // byte offset of (row token, kv head, 16-B piece) from the tile's first token row
#pragma unroll
    for (int j = 0; j < 4; ++j)
      voff[j] = (uint32_t)((((int64_t)(4 * r + tok_off[j])) * p.kv_stride_tok + (int64_t)kvh * p.kv_stride_head +
                            c * 8) * 2);

    float m[G], l[G], o[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      m[g] = kNegBig;
      l[g] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[g][e] = 0.f;
    }

    // Page-table slots of a 16-token tile (tokens tb+4r+i for lane row r, load i).
    //  kRun (caller guarantees that every aligned run of 16 positions maps to 16 consecutive slots,
    //  i.e. page_size >= 16 with the reference's page-aligned allocation, P/scheduler/cache.py:42-53,
    //  127-146 -- the same property fa.py:92-97 relies on): ONE wave-uniform int per tile, read through
    //  the scalar cache (constant address space => s_load, lgkmcnt), so the in-order vmcnt queue holds
    //  K/V loads only and the per-tile address arithmetic is scalar.
    //  generic: lane row r loads its own 4 slots as one 16-B vector load, issued AFTER the tile's K/V
    //  loads (vmcnt retires in order: waiting for the older tile must not wait for this load).
    // Entries at or past t1 are never dereferenced (such tokens re-read the tile's first token).
    using Slots = std::conditional_t<kRun, int, int4>;
    const CInt* cpt = (const CInt*)pt;
    auto slots_raw = [&](int tb) -> Slots {
      if constexpr (kRun) {
        return cpt[tb];
      } else {
        int tq = tb + 4 * r;
        if (tq >= t1) tq = tb;
        return *reinterpret_cast<const int4*>(pt + tq);
      }
    };
    auto slots_fix = [&](Slots sl, int tb) -> Slots {
      if constexpr (kRun) {
        asm volatile("" : "+s"(sl));  // consumers (scalar address math) must not float above this point
      } else {
        int tq = tb + 4 * r;
        if (tq >= t1) tq = tb;
        if (tq + 1 >= t1) sl.y = sl.x;
        if (tq + 2 >= t1) sl.z = sl.x;
        if (tq + 3 >= t1) sl.w = sl.x;
      }
      return sl;
    };
    auto load_slots = [&](int tb) -> Slots { return slots_fix(slots_raw(tb), tb); };
    // full tile: every token valid
    auto pick = [&](const int4& sl4, int idx) -> int {
      return idx == 0 ? sl4.x : idx == 1 ? sl4.y : idx == 2 ? sl4.z : sl4.w;
    };
    auto load_tile = [&](const Slots& sl, Tile& t) {
      if constexpr (kRun) {
        // buffer loads: a scalar 48-bit base (pool + first slot of the tile) in the descriptor plus the
        // lane's constant 32-bit offset => no vector address arithmetic and no 64-bit pointers in VGPRs
        const int64_t tile_bytes = (int64_t)sl * p.kv_stride_tok * 2;
        const __amdgpu_buffer_rsrc_t kd = make_rsrc(reinterpret_cast<const char*>(p.k) + tile_bytes);
        const __amdgpu_buffer_rsrc_t vd = make_rsrc(reinterpret_cast<const char*>(p.v) + tile_bytes);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t.k[j] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(kd, (int)voff[j], 0, 0));
          t.v[j] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(vd, (int)voff[j], 0, 0));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t off = (int64_t)pick(sl, tok_off[j]) * p.kv_stride_tok;
          t.k[j] = *reinterpret_cast<const V4*>(kb + off);
          t.v[j] = *reinterpret_cast<const V4*>(vb + off);
        }
      }
    };
    // possibly partial tile (the unit's last one)
    auto load_tile_tail = [&](const Slots& sl, Tile& t, int tb) {
      if constexpr (kRun) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int tk = 4 * r + tok_off[i];
          if (tb + tk >= t1) tk = 0;
          const int64_t off = (int64_t)(sl + tk) * p.kv_stride_tok;
          t.k[i] = *reinterpret_cast<const V4*>(kb + off);
          t.v[i] = *reinterpret_cast<const V4*>(vb + off);
        }
      } else {
        load_tile(sl, t);
      }
    };
    auto compute = [&](const Tile& t, int tb, auto masked_tag) {
      constexpr bool kMasked = decltype(masked_tag)::value;
      float acc[G][4];  // partial dots over this lane's 8 dims, by role
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float a = Elem<T>::dot2_first(qr[g][0], t.k[j].x);
          a = Elem<T>::dot2(qr[g][1], t.k[j].y, a);
          a = Elem<T>::dot2(qr[g][2], t.k[j].z, a);
          a = Elem<T>::dot2(qr[g][3], t.k[j].w, a);
          acc[g][j] = a;
        }
      }
      // reduce-scatter over the 16 lanes of the row: the row-mirror partner (own ^ 3) sends its role-2/3
      // partials, which are this lane's role-0/1 tokens; the half-mirror partner (own ^ 1) its role-1;
      // then an all-reduce inside the quad.  Every lane of the row is summed exactly once.
      float sc[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float a0 = acc[g][0] + dpp_get<kDppRowMirror>(acc[g][2]);
        const float a1 = acc[g][1] + dpp_get<kDppRowMirror>(acc[g][3]);
        float a = a0 + dpp_get<kDppHalfMirror>(a1);
        a += dpp_get<kDppXor2>(a);
        a += dpp_get<kDppXor1>(a);
        sc[g] = a * p.scale_log2;
      }
      if constexpr (kMasked) {
        if (tb + 4 * r + own >= t1) {
#pragma unroll
          for (int g = 0; g < G; ++g) sc[g] = -INFINITY;
        }
      }
      float pr[G][4];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float mx = fmaxf(sc[g], dpp_get<kDppRowMirror>(sc[g]));  // over the row's four tokens
        mx = fmaxf(mx, dpp_get<kDppHalfMirror>(mx));
        mx = fmaxf(mx, m[g]);
        const float alpha = __builtin_amdgcn_exp2f(m[g] - mx);
        const float pe = __builtin_amdgcn_exp2f(sc[g] - mx);
        l[g] = fmaf(l[g], alpha, pe);  // this quad's token stream only; quads are summed at the end
        m[g] = mx;
        pr[g][0] = pe;
        pr[g][1] = dpp_get<kDppHalfMirror>(pe);
        pr[g][2] = dpp_get<kDppRowMirror>(pe);
        pr[g][3] = dpp_get<kDppRor8>(pe);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[g][e] *= alpha;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float vf[8] = {Elem<T>::lo(t.v[j].x), Elem<T>::hi(t.v[j].x), Elem<T>::lo(t.v[j].y),
                             Elem<T>::hi(t.v[j].y), Elem<T>::lo(t.v[j].z), Elem<T>::hi(t.v[j].z),
                             Elem<T>::lo(t.v[j].w), Elem<T>::hi(t.v[j].w)};
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[g][e] = fmaf(pr[g][j], vf[e], o[g][e]);
        }
      }
    };
    using Full = std::integral_constant<bool, false>;
    using Masked = std::integral_constant<bool, true>;

    const int ntiles = (t1 - t0 + 15) >> 4;
    const int last = t0 + (ntiles - 1) * 16;
    Tile A, B;
    Slots sA = load_slots(t0);
    if (ntiles > 1) load_tile(sA, A);
    else load_tile_tail(sA, A, t0);
    Slots sB = load_slots(min(t0 + 16, last));
    int tix = 0;
    // steady state: tiles tix and tix+1 are full (tix+2 exists), no branches inside.
    // order per half: issue the next tile's loads, pin, compute the resident tile (see MSGL_PIN_MEM)
    for (; tix + 3 < ntiles; tix += 2) {
      const int tb = t0 + tix * 16;
      const int tb3 = min(tb + 48, last);
      Slots rawA, rawB;
      if constexpr (kRun) rawA = slots_raw(tb + 32);
      load_tile(sB, B);
      if constexpr (!kRun) rawA = slots_raw(tb + 32);
      pin_tile(A);  // waits for A only: B (issued just above) stays in flight under compute(A)
      compute(A, tb, Full{});
      MSGL_PIN_MEM();
      sA = slots_fix(rawA, tb + 32);
      if constexpr (kRun) rawB = slots_raw(tb3);
      load_tile(sA, A);  // tile tix+2 is full: tix+3 < ntiles
      if constexpr (!kRun) rawB = slots_raw(tb3);
      pin_tile(B);
      compute(B, tb + 16, Full{});
      MSGL_PIN_MEM();
      sB = slots_fix(rawB, tb3);
    }
    // one, two or three tiles left; A holds tile tix, sB the slots of tile tix+1 (if any)
    for (; tix < ntiles; ++tix) {
      const int tb = t0 + tix * 16;
      const bool more = tix + 1 < ntiles;
      if (more) {
        if (tix + 2 < ntiles) load_tile(sB, B);
        else load_tile_tail(sB, B, tb + 16);
      }
      if (more) compute(A, tb, Full{});
      else compute(A, tb, Masked{});
      if (more) {
        A = B;
        sB = load_slots(min(tb + 32, last));
      }
    }

    // l was kept per quad (token stream); m and o are already per row
#pragma unroll
    for (int g = 0; g < G; ++g) {
      l[g] += dpp_mov<kDppRowMirror>(l[g]);
      l[g] += dpp_mov<kDppHalfMirror>(l[g]);
    }
    // merge the four DPP rows (disjoint token subsets) -> every lane holds the unit's state
#pragma unroll
    for (int step = 16; step <= 32; step <<= 1) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float m2 = __shfl_xor(m[g], step, 64);
        const float l2 = __shfl_xor(l[g], step, 64);
        const float mx = fmaxf(m[g], m2);
        const float a1 = __builtin_amdgcn_exp2f(m[g] - mx);
        const float a2 = __builtin_amdgcn_exp2f(m2 - mx);
        l[g] = l[g] * a1 + l2 * a2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float o2 = __shfl_xor(o[g][e], step, 64);
          o[g][e] = o[g][e] * a1 + o2 * a2;
        }
        m[g] = mx;
      }
    }

    const bool single = sgpr(n_chunks[b]) == 1;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if ((g & 3) != r) continue;  // spread the G head rows over the four DPP rows
      const int hq = hq0 + g;
      if (single) {
        const float inv = 1.0f / l[g];
        U4 u;
        u.x = Elem<T>::pack(o[g][0] * inv, o[g][1] * inv);
        u.y = Elem<T>::pack(o[g][2] * inv, o[g][3] * inv);
        u.z = Elem<T>::pack(o[g][4] * inv, o[g][5] * inv);
        u.w = Elem<T>::pack(o[g][6] * inv, o[g][7] * inv);
        stg16(p.out + (int64_t)b * p.out_stride + (int64_t)hq * D + c * 8, u);
      } else {
        float* po = p.part_o + ((int64_t)item * p.hq + hq) * D + c * 8;
        *reinterpret_cast<float4*>(po) = make_float4(o[g][0], o[g][1], o[g][2], o[g][3]);
        *reinterpret_cast<float4*>(po + 4) = make_float4(o[g][4], o[g][5], o[g][6], o[g][7]);
        if (c == 0) {
          float* pm = p.part_ml + ((int64_t)item * p.hq + hq) * 2;
          *reinterpret_cast<float2*>(pm) = make_float2(m[g], l[g]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------
// partial attention, matrix-core formulation (pages of >= 16 tokens).
//
// The streaming kernel above spends ~350 VALU operations per 16-token tile and wave (dot2 chains, DPP
// reduce-scatter, fp32 P.V) and half of its wave cycles waiting to issue them (profiles/r01e_pmc_sq_*).  Here the
// two products go to the otherwise idle matrix cores and the wave is left with ~30 VALU operations per tile:
//   S^T[16 tokens x 16 heads] = K[16 x 128] . Q^T[128 x 16]   4 x v_mfma_f32_16x16x32 (heads >= G are zero columns)
//        A = K straight from HBM: lane (token = l & 15, quarter = l >> 4) loads 16 B at dims 32 kk + 8 quarter;
//   the result puts ONE head (l & 15) and FOUR tokens (4 (l >> 4) + i) in a lane, which is exactly the B-operand
//   layout of the second product, so P never moves between lanes:
//   O^T[128 dims x 16 heads] += V^T[128 x 16 tokens] . P^T[16 x 16]   8 x v_mfma_f32_16x16x16
//        A = V^T: the V rows land in a wave-private 4-KB LDS image (row-major, XOR-swizzled) and come back
//        transposed through ds_read_b64_tr_b16; no barrier (one wave owns the image, LDS is in order per wave).
// Softmax: the tile max over the four lane groups by v_permlane16/32_swap; the O rescale (32 registers) runs only
// on tiles where some head's running max moved.  P is rounded to the 16-bit type for the second product (as in
// the prefill kernel and in the reference's tensor-core decode, flashinfer), l is summed from the fp32 values.
// K/V tiles sit in a ring of kStages register sets: kStages - 1 tiles (8 KB each) in flight per wave while one
// is consumed.  Work split (plan, slots, pieces), partial layout and merge are those of the streaming kernel.
// ------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 dbf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 df16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 df16x4;
typedef __attribute__((ext_vector_type(4))) short ds16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) ds16x4 lds_ds16x4;
typedef __attribute__((address_space(3))) char dlds_char;

template <typename T>
__device__ __forceinline__ f32x4 mfma_k32(const V4& a, const V4& b, f32x4 c) {
  if constexpr (std::is_same_v<T, BF16>)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dbf16x8, a), __builtin_bit_cast(dbf16x8, b), c, 0,
                                                   0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(df16x8, a), __builtin_bit_cast(df16x8, b), c, 0, 0,
                                                  0);
}
template <typename T>
__device__ __forceinline__ f32x4 mfma_k16(const uint2& a, const uint2& b, f32x4 c) {
  if constexpr (std::is_same_v<T, BF16>)
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(ds16x4, a), __builtin_bit_cast(ds16x4, b), c, 0, 0,
                                                     0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(df16x4, a), __builtin_bit_cast(df16x4, b), c, 0, 0, 0);
}

// max over the four 16-lane groups (every lane ends with it)
__device__ __forceinline__ float max_over_rows(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const uint32_t w = __float_as_uint(x);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float sum_over_rows(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const uint32_t w = __float_as_uint(x);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// V image: [16 tokens][256 B]; 64-B blocks XORed by the token's low two bits, the 32-B half by bit 2: the 32 lanes
// of one ds_read_b64_tr_b16 pass (16 lanes x 2 groups) then cover 32 distinct 8-B bank pairs
__device__ __forceinline__ int vimg_off(int tok, int byte_in_row) {
  return tok * 256 + (byte_in_row ^ ((tok & 3) << 6) ^ (((tok >> 2) & 1) << 5));
}

// K image (kLines): [16 tokens][256 B], the row's sixteen 16-byte pieces XORed by the token: the 16 lanes of one ds_read_b128
// pass (one piece index, tokens 0..15) and of one ds_write_b128 pass (one token, pieces 0..15) both cover all 64 banks
__device__ __forceinline__ int kimg_off(int tok, int piece) { return tok * 256 + (((piece ^ tok) & 15) << 4); }

// kStages: register sets of the request ring; kMfmaWaves: waves per workgroup (8 = the kv heads of one slot at
// hv = 8); kMinW: waves per SIMD the register budget is held to
// kLoadsOnly (diagnosis, variant 92): the same requests and waits with the products left out -- what the request
// pattern alone costs
template <typename T, int kStages, int kMfmaWaves, int kMinW, bool kLoadsOnly = false, bool kTrace = false,
          bool kCombine = false, bool kPrefetch = true, bool kLines = true, bool kRun = true>
__global__ __launch_bounds__(64 * kMfmaWaves, kMinW) void attn_decode_mfma_kernel(const DecodeParams p) {
  static_assert(kRun || kLines, "token-granular tables are gathered by whole-line requests only");
  constexpr int D = 128;
  // per wave: the V image (4 KB) and, with kLines, the K image (4 KB) behind it
  __shared__ __attribute__((aligned(16))) char lds[kMfmaWaves * (kLines ? 8192 : 4096)];
  unsigned long long stamp[14];  // kTrace (variant 93) only
  int n_stamp = 0;
  auto mark = [&]() {
    if constexpr (kTrace) {
      if (n_stamp < 14) stamp[n_stamp] = __builtin_readcyclecounter();
      ++n_stamp;
    }
  };
  mark();  // 0: entry
  const int lane = threadIdx.x & 63;
  const int tok = lane & 15;  // token row of the tile for loads; head column for the products' results
  const int qd = lane >> 4;   // 8-dim quarter of a 32-dim step for loads; token group of the results
  const int wv = sgpr((int)(threadIdx.x >> 6));
  const int gw = sgpr((int)blockIdx.x * kMfmaWaves + wv);
  const int slot = gw / p.hv;
  const int h = gw - slot * p.hv;
  const int n_slots = p.plan[3];
  if (slot >= n_slots) return;
  const int G = sgpr(p.hq / p.hv);
  const int* n_chunks = p.plan + plan_off_n_chunks(p.max_bs);
  const int* slot_first = p.plan + plan_off_slot_first(p.max_bs);
  const int4* items = reinterpret_cast<const int4*>(p.plan + plan_off_items(p.max_bs, p.capacity));
  const int item_begin = sgpr(slot_first[slot]);
  const int item_end = sgpr(slot_first[slot + 1]);
  dlds_char* img = (dlds_char*)lds + wv * (kLines ? 8192 : 4096);
  // kLines: request j of a tile covers token rows 4 j + qd, lane piece `tok` (16 B) of the head's 256-B row -- whole 128-B
  // lines per request, which is what lets the nt policy pay (see the banner above the kernel)
  // !kRun (token-granular page tables, round 6): lane group qd owns the tile's tokens 4 qd .. 4 qd + 3 -- ONE 16-byte table
  // read per tile gives it their four pool slots -- and request kk fetches token 4 qd + kk's row by a per-lane address
  int wr_off[4], rd_off[8], kwr_off[4], krd_off[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int wtok = kRun ? 4 * kk + qd : 4 * qd + kk;  // token of the tile this lane's request kk carries (kLines)
    wr_off[kk] = kLines ? vimg_off(wtok, tok * 16) : vimg_off(tok, kk * 64 + qd * 16);
    kwr_off[kk] = 4096 + kimg_off(wtok, tok);
    krd_off[kk] = 4096 + kimg_off(tok, 4 * kk + qd);
  }
#pragma unroll
  for (int db = 0; db < 8; ++db) rd_off[db] = vimg_off(4 * qd + (tok >> 2), db * 32 + (tok & 3) * 8);

  const __amdgpu_buffer_rsrc_t part_rsrc = make_rsrc(p.part_o);  // the fp32 partial sums (offsets < 2^31: checked at launch)
  mark();  // 1: slot known
  // A piece's scalars come from its two plan records (address known from the piece number); the records of piece i + 1, its
  // table row and the pool slots of its first tiles are fetched while piece i's first tiles are in flight, so a wave's
  // second and later pieces start requesting K/V without a metadata chain in front (tools/decode_trace.py measured
  // ~10 k clocks of it per further piece with the loads queued behind the stream).
  const int4* items2 = reinterpret_cast<const int4*>(p.plan + plan_off_items2(p.max_bs, p.capacity));
  if (item_begin >= item_end) return;  // (every slot of a plan starts with a piece; this keeps a corrupt plan from indexing with garbage)
  int4 it = items[item_begin], it2 = items2[item_begin];
  int b = sgpr(it.x);
  int row = p.req_rows ? sgpr(p.req_rows[b]) : b;
  using Slots = std::conditional_t<kRun, int, int4>;  // a tile's pool slots: its first one (runs of >= 16) / the lane group's four
  Slots pre_sl[kStages];  // pool slots of the first kStages tiles of the piece about to start (valid from the second piece on)
#pragma unroll
  for (int st = 0; st < kStages; ++st) pre_sl[st] = Slots{};
  for (int item = item_begin; item < item_end; ++item) {
    if constexpr (!kPrefetch) {  // select code 94 (diagnosis): the dependent chain of the earlier form, for same-process A/B
      it = items[item];
      b = sgpr(it.x);
      it2.x = p.seq_lens[b];
      row = p.req_rows ? sgpr(p.req_rows[b]) : b;
      it2.y = n_chunks[b];
      it2.z = p.plan[plan_off_item_start() + b];
    }
    const int S = sgpr(it2.x);
    const int t0 = sgpr(it.y) * 16;
    const int t1 = min(S, sgpr(it.z) * 16);
    const int nch = sgpr(it2.y);
    const int first_item = sgpr(it2.z);
    const bool single = nch == 1;
    const CInt* cpt = (const CInt*)(p.page_table + (int64_t)row * p.pt_stride);
    const int nxt = min(item + 1, item_end - 1);  // the last piece re-reads its own records
    const int4 nit = items[nxt], nit2 = items2[nxt];
    const int hq0 = h * G;
    const int kvh = hq0 / p.group;
    mark();  // 2 + 4 i: piece i metadata known

    V4 qf[4];  // B operand of the first product: column = head (tok), k = dims 32 kk + 8 qd ..
    {
      const uint16_t* qp = p.q + (int64_t)b * p.q_stride + (int64_t)(hq0 + tok) * D + qd * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        V4 z = {0u, 0u, 0u, 0u};
        if (tok < G) z = *reinterpret_cast<const V4*>(qp + kk * 32);
        qf[kk] = z;
      }
    }
    // byte offset of (token row tok, kv head, quarter) from the tile's first token row; step kk adds 64 B
    // kLines: (token row qd of a group of four, piece tok); request kk adds four token rows
    const uint32_t voff = kLines ? (uint32_t)(((int64_t)qd * p.kv_stride_tok + (int64_t)kvh * p.kv_stride_head + tok * 8) * 2)
                                 : (uint32_t)(((int64_t)tok * p.kv_stride_tok + (int64_t)kvh * p.kv_stride_head + qd * 8) * 2);
    const uint32_t voff0 = (uint32_t)(((int64_t)kvh * p.kv_stride_head + (kLines ? tok : qd) * 8) * 2);
    const int row4 = sgpr((int)(p.kv_stride_tok * 8));  // bytes of four token rows (kLines)

    float m = kNegBig, l = 0.f;
    f32x4 o[8];
#pragma unroll
    for (int db = 0; db < 8; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = (t1 - t0 + 15) >> 4;
    // Request tile `ti` of the piece (16 tokens from position t0 + 16 ti, first pool slot `sl`).  Tokens at or past t1
    // re-read the tile's first token (finite data for the masked lanes).  A tile past the piece's end is requested
    // through a descriptor of ZERO records: the eight loads still issue and retire in order -- the compiler's vmcnt
    // bookkeeping stays exact, with no branch around the requests -- but touch no memory and return zeros.
    // token-granular tables: the four slots of tile-local tokens 4 qd .. 4 qd + 3 of the tile at position tb of table row `pt`
    // whose tokens end at `end`; a lane group past the end reads the tile's first four entries instead (entries at or beyond
    // a request's length are never dereferenced; load_tile replaces those of a partial group by a valid token's slot)
    auto table_slots = [&](const int* pt, int tb, int end) -> int4 {
      int tq = tb + 4 * qd;
      if (tq >= end) tq = tb;
      return *reinterpret_cast<const int4*>(pt + tq);
    };
    const int* vpt = p.page_table + (int64_t)row * p.pt_stride;
    const uint64_t lane_bytes = (uint64_t)(((int64_t)kvh * p.kv_stride_head + tok * 8) * 2);  // (kv head, piece) inside a token row
    const uint32_t row_bytes = (uint32_t)(p.kv_stride_tok * 2);
    auto load_tile = [&](Slots sl, Tile& t, int ti) {
      if constexpr (!kRun) {
        // the slots in `sl` are those of tile min(ti, lt).  A tile past the piece's end still issues its eight requests (the
        // compiler's vmcnt bookkeeping stays exact, no branch around them) but every lane then asks for the SAME 16 bytes of
        // a valid row: one L2 hit per request, no HBM traffic.
        const int tbc = t0 + min(ti, ntiles - 1) * 16;
        int tq = tbc + 4 * qd;
        if (tq >= t1) tq = tbc;
        const int s4[4] = {sl.x, tq + 1 < t1 ? sl.y : sl.x, tq + 2 < t1 ? sl.z : sl.x, tq + 3 < t1 ? sl.w : sl.x};
        const bool dead = ti >= ntiles;
        const int s0 = sgpr(sl.x);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const uint64_t off = (uint64_t)(uint32_t)(dead ? s0 : s4[kk]) * row_bytes + (dead ? 0ull : lane_bytes);
          const char* ka = reinterpret_cast<const char*>(p.k) + off;
          const char* va = reinterpret_cast<const char*>(p.v) + off;
          t.k[kk] = __builtin_nontemporal_load(reinterpret_cast<const V4*>(ka));
          t.v[kk] = __builtin_nontemporal_load(reinterpret_cast<const V4*>(va));
        }
        return;
      } else {
      const int64_t tile_bytes = (int64_t)sl * p.kv_stride_tok * 2;
      const int records = sgpr(ti < ntiles ? -1 : 0);
      const __amdgpu_buffer_rsrc_t kd = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(p.k) + tile_bytes), (short)0, records, 0x00020000);
      const __amdgpu_buffer_rsrc_t vd = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(p.v) + tile_bytes), (short)0, records, 0x00020000);
      if constexpr (kLines) {
        const int left = t1 - (t0 + ti * 16) - qd;  // token 4 kk + qd of the tile is valid iff 4 kk < left
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int vo = 4 * kk < left ? (int)voff + kk * row4 : (int)voff0;
          t.k[kk] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(kd, vo, 0, 2));
          t.v[kk] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(vd, vo, 0, 2));
        }
      } else {
        const int vo = (int)(t0 + ti * 16 + tok < t1 ? voff : voff0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          t.k[kk] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(kd, vo + kk * 64, 0, 0));
          t.v[kk] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(vd, vo + kk * 64, 0, 0));
        }
      }
      }
    };
    auto compute = [&](const Tile& t, int tb) {
      if constexpr (kLoadsOnly) {
        l += __uint_as_float((t.k[0].x ^ t.k[1].y ^ t.k[2].z ^ t.k[3].w ^ t.v[0].x ^ t.v[1].y ^ t.v[2].z ^ t.v[3].w) & 0x3fffffffu);
        return;
      }
      // V rows -> the wave's LDS image (the transposing reads below come after these writes: LDS is in order)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) *reinterpret_cast<__attribute__((address_space(3))) V4*>(img + wr_off[kk]) = t.v[kk];
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      if constexpr (kLines) {  // K rows -> the wave's K image, back in A-operand layout (token = lane & 15, dims 32 kk + 8 qd ..)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) *reinterpret_cast<__attribute__((address_space(3))) V4*>(img + kwr_off[kk]) = t.k[kk];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const V4 kf = *reinterpret_cast<__attribute__((address_space(3))) V4*>(img + krd_off[kk]);
          s = mfma_k32<T>(kf, qf[kk], s);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) s = mfma_k32<T>(t.k[kk], qf[kk], s);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)  // only the piece's last tile can be partial; four selects are cheaper than a second body
        if (tb + 4 * qd + i >= t1) s[i] = -INFINITY;
      float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      mx = max_over_rows(mx) * p.scale_log2;
      if (__builtin_amdgcn_ballot_w64(mx > m) != 0) {  // some head's running max moved (wave-uniform branch)
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        l *= alpha;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
#pragma unroll
          for (int i = 0; i < 4; ++i) o[db][i] *= alpha;
        }
      }
      float pe[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pe[i] = __builtin_amdgcn_exp2f(fmaf(s[i], p.scale_log2, -m));
      l += (pe[0] + pe[1]) + (pe[2] + pe[3]);
      uint2 pf;
      pf.x = Elem<T>::pack(pe[0], pe[1]);
      pf.y = Elem<T>::pack(pe[2], pe[3]);
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        const ds16x4 vt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ds16x4*)(img + rd_off[db]));
        o[db] = mfma_k16<T>(__builtin_bit_cast(uint2, vt), pf, o[db]);
      }
    };
    // One branch-free request pattern for the whole piece: kStages - 1 tiles in flight, then per tile: request tile
    // cur + kStages - 1, wait for tile cur (pin_tile: that tile only, the younger requests stay in flight under the
    // products), consume it.  Pool slots come through the scalar cache one tile ahead of their use.
    Tile ring[kStages];
    const int lt = ntiles - 1;
    // (token-granular tables at three waves per SIMD: no register room for the next piece's slots, they are read at its start)
    constexpr bool kPreSlots = kPrefetch && (kRun || kMinW < 3);
    const bool have_pre = kPreSlots && item > item_begin;
    auto tile_slots = [&](int ti) -> Slots {  // of tile min(ti, lt) of this piece
      if constexpr (kRun) return cpt[t0 + min(ti, lt) * 16];
      else return table_slots(vpt, t0 + min(ti, lt) * 16, t1);
    };
#pragma unroll
    for (int st = 0; st < kStages - 1; ++st) {
      load_tile(have_pre ? pre_sl[st] : tile_slots(st), ring[st], st);
      MSGL_PIN_MEM();  // oldest tile first: the scheduler is otherwise free to request them in any order
    }
    Slots sn = have_pre ? pre_sl[kStages - 1] : tile_slots(kStages - 1);  // slot(s) of the next tile to request
    // the next piece's row and first slots (its records were requested above; this wait overlaps the first tiles' flight)
    const int nb = sgpr(nit.x);
    const int nrow = p.req_rows ? sgpr(p.req_rows[nb]) : nb;
    {
      const CInt* ncpt = (const CInt*)(p.page_table + (int64_t)nrow * p.pt_stride);
      const int nt0 = sgpr(nit.y) * 16;
      const int nt1 = min(sgpr(nit2.x), sgpr(nit.z) * 16);
      const int nlt = ((nt1 - nt0 + 15) >> 4) - 1;
#pragma unroll
      for (int st = 0; st < kStages; ++st) {
        if constexpr (kRun) pre_sl[st] = ncpt[nt0 + min(st, nlt) * 16];
        else if constexpr (kPreSlots) pre_sl[st] = table_slots(p.page_table + (int64_t)nrow * p.pt_stride, nt0 + min(st, nlt) * 16, nt1);
      }
    }
    for (int tix = 0; tix < ntiles; tix += kStages) {
#pragma unroll
      for (int st = 0; st < kStages; ++st) {
        const int cur = tix + st;
        // (token-granular tables: a vector load, requested BEFORE the tile's K/V so that waiting for it next round does not
        // wait for them -- vmcnt retires in order)
        const Slots raw = tile_slots(cur + kStages);
        if constexpr (!kRun) MSGL_PIN_MEM();
        load_tile(sn, ring[(st + kStages - 1) % kStages], cur + kStages - 1);
        pin_tile(ring[st]);
        if constexpr (kTrace) {
          if (cur == 0) mark();  // 3 + 4 i: first tile of the piece has arrived
        }
        if (cur < ntiles) compute(ring[st], t0 + cur * 16);
        MSGL_PIN_MEM();
        sn = raw;
        if constexpr (kRun) asm volatile("" : "+s"(sn));
      }
    }

    mark();  // 4 + 4 i: last tile consumed
    l = sum_over_rows(l);  // each lane summed its own four tokens per tile
    if (tok < G) {
      const int hq = hq0 + tok;
      if (single) {
        const float inv = 1.0f / l;
        uint16_t* op = p.out + (int64_t)b * p.out_stride + (int64_t)hq * D + 4 * qd;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
          uint2 u;
          u.x = Elem<T>::pack(o[db][0] * inv, o[db][1] * inv);
          u.y = Elem<T>::pack(o[db][2] * inv, o[db][3] * inv);
          *reinterpret_cast<uint2*>(op + db * 16) = u;
        }
      } else if constexpr (!kCombine) {
        float* po = p.part_o + ((int64_t)item * p.hq + hq) * D + 4 * qd;
#pragma unroll
        for (int db = 0; db < 8; ++db)
          *reinterpret_cast<float4*>(po + db * 16) = make_float4(o[db][0], o[db][1], o[db][2], o[db][3]);
        if (qd == 0) {
          float* pm = p.part_ml + ((int64_t)item * p.hq + hq) * 2;
          *reinterpret_cast<float2*>(pm) = make_float2(m, l);
        }
      } else {
        // publish write-through (sc1): the partial sums are complete in memory once this wave's vmcnt drains, with no
        // release fence (cdna_hip_programming.md, split-K hand-off by arrival counter, the write-through form)
        const int vo = (int)((((int64_t)item * p.hq + hq) * D + 4 * qd) * 4);
#pragma unroll
        for (int db = 0; db < 8; ++db)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(V4, o[db]), part_rsrc, vo + db * 64, 0, 16);
        if (qd == 0) {
          const float2 ml = make_float2(m, l);
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.part_ml + ((int64_t)item * p.hq + hq) * 2),
                             __builtin_bit_cast(unsigned long long, ml), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if constexpr (kCombine) {
      if (!single) {
        // arrival counter of (request, kv head): every piece adds one after its partial sums are in memory; the piece
        // that draws nch - 1 is the last and combines all of them -- in piece order with the arithmetic of
        // attn_decode_merge_kernel, so the result does not depend on who arrives last -- reading with sc1 loads
        // (the producers stored write-through; this CU's L1 may hold stale lines of the workspace).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int* ticket = p.arrivals + (int64_t)b * kTicketHeads + h;
        int drawn = 0;
        if (lane == 0) drawn = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        drawn = sgpr(drawn);
        if (drawn == nch - 1) {
          if (lane == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
          if (tok < G) {
            const int hq = hq0 + tok;
            float mx = kNegBig;
            for (int j = 0; j < nch; ++j) {
              const unsigned long long u = __hip_atomic_load(
                  reinterpret_cast<unsigned long long*>(p.part_ml + ((int64_t)(first_item + j) * p.hq + hq) * 2),
                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              mx = fmaxf(mx, __builtin_bit_cast(float2, u).x);
            }
            f32x4 acc[8];
#pragma unroll
            for (int db = 0; db < 8; ++db) acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
            float den = 0.f;
            for (int j = 0; j < nch; ++j) {
              const int64_t base = (int64_t)(first_item + j) * p.hq + hq;
              const unsigned long long u = __hip_atomic_load(reinterpret_cast<unsigned long long*>(p.part_ml + base * 2),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const float2 ml = __builtin_bit_cast(float2, u);
              const float wgt = __builtin_amdgcn_exp2f(ml.x - mx);
              const int vo = (int)((base * D + 4 * qd) * 4);
#pragma unroll
              for (int db = 0; db < 8; ++db) {
                const f32x4 oj = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(part_rsrc, vo + db * 64, 0, 16));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[db][i] = fmaf(wgt, oj[i], acc[db][i]);
              }
              den = fmaf(wgt, ml.y, den);
            }
            const float inv = 1.0f / den;
            uint16_t* op = p.out + (int64_t)b * p.out_stride + (int64_t)hq * D + 4 * qd;
#pragma unroll
            for (int db = 0; db < 8; ++db) {
              uint2 u;
              u.x = Elem<T>::pack(acc[db][0] * inv, acc[db][1] * inv);
              u.y = Elem<T>::pack(acc[db][2] * inv, acc[db][3] * inv);
              *reinterpret_cast<uint2*>(op + db * 16) = u;
            }
          }
        }
      }
    }
    mark();  // 5 + 4 i: results stored (issued)
    it = nit;
    it2 = nit2;
    b = nb;
    row = nrow;
  }
  if constexpr (kTrace) {
    if (p.trace && lane == 0) {
      unsigned long long* tp = p.trace + (int64_t)gw * 16;
      for (int i = 0; i < 14; ++i) tp[i] = i < n_stamp ? stamp[i] : 0ull;
      tp[14] = (unsigned long long)n_stamp;
      tp[15] = __builtin_readcyclecounter();  // exit
    }
  }
}

// ------------------------------------------------------------------------------
// merge split-KV partials: one wave per (request, q head), 2 output dims per lane
// ------------------------------------------------------------------------------
// launched_slots: slots the attention launch in front of this one had waves for.  The plan is made (and its slot count
// fixed) by an earlier launch; if the kernel selection changed in between (msgl_attn_decode_select between a plan /
// a graph capture and a later launch) the grid can be short of the plan's slots, pieces are then never computed and
// this kernel would add uninitialised partial sums: it NaN-poisons every output row instead (loud, not plausible).
template <typename T>
__global__ __launch_bounds__(256) void attn_decode_merge_kernel(const DecodeParams p, int batch, int launched_slots) {
  constexpr int D = 128;
  const int lane = threadIdx.x & 63;
  const int w = sgpr((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
  if (w >= batch * p.hq) return;
  const int b = w / p.hq;
  const int hq = w - b * p.hq;
  if (p.plan[3] > launched_slots) {
    reinterpret_cast<uint32_t*>(p.out + (int64_t)b * p.out_stride + (int64_t)hq * D)[lane] = 0xffffffffu;
    return;
  }
  const int* item_start = p.plan + plan_off_item_start();
  const int* n_chunks = p.plan + plan_off_n_chunks(p.max_bs);
  const int n = sgpr(n_chunks[b]);
  if (n <= 1) return;  // single-piece requests were finished by the attention kernel
  const int i0 = sgpr(item_start[b]);
  // Same arithmetic in the same order as before (pieces in index order: acc = fma(w_j, o_j, acc)), but the loads no
  // longer form a chain of n dependent round trips: the pieces' (m, l) pairs are fetched by the LANES (one coalesced
  // load for up to 64 pieces), and the partial sums eight pieces at a time, the first eight before the (m, l) pairs are
  // back.  A single request at batch 1 is cut into ~14 pieces: 11.1 -> ~5 us per layer there.
  const float* po = p.part_o + ((int64_t)i0 * p.hq + hq) * D + lane * 2;
  const int64_t po_step = (int64_t)p.hq * D;
  float2 ov[8];
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (u < n) ov[u] = *reinterpret_cast<const float2*>(po + u * po_step);
  float acc0 = 0.f, acc1 = 0.f, den = 0.f;
  if (n <= 64) {
    float2 ml = make_float2(kNegBig, 0.f);
    if (lane < n) ml = *reinterpret_cast<const float2*>(p.part_ml + ((int64_t)(i0 + lane) * p.hq + hq) * 2);
    const float mx = wave_max(ml.x);
    const float wl = __builtin_amdgcn_exp2f(ml.x - mx);
    for (int j0 = 0; j0 < n; j0 += 8) {
      if (j0 > 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + u < n) ov[u] = *reinterpret_cast<const float2*>(po + (j0 + u) * po_step);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (j0 + u < n) {
          const float wgt = __shfl(wl, j0 + u, 64), lj = __shfl(ml.y, j0 + u, 64);
          acc0 = fmaf(wgt, ov[u].x, acc0);
          acc1 = fmaf(wgt, ov[u].y, acc1);
          den = fmaf(wgt, lj, den);
        }
      }
    }
  } else {  // more pieces than lanes (capacity-bound plans): the plain loop
    float mx = kNegBig;
    for (int j = 0; j < n; ++j) mx = fmaxf(mx, p.part_ml[((int64_t)(i0 + j) * p.hq + hq) * 2]);
    for (int j = 0; j < n; ++j) {
      const int64_t base = (int64_t)(i0 + j) * p.hq + hq;
      const float2 ml = *reinterpret_cast<const float2*>(p.part_ml + base * 2);
      const float wgt = __builtin_amdgcn_exp2f(ml.x - mx);
      const float2 o = *reinterpret_cast<const float2*>(p.part_o + base * D + lane * 2);
      acc0 = fmaf(wgt, o.x, acc0);
      acc1 = fmaf(wgt, o.y, acc1);
      den = fmaf(wgt, ml.y, den);
    }
  }
  const float inv = 1.0f / den;
  uint32_t* op = reinterpret_cast<uint32_t*>(p.out + (int64_t)b * p.out_stride + (int64_t)hq * D) + lane;
  *op = Elem<T>::pack(acc0 * inv, acc1 * inv);
}

// waves that can be resident at once for heads-per-unit G (drives both the plan and the grid)
template <typename T, int G>
static int resident_waves_of() {
  static int cached = 0;
  if (cached == 0) {
    constexpr int kThreads = 64 * DecodeGeom<G>::kWavesPerBlock;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, attn_decode_kernel<T, G, false>, kThreads, 0) !=
            hipSuccess ||
        nb <= 0) {
      (void)hipGetLastError();
      nb = 1;
    }
    const int cus = device_cu_count() > 0 ? device_cu_count() : 256;
    cached = nb * cus * DecodeGeom<G>::kWavesPerBlock;
  }
  return cached;
}

static int decode_impl();
static int mfma_variant(int G);

static int resident_waves_streaming(int G) {
  switch (G) {
    case 1: return resident_waves_of<BF16, 1>();
    case 2: return resident_waves_of<BF16, 2>();
    case 3: return resident_waves_of<BF16, 3>();
    case 4: return resident_waves_of<BF16, 4>();
    case 5: return resident_waves_of<BF16, 5>();
    case 6: return resident_waves_of<BF16, 6>();
    case 7: return resident_waves_of<BF16, 7>();
    default: return resident_waves_of<BF16, 8>();
  }
}

// the plan does not know the page size, so it is made for whichever of the two kernels keeps more waves resident
static int resident_waves(int G) {
  const int streaming = resident_waves_streaming(G);
  if (decode_impl() == 1) return streaming;
  const int cus = device_cu_count() > 0 ? device_cu_count() : 256;
  const int mfma = cus * 4 * (mfma_variant(G) / 10);  // attn_decode_mfma_kernel: that many waves per SIMD
  return mfma > streaming ? mfma : streaming;
}

// slots the plan may create: one per resident wave of a (virtual) kv head, bounded by the workspace
static int decode_target_slots(int G, int hv, int capacity, int max_bs) {
  int slots = resident_waves(G) / hv;
  if (slots > capacity - max_bs + 1) slots = capacity - max_bs + 1;
  return slots < 1 ? 1 : slots;
}

template <typename T, int G, bool kRun>
static int launch_decode_run(const DecodeParams& p, int batch, int capacity, hipStream_t s) {
  constexpr int kWPB = DecodeGeom<G>::kWavesPerBlock;
  const int64_t waves = (int64_t)decode_target_slots(G, p.hv, capacity, p.max_bs) * p.hv;
  const int64_t blocks = (waves + kWPB - 1) / kWPB;
  attn_decode_kernel<T, G, kRun><<<dim3((unsigned)blocks), dim3(64 * kWPB), 0, s>>>(p);
  const int64_t mblocks = ((int64_t)batch * p.hq + 3) / 4;
  attn_decode_merge_kernel<T><<<dim3((unsigned)mblocks), dim3(256), 0, s>>>(p, batch, (int)(blocks * kWPB / p.hv));
  return MSGL_OK;
}

// MSGL_DECODE_IMPL: 0 / unset = matrix-core kernel where it applies (pages of >= 16 tokens), 1 = streaming kernel only
static int g_decode_impl = -1;
static unsigned long long* g_decode_trace = nullptr;
static int decode_impl() {
  if (g_decode_impl < 0) g_decode_impl = getenv("MSGL_DECODE_IMPL") ? atoi(getenv("MSGL_DECODE_IMPL")) : 0;
  return g_decode_impl;
}
// matrix-core kernel variants: code = 10 * (waves per SIMD) + ring stages.  Two stages (deeper rings measured the same or
// slower in rounds 2-3: profiles/r02d_decode_ab.txt, and again in round 6 on the whole-line request shape -- 3 / 4 stages at
// two waves per SIMD and 3 / 4 / 6 stages at one: 146.5 ... 151.8 us against 147.6 at Qwen3-14B TP1, profiles/
// r06h_decode_ab_ring_depth_and_residency.txt; their instantiations are not kept) at the residency the plan is made
// for -- three waves per SIMD where the streaming kernel also has three (G <= 2), two otherwise.  92 = variant 22 without the
// products (diagnosis).
static int mfma_variant(int G) {
  const int c = decode_impl();
  if (c == 72) return 22;  // variant 22 with the in-kernel combine
  if (c == 71) return G <= 2 ? 32 : 22;  // the default variant, merge kernel forced
  if (c >= 10 && c < 60) return c;
  return G <= 2 && c < 92 ? 32 : 22;
}

template <typename T, int G>
static int launch_decode_mfma(const DecodeParams& p, int batch, int capacity, hipStream_t s) {
  // same number of waves as the plan was made for (decode_target_slots): any grid >= n_slots * hv is correct
  const int64_t waves = (int64_t)decode_target_slots(G, p.hv, capacity, p.max_bs) * p.hv;
  // select code 72: the last-arriving piece of a request combines the partial sums inside the kernel instead of the merge
  // launch (needs the arrival counters to cover the kv heads and 32-bit offsets into the partial sums).  Bit-identical
  // to the merge kernel.  NOT the default: round 4 measured it INSIDE the captured 256-sequence Qwen3-14B step (tools/
  // step_ab.py) at -45 / -25 / +-0 us per step on three boxes (one launch boundary per layer less), but stand-alone on the
  // TP-shard shapes it LOSES 3-6 us per layer (14B TP4 49.1 vs 45.8 us, 70B TP8 35.4 vs 29.6, Qwen3-0.6B 174.3 vs 171.8:
  // the combiner's dependent sc1 round trips at the very end of a short kernel; profiles/r04_decode_ab_with_combine_as_impl0.txt, r04_decode_ab_final.txt) and 8 us at
  // B = 32.  Select code 71 = the default variant by its number (A/B partner of 72).
  // Request shape.  Round 6 (tools/kv_stream_probe.hip, profiles/r06a_kv_stream_probe.txt): whole-line requests (4 token rows x
  // 256 B) under the nt policy stream the paged pool at 6.8 TB/s where round 5's 16 rows x 64 B reach 5.9 (and LOSE with nt:
  // nt lines bypass the L1, so the second 64-B half of a line is fetched again).  The price is K's trip through the wave's
  // LDS image; it pays where the stream is long -- token rows of >= 1 KB (>= 4 local kv heads): 14B TP1 171.7 -> 153.6 us,
  // Qwen3-0.6B 165.6 -> 145.1 -- and is neutral to -3 % on the short TP-shard launches (profiles/r06b_decode_ab.txt), which keep
  // the direct-to-operand shape.  Select 60 / 61 force the round-5 / the whole-line shape (A/B; same bits either way).
  const bool lines = decode_impl() == 61 || (decode_impl() != 60 && p.kv_stride_tok * 2 >= 1024);
  const bool combine = decode_impl() == 72 && p.hv <= kTicketHeads &&
                       (int64_t)capacity * p.hq * 128 * (int64_t)sizeof(float) < (1ll << 31);
  // token-granular page tables (slot_run < 16; the reference's default page_size = 1, P/engine/config.py:25): round 6 gives
  // them the matrix-core kernel too -- whole-line requests by per-lane addresses, one 16-byte table read per lane group and
  // tile -- instead of the round-1 streaming kernel
  const bool run = p.slot_run >= 16;
#define MSGL_MFMA_LAUNCH(STAGES, WAVES, MINW, ...)                                                    \
  attn_decode_mfma_kernel<T, STAGES, WAVES, MINW, __VA_ARGS__>                                        \
      <<<dim3((unsigned)((waves + WAVES - 1) / WAVES)), dim3(64 * WAVES), 0, s>>>(p)
#define MSGL_MFMA_VARIANT(STAGES, WAVES, MINW)                                                                 \
  do {                                                                                                         \
    if (!run && combine) MSGL_MFMA_LAUNCH(STAGES, WAVES, MINW, false, false, true, true, true, false);         \
    else if (!run) MSGL_MFMA_LAUNCH(STAGES, WAVES, MINW, false, false, false, true, true, false);              \
    else if (!lines && combine) MSGL_MFMA_LAUNCH(STAGES, WAVES, MINW, false, false, true, true, false);        \
    else if (!lines) MSGL_MFMA_LAUNCH(STAGES, WAVES, MINW, false, false, false, true, false);                  \
    else if (combine) MSGL_MFMA_LAUNCH(STAGES, WAVES, MINW, false, false, true);                               \
    else MSGL_MFMA_LAUNCH(STAGES, WAVES, MINW, false, false, false);                                           \
  } while (0)
  switch (run && decode_impl() >= 92 ? decode_impl() : mfma_variant(G)) {
    case 32: MSGL_MFMA_VARIANT(2, 4, 3); break;
    case 92:  // variant 22 without the products
      if (combine) MSGL_MFMA_LAUNCH(2, 8, 2, true, false, true);
      else MSGL_MFMA_LAUNCH(2, 8, 2, true, false, false);
      break;
    case 94:  // variant 22 with each piece's scalars fetched by the dependent chain request -> length, row, slots (A/B)
      MSGL_MFMA_LAUNCH(2, 8, 2, false, false, false, false);
      break;
    case 93:  // variant 22 with clock stamps per wave (msgl_attn_decode_trace)
      if (combine) MSGL_MFMA_LAUNCH(2, 8, 2, false, true, true);
      else MSGL_MFMA_LAUNCH(2, 8, 2, false, true, false);
      break;
    default: MSGL_MFMA_VARIANT(2, 8, 2); break;  // 22
  }
#undef MSGL_MFMA_VARIANT
#undef MSGL_MFMA_LAUNCH
  if (!combine) {
    const int64_t mblocks = ((int64_t)batch * p.hq + 3) / 4;
    attn_decode_merge_kernel<T><<<dim3((unsigned)mblocks), dim3(256), 0, s>>>(p, batch, (int)(waves / p.hv));
  }
  return MSGL_OK;
}

template <typename T, int G>
static int launch_decode(const DecodeParams& p, int batch, int capacity, hipStream_t s) {
  if (decode_impl() != 1) return launch_decode_mfma<T, G>(p, batch, capacity, s);
  return p.slot_run >= 16 ? launch_decode_run<T, G, true>(p, batch, capacity, s)
                          : launch_decode_run<T, G, false>(p, batch, capacity, s);
}

template <typename T>
static int dispatch_group(int G, const DecodeParams& p, int batch, int capacity, hipStream_t s) {
  switch (G) {
    case 1: return launch_decode<T, 1>(p, batch, capacity, s);
    case 2: return launch_decode<T, 2>(p, batch, capacity, s);
    case 3: return launch_decode<T, 3>(p, batch, capacity, s);
    case 4: return launch_decode<T, 4>(p, batch, capacity, s);
    case 5: return launch_decode<T, 5>(p, batch, capacity, s);
    case 6: return launch_decode<T, 6>(p, batch, capacity, s);
    case 7: return launch_decode<T, 7>(p, batch, capacity, s);
    case 8: return launch_decode<T, 8>(p, batch, capacity, s);
  }
  set_error("attn_decode: unsupported heads-per-unit %d", G);
  return MSGL_EINVAL;
}

// heads packed per work unit: the whole GQA group when it is <= 8, else its largest divisor <= 8
static int heads_per_unit(int group) {
  if (group <= 8) return group;
  for (int g = 8; g >= 1; --g)
    if (group % g == 0) return g;
  return 1;
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_attn_decode_select(int impl) {
  MSGL_REQUIRE(impl == 0 || impl == 1 || impl == 22 || impl == 32 || impl == 71 || impl == 72 || impl == 92 || impl == 93 || impl == 94 || impl == 60 || impl == 61,
               "attn_decode_select: impl %d (0 = default, 1 = streaming kernel only, 22 / 32 = matrix-core kernel with "
               "2 / 3 waves per SIMD (a two-stage request ring), 71 / 72 = the default variant with the merge kernel forced / with the "
               "in-kernel combine forced, 60 / 61 = the default variant with round 5's 16-row x 64-B requests / with whole-line nt requests forced)", impl);
  g_decode_impl = impl;
  return MSGL_OK;
}

extern "C" int msgl_attn_decode_trace(void* stamps) {
  g_decode_trace = static_cast<unsigned long long*>(stamps);
  return MSGL_OK;
}

extern "C" int64_t msgl_attn_decode_plan_words(int max_bs, int capacity) {
  if (max_bs < 1 || capacity < max_bs) return MSGL_EINVAL;
  return plan_off_arrivals(max_bs, capacity) + (int64_t)max_bs * kTicketHeads;
}

extern "C" int64_t msgl_attn_decode_workspace_bytes(int capacity, int num_q_heads, int head_dim) {
  if (capacity < 1 || num_q_heads < 1 || head_dim < 1) return MSGL_EINVAL;
  return (int64_t)capacity * num_q_heads * (head_dim + 2) * (int64_t)sizeof(float);
}

extern "C" int msgl_attn_decode_plan(int32_t* plan, const int32_t* seq_lens, int batch, int max_bs,
                                     int capacity, int num_q_heads, int num_kv_heads, int min_chunk,
                                     void* stream) {
  MSGL_REQUIRE(plan && seq_lens, "attn_decode_plan: null pointer");
  MSGL_REQUIRE((reinterpret_cast<uintptr_t>(plan) & 15u) == 0, "attn_decode_plan: plan must be 16-byte aligned");
  MSGL_REQUIRE(batch >= 1 && batch <= max_bs, "attn_decode_plan: batch %d outside [1, %d]", batch, max_bs);
  MSGL_REQUIRE(capacity >= max_bs, "attn_decode_plan: capacity %d < max_bs %d", capacity, max_bs);
  MSGL_REQUIRE(num_kv_heads >= 1 && num_q_heads >= num_kv_heads && num_q_heads % num_kv_heads == 0,
               "attn_decode_plan: %d q heads / %d kv heads", num_q_heads, num_kv_heads);
  if (min_chunk <= 0) min_chunk = 64;
  MSGL_REQUIRE(min_chunk % 16 == 0 && (min_chunk & (min_chunk - 1)) == 0,
               "attn_decode_plan: min_chunk %d must be a power of two >= 16", min_chunk);
  const int G = heads_per_unit(num_q_heads / num_kv_heads);
  const int target = decode_target_slots(G, num_q_heads / G, capacity, max_bs);
  decode_plan_kernel<<<dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
      plan, seq_lens, batch, max_bs, capacity, target, min_chunk / 16);
  MSGL_CHECK_LAUNCH("attn_decode_plan");
  return MSGL_OK;
}

extern "C" int msgl_attn_decode(void* out, const void* q, const void* k_cache, const void* v_cache,
                                const int32_t* page_table, int64_t pt_stride, const int32_t* req_rows,
                                const int32_t* seq_lens, const int32_t* plan, void* workspace, int batch,
                                int max_bs, int capacity, int num_q_heads, int num_kv_heads, int head_dim,
                                int64_t q_stride_tok, int64_t kv_stride_tok, int64_t kv_stride_head,
                                int64_t out_stride_tok, float sm_scale, int slot_run, int dtype,
                                void* stream) {
  MSGL_REQUIRE(out && q && k_cache && v_cache && page_table && seq_lens && plan && workspace,
               "attn_decode: null pointer");
  MSGL_REQUIRE(slot_run >= 1 && (slot_run & (slot_run - 1)) == 0, "attn_decode: slot_run %d must be a power of two",
               slot_run);
  MSGL_REQUIRE(batch >= 1 && batch <= max_bs && capacity >= max_bs, "attn_decode: bad batch/capacity");
  MSGL_REQUIRE(head_dim == 128, "attn_decode: head_dim %d unsupported (128 only)", head_dim);
  MSGL_REQUIRE(num_kv_heads >= 1 && num_q_heads % num_kv_heads == 0, "attn_decode: %d q heads / %d kv heads",
               num_q_heads, num_kv_heads);
  MSGL_REQUIRE(pt_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(page_table) & 15u) == 0,
               "attn_decode: page table rows must be 16-byte aligned");
  MSGL_REQUIRE(q_stride_tok % 8 == 0 && kv_stride_tok % 8 == 0 && kv_stride_head % 8 == 0 &&
                   out_stride_tok % 8 == 0,
               "attn_decode: strides must be multiples of 8 elements");
  MSGL_REQUIRE(aligned16(out) && aligned16(q) && aligned16(k_cache) && aligned16(v_cache) && aligned16(workspace) &&
                   aligned16(plan),
               "attn_decode: pointers must be 16-byte aligned");
  const int group = num_q_heads / num_kv_heads;
  const int G = heads_per_unit(group);
  DecodeParams p;
  p.q = (const uint16_t*)q;
  p.k = (const uint16_t*)k_cache;
  p.v = (const uint16_t*)v_cache;
  p.page_table = page_table;
  p.req_rows = req_rows;
  p.seq_lens = seq_lens;
  p.plan = plan;
  p.out = (uint16_t*)out;
  p.part_o = (float*)workspace;
  p.part_ml = p.part_o + (int64_t)capacity * num_q_heads * head_dim;
  p.pt_stride = pt_stride;
  p.q_stride = q_stride_tok;
  p.kv_stride_tok = kv_stride_tok;
  p.kv_stride_head = kv_stride_head;
  p.out_stride = out_stride_tok;
  p.max_bs = max_bs;
  p.capacity = capacity;
  p.hq = num_q_heads;
  p.hv = num_q_heads / G;
  p.group = group;
  p.slot_run = slot_run;
  p.scale_log2 = sm_scale * 1.4426950408889634f;
  p.trace = g_decode_trace;
  p.arrivals = const_cast<int*>(plan) + plan_off_arrivals(max_bs, capacity);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (dtype == MSGL_BF16) rc = dispatch_group<BF16>(G, p, batch, capacity, s);
  else if (dtype == MSGL_FP16) rc = dispatch_group<FP16>(G, p, batch, capacity, s);
  else {
    set_error("attn_decode: unsupported dtype code %d", dtype);
    return MSGL_EINVAL;
  }
  if (rc != MSGL_OK) return rc;
  MSGL_CHECK_LAUNCH("attn_decode");
  return MSGL_OK;
}
