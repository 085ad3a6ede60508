// Peer-to-peer collectives over xGMI for decode-size messages.
//
// The reference stages messages <= max_bytes through a symmetric buffer (ncclMemAlloc +
// ncclCommWindowRegister(NCCL_WIN_COLL_SYMMETRIC), C/src/pynccl.cu:81-90, 105-123) so that NCCL's symmetric
// kernels can read the peers' copies directly.  The MI355X equivalent written out: every rank owns one buffer of
// fine-grained (uncached) device memory, all ranks map all buffers (hipIpc; xGMI peer access), and ONE kernel per
// collective does the whole exchange with direct loads from the peers:
//
//   one-shot  (small messages): copy in -> flag barrier -> every rank sums all peers' copies in rank order
//   two-shot  (larger):         copy in -> barrier -> rank r reduces chunk r from all peers into its own result
//                               area -> barrier -> every rank gathers the reduced chunks from their owners
//   all-gather:                 copy in -> barrier -> read every peer's copy
//
// xGMI is point to point (7 links per GPU): a ring all-reduce is bound by one link (2 (n-1)/n N / 153 GB/s), the
// two-shot exchange drives all links at once (~2 N / n per link); at the 0.5-4 MB messages of a decode step the rest
// is latency, which is one kernel and two or three flag round trips here.
//
// Rules: sums are taken in rank order 0..n-1 by every reader => all ranks hold identical bits (the replicated
// schedulers and samplers of the ranks must not diverge); flags are monotonically increasing sequence numbers kept
// in device memory (no host state changes per call => capturable in a hipGraph); a block synchronises only with
// the same block index on the peers, and the copy-in / reduce / gather partitions are chosen so that block b only
// ever reads what the peers' block b wrote; spin loops are bounded (error flag instead of a hung GPU).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "common.h"

namespace msgl {

constexpr int kP2PMaxRanks = 8;
constexpr int kP2PMaxBlocks = 64;
constexpr int kP2PThreads = 512;
constexpr int kP2PPhases = 3;
constexpr int kP2PHeaderBytes = 16384;
constexpr uint32_t kP2PSpinLimit = 40u * 1000u * 1000u;

struct P2PHeader {  // at offset 0 of every rank's buffer; flags are written by the peers
  uint32_t flag[kP2PMaxBlocks][kP2PPhases][kP2PMaxRanks];
  uint32_t seq[kP2PMaxBlocks][kP2PPhases];  // this rank's barrier counters (local)
  uint32_t error;
};
static_assert(sizeof(P2PHeader) <= kP2PHeaderBytes, "header");

struct P2PPeers {
  unsigned char* base[kP2PMaxRanks];  // mapped buffers, [rank]
};

// Returns true if this barrier -- or an earlier one of this communicator (`bad` in, the sticky error word) -- timed
// out.  The caller then POISONS what its block would have written (all-ones = NaN in bf16 / fp16): a collective
// whose barrier gave up must never hand back a plausible partial sum (a lagging or dead peer would otherwise turn
// into silently diverging replicas).  Once the error word is set no later barrier spins: every following collective
// poisons at once and the host raises at its next poll (kernel.P2PCommunicator.poll_error).
// Error word (sticky, 0 = healthy): bits 0-3 = 1 + barrier phase, 4-7 = collective kind (kP2PKind*), 8-15 = block index,
// 16-19 = the peer whose flag never came, bit 20 = set on the copies a failing rank writes into its PEERS' headers.
enum { kP2PKindOneShot = 1, kP2PKindTwoShot = 2, kP2PKindFusedNorm = 3, kP2PKindGather = 4 };

__device__ __forceinline__ bool p2p_barrier(const P2PPeers& peers, int rank, int world, int phase, bool release,
                                            uint32_t spin_limit, bool bad, int kind) {
  __shared__ int s_bad, s_gave_up;
  P2PHeader* self = reinterpret_cast<P2PHeader*>(peers.base[rank]);
  const int b = blockIdx.x;
  if (threadIdx.x == 0) { s_bad = bad ? 1 : 0; s_gave_up = 0; }
  __syncthreads();  // everything this block did before the barrier is issued
  if (release) __threadfence_system();
  uint32_t seq = 0;
  if (threadIdx.x < world) {
    seq = self->seq[b][phase] + 1;  // every lane < world reads the same value
    P2PHeader* peer = reinterpret_cast<P2PHeader*>(peers.base[threadIdx.x]);
    __hip_atomic_store(&peer->flag[b][phase][rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t spins = 0;
    while (!bad &&
           __hip_atomic_load(&self->flag[b][phase][threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      __builtin_amdgcn_s_sleep(2);
      ++spins;
      // a peer that gave up writes ITS verdict into every rank's header (below): a live rank stops waiting for flags that
      // will never come and poisons with it, instead of passing the barrier later and summing staging areas the failed
      // rank has meanwhile overwritten
      if ((spins & 1023u) == 0 &&
          __hip_atomic_load(&self->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
        s_bad = 1;
        break;
      }
      if (spins > spin_limit) {
        // first verdict sticks (a peer's told word may have landed meanwhile: it names the root cause, keep it)
        uint32_t expect = 0;
        __hip_atomic_compare_exchange_strong(
            &self->error, &expect,
            (1u + (uint32_t)phase) | ((uint32_t)kind << 4) | (((uint32_t)b & 255u) << 8) | (threadIdx.x << 16),
            __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_bad = 1;
        s_gave_up = 1;
        break;
      }
    }
  }
  __syncthreads();
  const bool out = s_bad != 0;
  if (s_gave_up != 0 && threadIdx.x < world && (int)threadIdx.x != rank) {
    // only a rank whose OWN wait ran out tells its peers (a rank that left its spin because a peer's word arrived would
    // otherwise overwrite the root cause at world >= 3), and it writes by compare-and-swap from 0 so that the first
    // verdict in a header sticks.  Sticky there as here: the peers' kernels poison from their next barrier / next launch
    // on, and their hosts raise at the next poll.
    P2PHeader* peer = reinterpret_cast<P2PHeader*>(peers.base[threadIdx.x]);
    uint32_t expect = 0;
    __hip_atomic_compare_exchange_strong(
        &peer->error, &expect,
        (1u + (uint32_t)phase) | ((uint32_t)kind << 4) | (((uint32_t)b & 255u) << 8) | ((uint32_t)rank << 16) | (1u << 20),
        __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) self->seq[b][phase] = seq;
  return out;
}

__device__ __forceinline__ bool p2p_sticky_error(const P2PPeers& peers, int rank) {
  return __hip_atomic_load(&reinterpret_cast<P2PHeader*>(peers.base[rank])->error, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM) != 0;
}

__device__ __forceinline__ U4 p2p_poison() { return U4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}; }

template <typename T>
__device__ __forceinline__ void acc8(float (&a)[8], const U4& v) {
  a[0] += Elem<T>::lo(v.x); a[1] += Elem<T>::hi(v.x); a[2] += Elem<T>::lo(v.y); a[3] += Elem<T>::hi(v.y);
  a[4] += Elem<T>::lo(v.z); a[5] += Elem<T>::hi(v.z); a[6] += Elem<T>::lo(v.w); a[7] += Elem<T>::hi(v.w);
}
template <typename T>
__device__ __forceinline__ U4 pack8f(const float (&a)[8]) {
  U4 u;
  u.x = Elem<T>::pack(a[0], a[1]); u.y = Elem<T>::pack(a[2], a[3]);
  u.z = Elem<T>::pack(a[4], a[5]); u.w = Elem<T>::pack(a[6], a[7]);
  return u;
}

// slice b of [0, n): contiguous, 16-byte packs
__device__ __forceinline__ void block_slice(int64_t n, int64_t& lo, int64_t& hi) {
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  lo = min((int64_t)blockIdx.x * per, n);
  hi = min(lo + per, n);
}

// data: in place, `packs` x 16 B.  area A (copy of the input) at base + header, area B (two-shot results) behind it.
template <typename T, bool TWO_SHOT>
__global__ __launch_bounds__(kP2PThreads) void p2p_all_reduce_kernel(P2PPeers peers, int rank, int world, U4* data,
                                                                     int64_t packs, int64_t area_packs,
                                                                     uint32_t spin_limit) {
  U4* mine = reinterpret_cast<U4*>(peers.base[rank] + kP2PHeaderBytes);
  bool bad = p2p_sticky_error(peers, rank);
  if constexpr (!TWO_SHOT) {
    int64_t lo, hi;
    block_slice(packs, lo, hi);
    for (int64_t i = lo + threadIdx.x; i < hi; i += kP2PThreads) mine[i] = data[i];
    bad = p2p_barrier(peers, rank, world, 0, true, spin_limit, bad, TWO_SHOT ? kP2PKindTwoShot : kP2PKindOneShot);
    for (int64_t i = lo + threadIdx.x; i < hi; i += kP2PThreads) {
      float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int r = 0; r < world; ++r) acc8<T>(a, reinterpret_cast<const U4*>(peers.base[r] + kP2PHeaderBytes)[i]);
      data[i] = bad ? p2p_poison() : pack8f<T>(a);
    }
    bad = p2p_barrier(peers, rank, world, 1, false, spin_limit, bad, TWO_SHOT ? kP2PKindTwoShot : kP2PKindOneShot);  // nobody refills its copy while a peer still reads it
    if (bad)
      for (int64_t i = lo + threadIdx.x; i < hi; i += kP2PThreads) data[i] = p2p_poison();
  } else {
    const int64_t chunk = (packs + world - 1) / world;  // packs per owner
    // copy in: for every chunk c this block's slice of c (what the peers' block b will read from me)
    for (int c = 0; c < world; ++c) {
      const int64_t c0 = min((int64_t)c * chunk, packs), cn = min(chunk, packs - c0);
      int64_t lo, hi;
      block_slice(cn, lo, hi);
      for (int64_t i = c0 + lo + threadIdx.x; i < c0 + hi; i += kP2PThreads) mine[i] = data[i];
    }
    bad = p2p_barrier(peers, rank, world, 0, true, spin_limit, bad, TWO_SHOT ? kP2PKindTwoShot : kP2PKindOneShot);
    // reduce-scatter: my chunk, summed in rank order, into my result area
    const int64_t m0 = min((int64_t)rank * chunk, packs), mn = min(chunk, packs - m0);
    int64_t lo, hi;
    block_slice(mn, lo, hi);
    U4* res = mine + area_packs;
    for (int64_t i = m0 + lo + threadIdx.x; i < m0 + hi; i += kP2PThreads) {
      float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int r = 0; r < world; ++r) acc8<T>(a, reinterpret_cast<const U4*>(peers.base[r] + kP2PHeaderBytes)[i]);
      res[i] = pack8f<T>(a);
    }
    bad = p2p_barrier(peers, rank, world, 1, true, spin_limit, bad, TWO_SHOT ? kP2PKindTwoShot : kP2PKindOneShot);
    // all-gather of the reduced chunks from their owners
    for (int c = 0; c < world; ++c) {
      const int64_t c0 = min((int64_t)c * chunk, packs), cn = min(chunk, packs - c0);
      int64_t l2, h2;
      block_slice(cn, l2, h2);
      const U4* src = reinterpret_cast<const U4*>(peers.base[c] + kP2PHeaderBytes) + area_packs;
      for (int64_t i = c0 + l2 + threadIdx.x; i < c0 + h2; i += kP2PThreads) data[i] = bad ? p2p_poison() : src[i];
    }
    bad = p2p_barrier(peers, rank, world, 2, false, spin_limit, bad, TWO_SHOT ? kP2PKindTwoShot : kP2PKindOneShot);
    if (bad)
      for (int c = 0; c < world; ++c) {
        const int64_t c0 = min((int64_t)c * chunk, packs), cn = min(chunk, packs - c0);
        int64_t l2, h2;
        block_slice(cn, l2, h2);
        for (int64_t i = c0 + l2 + threadIdx.x; i < c0 + h2; i += kP2PThreads) data[i] = p2p_poison();
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// all-reduce + residual add + RMSNorm in ONE kernel (decode-size row blocks): what the reference runs as
// `y = all_reduce(F.linear(x, w))` (P/layers/linear.py:102-106, 123-127) followed by RMSNormFused's
// fused_add_rmsnorm(y, residual, weight, eps) (P/layers/norm.py:33-38; P/models/qwen3.py:36-41).
//
// Two-shot exchange partitioned by ROWS: block b handles rows [b rpb, (b + 1) rpb) on EVERY rank (a block synchronises
// only with the same block index on the peers); row i is owned by rank i / per.  Copy my partial rows in -> barrier ->
// the owner sums its rows over the ranks in rank order and parks the sum, ROUNDED TO THE 16-BIT TYPE (= the
// all-reduce's output), in its result area -> barrier -> every block gathers its rows from their owners -- all
// requests in flight at once, one 16-byte piece of a row per thread -- and finishes them itself: + residual (fp32),
// residual <- rounded sum, x <- rmsnorm(fp32 sum) * weight.  The finishing arithmetic and its reduction order are
// those of rmsnorm_wide_row_kernel (csrc/norm_rope_act.hip: one piece per thread, wave sums added in wave order), so
// the result is bit-identical to all-reduce-then-norm, on every rank.  blockDim = dim / 8 rounded up to whole waves
// (<= 1024); up to kFusedRows rows per block, finished TOGETHER (one LDS exchange for all of them).  The block count
// stays that of the other collectives (a few dozen): every block pays three flag barriers with system-scope fences,
// and one block per row (256 of them) measured 70 us per call where this form costs ~10.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kFusedRows = 8;

template <typename T>
__global__ __launch_bounds__(1024) void p2p_all_reduce_add_rmsnorm_kernel(P2PPeers peers, int rank, int world, U4* x,
                                                                          U4* residual, const U4* __restrict__ weight,
                                                                          float eps, int rows, int ppr, int64_t xs,
                                                                          int64_t rs, int64_t area_packs,
                                                                          uint32_t spin_limit) {
  U4* mine = reinterpret_cast<U4*>(peers.base[rank] + kP2PHeaderBytes);
  bool bad = p2p_sticky_error(peers, rank);
  const int per = (rows + world - 1) / world;               // rows per owner
  const int rpb = (rows + (int)gridDim.x - 1) / (int)gridDim.x;  // rows per block (<= kFusedRows)
  const int r0 = min((int)blockIdx.x * rpb, rows), r1 = min(r0 + rpb, rows);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const bool has = tid < ppr;
  // 1. copy my partial rows in (x may be row-strided; the staging image is dense [rows][ppr])
  for (int r = r0; r < r1; ++r)
    for (int p = tid; p < ppr; p += nthr) mine[(int64_t)r * ppr + p] = x[(int64_t)r * xs + p];
  bad = p2p_barrier(peers, rank, world, 0, true, spin_limit, bad, kP2PKindFusedNorm);
  // 2. the rows of this block that I own: sum over the ranks in rank order, rounded -> my result area
  {
    U4* res = mine + area_packs;
    for (int r = r0; r < r1; ++r) {
      if (r / per != rank) continue;
      for (int p = tid; p < ppr; p += nthr) {
        const int64_t i = (int64_t)r * ppr + p;
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < world; ++k) acc8<T>(a, reinterpret_cast<const U4*>(peers.base[k] + kP2PHeaderBytes)[i]);
        res[i] = pack8f<T>(a);
      }
    }
  }
  bad = p2p_barrier(peers, rank, world, 1, true, spin_limit, bad, kP2PKindFusedNorm);
  // 3. gather this block's rows from their owners (all loads first), then finish them together
  __shared__ float lds_ss[kFusedRows][16];
  U4 sum[kFusedRows], resq[kFusedRows];
  const int n = r1 - r0;
#pragma unroll
  for (int k = 0; k < kFusedRows; ++k) {
    if (k < n && has) {
      const int r = r0 + k;
      const U4* src = reinterpret_cast<const U4*>(peers.base[r / per] + kP2PHeaderBytes) + area_packs;
      sum[k] = src[(int64_t)r * ppr + tid];
      resq[k] = residual[(int64_t)r * rs + tid];
    }
  }
  U4 wq = U4{0, 0, 0, 0};
  if (has) wq = weight[tid];
  const int w = tid >> 6, nw = (nthr + 63) >> 6;
  auto row_values = [&](int k, float (&v)[8]) {  // fp32 (all-reduced x, rounded) + residual of this thread's piece
    float rr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    acc8<T>(v, sum[k]);
    acc8<T>(rr, resq[k]);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += rr[e];
  };
#pragma unroll
  for (int k = 0; k < kFusedRows; ++k) {
    if (k >= n) break;
    float ss = 0.f;
    if (has) {
      float v[8];
      row_values(k, v);
      residual[(int64_t)(r0 + k) * rs + tid] = bad ? p2p_poison() : pack8f<T>(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
    }
    ss = wave_sum(ss);  // common.h: the association rmsnorm_wide_row_kernel uses
    if ((tid & 63) == 0) lds_ss[k][w] = ss;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kFusedRows; ++k) {
    if (k >= n) break;
    float tot = 0.f;
    for (int i = 0; i < nw; ++i) tot += lds_ss[k][i];
    const float inv = rsqrtf(tot / (float)(ppr * 8) + eps);
    if (has) {
      float v[8], wf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, y[8];
      row_values(k, v);
      acc8<T>(wf, wq);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = __fmul_rn(__fmul_rn(v[e], inv), wf[e]);
      x[(int64_t)(r0 + k) * xs + tid] = bad ? p2p_poison() : pack8f<T>(y);
    }
  }
  bad = p2p_barrier(peers, rank, world, 2, false, spin_limit, bad, kP2PKindFusedNorm);  // nobody refills its areas while a peer still reads
  if (bad && has)
    for (int r = r0; r < r1; ++r) {
      x[(int64_t)r * xs + tid] = p2p_poison();
      residual[(int64_t)r * rs + tid] = p2p_poison();
    }
}

// dst[r * packs + i] = src_of_rank_r[i]
__global__ __launch_bounds__(kP2PThreads) void p2p_all_gather_kernel(P2PPeers peers, int rank, int world, U4* dst,
                                                                     const U4* src, int64_t packs,
                                                                     uint32_t spin_limit) {
  U4* mine = reinterpret_cast<U4*>(peers.base[rank] + kP2PHeaderBytes);
  bool bad = p2p_sticky_error(peers, rank);
  int64_t lo, hi;
  block_slice(packs, lo, hi);
  for (int64_t i = lo + threadIdx.x; i < hi; i += kP2PThreads) mine[i] = src[i];
  bad = p2p_barrier(peers, rank, world, 0, true, spin_limit, bad, kP2PKindGather);
  for (int r = 0; r < world; ++r) {
    const U4* from = reinterpret_cast<const U4*>(peers.base[r] + kP2PHeaderBytes);
    for (int64_t i = lo + threadIdx.x; i < hi; i += kP2PThreads) dst[(int64_t)r * packs + i] = bad ? p2p_poison() : from[i];
  }
  bad = p2p_barrier(peers, rank, world, 1, false, spin_limit, bad, kP2PKindGather);
  if (bad)
    for (int r = 0; r < world; ++r)
      for (int64_t i = lo + threadIdx.x; i < hi; i += kP2PThreads) dst[(int64_t)r * packs + i] = p2p_poison();
}

}  // namespace msgl

using namespace msgl;

struct msgl_p2p {
  int rank = 0, world = 1;
  size_t max_bytes = 0;        // largest message (each of the two data areas holds this much)
  size_t total_bytes = 0;
  void* local = nullptr;       // this rank's buffer
  void* mapped[kP2PMaxRanks] = {nullptr};
  bool opened[kP2PMaxRanks] = {false};
  hipIpcMemHandle_t handle;
  P2PPeers peers;
  size_t one_shot_max = 256 << 10;
  int blocks = 32;
  uint32_t spin_limit = kP2PSpinLimit;  // polls of a peer's flag before a barrier gives up (msgl_p2p_set_spin_limit)
};

#define P2P_HIP(call, what)                                                     \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) {                                                     \
      ::msgl::set_error("%s: %s", what, hipGetErrorString(e_));                 \
      (void)hipGetLastError();                                                  \
      return MSGL_ELAUNCH;                                                      \
    }                                                                           \
  } while (0)

extern "C" int msgl_p2p_create(msgl_p2p_t* out, int rank, int world_size, size_t max_bytes) {
  MSGL_REQUIRE(out, "p2p_create: null pointer");
  MSGL_REQUIRE(world_size >= 1 && world_size <= kP2PMaxRanks && rank >= 0 && rank < world_size,
               "p2p_create: rank %d of %d (at most %d ranks)", rank, world_size, kP2PMaxRanks);
  MSGL_REQUIRE(max_bytes >= 16 && max_bytes <= ((size_t)1 << 31), "p2p_create: max_bytes %zu", max_bytes);
  msgl_p2p* c = new msgl_p2p();
  c->rank = rank;
  c->world = world_size;
  c->max_bytes = (max_bytes + 4095) / 4096 * 4096;
  c->total_bytes = kP2PHeaderBytes + 2 * c->max_bytes;
  // fine-grained device memory: peers' loads and this rank's flag polls see stores without cache maintenance
  hipError_t e = hipExtMallocWithFlags(&c->local, c->total_bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    set_error("p2p_create: hipExtMallocWithFlags(%zu, uncached): %s", c->total_bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    delete c;
    return MSGL_ELAUNCH;
  }
  if (hipMemset(c->local, 0, c->total_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
      hipIpcGetMemHandle(&c->handle, c->local) != hipSuccess) {
    set_error("p2p_create: memset / ipc handle: %s", hipGetErrorString(hipGetLastError()));
    (void)hipFree(c->local);
    delete c;
    return MSGL_ELAUNCH;
  }
  for (int r = 0; r < kP2PMaxRanks; ++r) c->peers.base[r] = nullptr;
  c->mapped[rank] = c->local;
  c->peers.base[rank] = static_cast<unsigned char*>(c->local);
  *out = c;
  return MSGL_OK;
}

extern "C" int msgl_p2p_ipc_handle(msgl_p2p_t c, void* out_handle) {
  MSGL_REQUIRE(c && out_handle, "p2p_ipc_handle: null pointer");
  memcpy(out_handle, &c->handle, sizeof(hipIpcMemHandle_t));
  return MSGL_OK;
}

// all_handles: world x MSGL_IPC_HANDLE_BYTES, rank order (gathered by the host over its CPU group)
extern "C" int msgl_p2p_open(msgl_p2p_t c, const void* all_handles) {
  MSGL_REQUIRE(c && all_handles, "p2p_open: null pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == MSGL_IPC_HANDLE_BYTES, "ipc handle size");
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, static_cast<const char*>(all_handles) + (size_t)r * MSGL_IPC_HANDLE_BYTES, sizeof(h));
    void* p = nullptr;
    P2P_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "p2p_open: hipIpcOpenMemHandle");
    c->mapped[r] = p;
    c->opened[r] = true;
    c->peers.base[r] = static_cast<unsigned char*>(p);
  }
  return MSGL_OK;
}

extern "C" int msgl_p2p_configure(msgl_p2p_t c, size_t one_shot_max_bytes, int blocks) {
  MSGL_REQUIRE(c, "p2p_configure: null pointer");
  MSGL_REQUIRE(blocks >= 1 && blocks <= kP2PMaxBlocks, "p2p_configure: blocks %d (1..%d)", blocks, kP2PMaxBlocks);
  c->one_shot_max = one_shot_max_bytes;
  c->blocks = blocks;
  return MSGL_OK;
}

static int p2p_ready(msgl_p2p_t c, const char* what) {
  for (int r = 0; r < c->world; ++r)
    if (!c->peers.base[r]) {
      set_error("%s: peer %d is not mapped (msgl_p2p_open not called)", what, r);
      return MSGL_EINVAL;
    }
  return MSGL_OK;
}

extern "C" int msgl_p2p_all_reduce_sum(msgl_p2p_t c, void* data, size_t count, int dtype, void* stream) {
  MSGL_REQUIRE(c && data, "p2p_all_reduce: null pointer");
  MSGL_REQUIRE(dtype == MSGL_BF16 || dtype == MSGL_FP16, "p2p_all_reduce: dtype code %d unsupported", dtype);
  if (count == 0) return MSGL_OK;
  const size_t bytes = count * 2;
  MSGL_REQUIRE(bytes % 16 == 0 && aligned16(data), "p2p_all_reduce: %zu bytes must be a multiple of 16, 16-byte aligned",
               bytes);
  MSGL_REQUIRE(bytes <= c->max_bytes, "p2p_all_reduce: %zu bytes exceed the buffer (%zu)", bytes, c->max_bytes);
  if (int rc = p2p_ready(c, "p2p_all_reduce")) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t packs = (int64_t)(bytes / 16), area = (int64_t)(c->max_bytes / 16);
  const bool two = bytes > c->one_shot_max && c->world > 1;
  const dim3 grid((unsigned)c->blocks), block(kP2PThreads);
#define MSGL_P2P(T, TWO) p2p_all_reduce_kernel<T, TWO><<<grid, block, 0, s>>>(c->peers, c->rank, c->world, (U4*)data, packs, area, c->spin_limit)
  if (dtype == MSGL_BF16) { if (two) MSGL_P2P(BF16, true); else MSGL_P2P(BF16, false); }
  else { if (two) MSGL_P2P(FP16, true); else MSGL_P2P(FP16, false); }
#undef MSGL_P2P
  MSGL_CHECK_LAUNCH("p2p_all_reduce");
  return MSGL_OK;
}

// x [rows, dim] (row stride x_stride elements) <- rmsnorm(sum over ranks of x + residual) * weight, residual <- the rounded
// sum, as ONE launch; every rank calls it with its own partial x and the same residual.  Returns MSGL_EINVAL (with a
// message) for what the kernel does not cover -- the caller then runs msgl_p2p_all_reduce_sum + msgl_fused_add_rmsnorm.
extern "C" int msgl_p2p_all_reduce_add_rmsnorm(msgl_p2p_t c, void* x, void* residual, const void* weight, float eps,
                                               int64_t rows, int64_t dim, int64_t x_stride, int64_t res_stride, int dtype,
                                               void* stream) {
  MSGL_REQUIRE(c && x && residual && weight, "p2p_all_reduce_add_rmsnorm: null pointer");
  MSGL_REQUIRE(dtype == MSGL_BF16 || dtype == MSGL_FP16, "p2p_all_reduce_add_rmsnorm: dtype code %d unsupported", dtype);
  MSGL_REQUIRE(rows >= 1 && dim >= 8 && dim % 8 == 0 && dim <= 8192, "p2p_all_reduce_add_rmsnorm: %lld rows x %lld",
               (long long)rows, (long long)dim);
  MSGL_REQUIRE(x_stride % 8 == 0 && res_stride % 8 == 0 && x_stride >= dim && res_stride >= dim && aligned16(x) &&
                   aligned16(residual) && aligned16(weight),
               "p2p_all_reduce_add_rmsnorm: strides / alignment");
  const size_t bytes = (size_t)rows * (size_t)dim * 2;
  MSGL_REQUIRE(bytes <= c->max_bytes, "p2p_all_reduce_add_rmsnorm: %zu bytes exceed the buffer (%zu)", bytes, c->max_bytes);
  if (int rc = p2p_ready(c, "p2p_all_reduce_add_rmsnorm")) return rc;
  // the communicator's block count (that of the other collectives) while kFusedRows rows per block suffice, more up to
  // the flag rows of the header
  MSGL_REQUIRE(rows <= (int64_t)kP2PMaxBlocks * kFusedRows, "p2p_all_reduce_add_rmsnorm: %lld rows (at most %d)",
               (long long)rows, kP2PMaxBlocks * kFusedRows);
  int blocks = c->blocks;
  while ((rows + blocks - 1) / blocks > kFusedRows) ++blocks;
  const int rpb = (int)((rows + blocks - 1) / blocks);
  blocks = (int)((rows + rpb - 1) / rpb);
  const int ppr = (int)(dim / 8);
  const unsigned threads = (unsigned)((ppr + 63) / 64 * 64);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t area = (int64_t)(c->max_bytes / 16);
  if (dtype == MSGL_BF16)
    p2p_all_reduce_add_rmsnorm_kernel<BF16><<<dim3((unsigned)blocks), dim3(threads), 0, s>>>(
        c->peers, c->rank, c->world, (U4*)x, (U4*)residual, (const U4*)weight, eps, (int)rows, ppr, x_stride / 8,
        res_stride / 8, area, c->spin_limit);
  else
    p2p_all_reduce_add_rmsnorm_kernel<FP16><<<dim3((unsigned)blocks), dim3(threads), 0, s>>>(
        c->peers, c->rank, c->world, (U4*)x, (U4*)residual, (const U4*)weight, eps, (int)rows, ppr, x_stride / 8,
        res_stride / 8, area, c->spin_limit);
  MSGL_CHECK_LAUNCH("p2p_all_reduce_add_rmsnorm");
  return MSGL_OK;
}

extern "C" int msgl_p2p_all_gather(msgl_p2p_t c, void* dst, const void* src, size_t count, int dtype, void* stream) {
  MSGL_REQUIRE(c && dst && src, "p2p_all_gather: null pointer");
  MSGL_REQUIRE(dtype == MSGL_BF16 || dtype == MSGL_FP16, "p2p_all_gather: dtype code %d unsupported", dtype);
  if (count == 0) return MSGL_OK;
  const size_t bytes = count * 2;
  MSGL_REQUIRE(bytes % 16 == 0 && aligned16(dst) && aligned16(src), "p2p_all_gather: %zu bytes per rank must be a "
               "multiple of 16, 16-byte aligned", bytes);
  MSGL_REQUIRE(bytes <= c->max_bytes, "p2p_all_gather: %zu bytes exceed the buffer (%zu)", bytes, c->max_bytes);
  if (int rc = p2p_ready(c, "p2p_all_gather")) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  p2p_all_gather_kernel<<<dim3((unsigned)c->blocks), dim3(kP2PThreads), 0, s>>>(c->peers, c->rank, c->world, (U4*)dst,
                                                                              (const U4*)src, (int64_t)(bytes / 16),
                                                                              c->spin_limit);
  MSGL_CHECK_LAUNCH("p2p_all_gather");
  return MSGL_OK;
}

// 0 = no barrier ever timed out; otherwise 1 + the phase that did (sticky).  Synchronises the device.
extern "C" int msgl_p2p_error(msgl_p2p_t c) {
  MSGL_REQUIRE(c, "p2p_error: null pointer");
  uint32_t e = 0;
  P2P_HIP(hipMemcpy(&e, &reinterpret_cast<P2PHeader*>(c->local)->error, sizeof(e), hipMemcpyDeviceToHost),
          "p2p_error: read");
  return (int)e;
}

// The same word without a device synchronisation: a 4-byte device-to-host copy enqueued on `stream` into
// `host_dst` (pinned host memory owned by the caller, read by it after a later synchronisation of that stream).
extern "C" int msgl_p2p_error_async(msgl_p2p_t c, void* host_dst, void* stream) {
  MSGL_REQUIRE(c && host_dst, "p2p_error_async: null pointer");
  P2P_HIP(hipMemcpyAsync(host_dst, &reinterpret_cast<P2PHeader*>(c->local)->error, sizeof(uint32_t),
                         hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)),
          "p2p_error_async: copy");
  return MSGL_OK;
}

// Polls of a peer's flag (each ~ s_sleep 2 + one uncached load) before a barrier gives up; default 40 M (tens of
// seconds: a peer may legitimately lag by a capture or a tuning pass).  Tests lower it.
extern "C" int msgl_p2p_set_spin_limit(msgl_p2p_t c, uint32_t spins) {
  MSGL_REQUIRE(c && spins > 0, "p2p_set_spin_limit: bad argument");
  c->spin_limit = spins;
  return MSGL_OK;
}

extern "C" void* msgl_p2p_get_buffer(msgl_p2p_t c) {
  return c ? static_cast<unsigned char*>(c->local) + kP2PHeaderBytes : nullptr;
}

// Buffers of destroyed communicators.  They are NOT unmapped / freed while the process lives: on this stack
// (ROCm 7.0 runtime under PyTorch, dmabuf IPC) the first allocations made after hipIpcCloseMemHandle + hipFree of an
// exported buffer were observed to read and write garbage (an engine built right after the communicator was destroyed
// produced NaNs; the same engine built before, or with the release postponed, is bit-exact -- tests/test_gpu_tp.py).
// A communicator lives as long as its engine and engines live as long as their process (the reference destroys its
// NCCL wrapper at shutdown only, P/engine/engine.py:208-211), so parking a few MB until exit costs nothing.
static std::vector<msgl_p2p*>& p2p_graveyard() {
  static std::vector<msgl_p2p*>* g = new std::vector<msgl_p2p*>();
  return *g;
}

extern "C" int msgl_p2p_destroy(msgl_p2p_t c) {
  if (!c) return MSGL_OK;
  (void)hipDeviceSynchronize();  // nothing of this communicator is in flight any more
  for (int r = 0; r < kP2PMaxRanks; ++r) c->peers.base[r] = nullptr;  // any further collective is refused
  p2p_graveyard().push_back(c);
  return MSGL_OK;
}

// Unmap and free everything destroyed so far (call only when no further device allocation will be made, e.g.
// right before process exit; never required).
extern "C" int msgl_p2p_release_all(void) {
  (void)hipDeviceSynchronize();
  for (msgl_p2p* c : p2p_graveyard()) {
    for (int r = 0; r < c->world; ++r)
      if (c->opened[r] && c->mapped[r]) (void)hipIpcCloseMemHandle(c->mapped[r]);
    if (c->local) (void)hipFree(c->local);
    delete c;
  }
  p2p_graveyard().clear();
  return MSGL_OK;
}
