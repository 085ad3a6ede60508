// KV pool scatter-store and embedding gather: pure 16-byte-lane byte movers.
//
// Both are HBM-bound row copies.  One lane moves one 16-byte piece; lanes are laid
// out row-major over (row, piece) so a wave always touches whole contiguous row
// segments (a 2048-B K row = 2 waves, a 256-B TP8 row = 16 lanes => 4 rows/wave),
// i.e. small rows are packed several-per-wave instead of one-row-per-warp.
#include "common.h"

namespace msgl {

template <typename IdxT>
__global__ __launch_bounds__(256) void store_kv_kernel(char* __restrict__ k_cache,
                                                       char* __restrict__ v_cache,
                                                       const IdxT* __restrict__ indices,
                                                       const char* __restrict__ k,
                                                       const char* __restrict__ v, int64_t total,
                                                       int pieces_per_row, int64_t cache_stride,
                                                       int64_t k_stride, int64_t v_stride) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int64_t row = gid / pieces_per_row;
  const int64_t off = (gid - row * pieces_per_row) * 16;
  const int64_t slot = (int64_t)indices[row];
  const U4 kv = ldg16(k + row * k_stride + off);
  const U4 vv = ldg16(v + row * v_stride + off);
  stg16(k_cache + slot * cache_stride + off, kv);
  stg16(v_cache + slot * cache_stride + off, vv);
}

template <typename IdxT, bool kMasked>
__global__ __launch_bounds__(256) void gather_rows_kernel(char* __restrict__ out,
                                                          const char* __restrict__ weight,
                                                          const IdxT* __restrict__ indices,
                                                          int64_t total, int pieces_per_row,
                                                          int64_t row_bytes, uint64_t mask_start,
                                                          uint64_t mask_length) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int64_t row = gid / pieces_per_row;
  const int64_t off = (gid - row * pieces_per_row) * 16;
  U4 val = {0u, 0u, 0u, 0u};
  if constexpr (kMasked) {
    // unsigned wrap makes ids below `start` fail the range test (C/jit/index.cu:84-92)
    const uint64_t pos = (uint64_t)(int64_t)indices[row] - mask_start;
    if (pos < mask_length) val = ldg16(weight + pos * row_bytes + off);
  } else {
    val = ldg16(weight + (int64_t)indices[row] * row_bytes + off);
  }
  stg16(out + row * row_bytes + off, val);
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_store_kv(void* k_cache, void* v_cache, const void* indices, int indices_is_i64,
                             const void* k, const void* v, int64_t num_tokens, int64_t row_bytes,
                             int64_t cache_stride_bytes, int64_t k_stride_bytes,
                             int64_t v_stride_bytes, void* stream) {
  MSGL_REQUIRE(num_tokens >= 0, "store_kv: negative length");
  if (num_tokens == 0) return MSGL_OK;
  MSGL_REQUIRE(k_cache && v_cache && indices && k && v, "store_kv: null pointer");
  MSGL_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0, "store_kv: row bytes %lld not a multiple of 16",
               (long long)row_bytes);
  MSGL_REQUIRE(cache_stride_bytes % 16 == 0 && k_stride_bytes % 16 == 0 && v_stride_bytes % 16 == 0,
               "store_kv: strides must be multiples of 16 bytes");
  MSGL_REQUIRE(aligned16(k_cache) && aligned16(v_cache) && aligned16(k) && aligned16(v),
               "store_kv: base pointers must be 16-byte aligned");
  MSGL_REQUIRE(cache_stride_bytes >= row_bytes && k_stride_bytes >= row_bytes &&
                   v_stride_bytes >= row_bytes,
               "store_kv: stride smaller than the row");
  const int ppr = (int)(row_bytes / 16);
  const int64_t total = num_tokens * ppr;
  const int64_t blocks = (total + 255) / 256;
  MSGL_REQUIRE(blocks < (1ll << 31), "store_kv: too many rows");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (indices_is_i64) {
    store_kv_kernel<int64_t><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
        (char*)k_cache, (char*)v_cache, (const int64_t*)indices, (const char*)k, (const char*)v, total,
        ppr, cache_stride_bytes, k_stride_bytes, v_stride_bytes);
  } else {
    store_kv_kernel<int32_t><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
        (char*)k_cache, (char*)v_cache, (const int32_t*)indices, (const char*)k, (const char*)v, total,
        ppr, cache_stride_bytes, k_stride_bytes, v_stride_bytes);
  }
  MSGL_CHECK_LAUNCH("store_kv");
  return MSGL_OK;
}

extern "C" int msgl_embedding_gather(void* out, const void* weight, const void* indices,
                                     int indices_is_i64, int64_t num_indices, int64_t row_bytes,
                                     int has_mask, int64_t mask_start, int64_t mask_length,
                                     void* stream) {
  MSGL_REQUIRE(num_indices >= 0, "embedding_gather: negative length");
  if (num_indices == 0) return MSGL_OK;
  MSGL_REQUIRE(out && weight && indices, "embedding_gather: null pointer");
  MSGL_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0,
               "embedding_gather: row bytes %lld not a multiple of 16", (long long)row_bytes);
  MSGL_REQUIRE(aligned16(out) && aligned16(weight), "embedding_gather: pointers must be 16-byte aligned");
  MSGL_REQUIRE(!has_mask || (mask_start >= 0 && mask_length >= 0), "embedding_gather: bad vocab range");
  const int ppr = (int)(row_bytes / 16);
  const int64_t total = num_indices * ppr;
  const int64_t blocks = (total + 255) / 256;
  MSGL_REQUIRE(blocks < (1ll << 31), "embedding_gather: too many rows");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 g((unsigned)blocks), b(256);
#define MSGL_GATHER(IDX, MASKED)                                                                   \
  gather_rows_kernel<IDX, MASKED><<<g, b, 0, s>>>((char*)out, (const char*)weight,                 \
                                                  (const IDX*)indices, total, ppr, row_bytes,     \
                                                  (uint64_t)mask_start, (uint64_t)mask_length)
  if (indices_is_i64) {
    if (has_mask) MSGL_GATHER(int64_t, true); else MSGL_GATHER(int64_t, false);
  } else {
    if (has_mask) MSGL_GATHER(int32_t, true); else MSGL_GATHER(int32_t, false);
  }
#undef MSGL_GATHER
  MSGL_CHECK_LAUNCH("embedding_gather");
  return MSGL_OK;
}
