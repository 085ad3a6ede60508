/*
 * oracle_bytes.c -- plain-C restatement of the reference's byte / integer kernels.
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): never linked into the product.
 *
 * Each function follows the device code of the reference one thread-of-work at a time:
 *   orc_store_kv         C/jit/store.cu:42-50   (one warp per token: copy K row, copy V row)
 *   orc_index            C/jit/index.cu:51-56   (out[i] = W[idx[i]])
 *   orc_index_masked     C/jit/index.cu:84-92   (pos = idx - start, UNSIGNED; zero fill)
 *   orc_compare_key      C/src/radix.cpp:19-40  (std::mismatch over the common length)
 *   orc_page_to_token    P/scheduler/cache.py:121-126 (page start -> page_size token slots)
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static int64_t load_idx(const void* idx, int is_i64, int64_t i) {
  return is_i64 ? ((const int64_t*)idx)[i] : (int64_t)((const int32_t*)idx)[i];
}

void orc_store_kv(uint8_t* k_cache, uint8_t* v_cache, const void* indices, int is_i64, const uint8_t* k,
                  const uint8_t* v, int64_t length, int64_t row_bytes, int64_t cache_stride,
                  int64_t input_stride) {
  for (int64_t w = 0; w < length; ++w) {
    const int64_t pos = load_idx(indices, is_i64, w);
    memcpy(k_cache + pos * cache_stride, k + w * input_stride, (size_t)row_bytes);
    memcpy(v_cache + pos * cache_stride, v + w * input_stride, (size_t)row_bytes);
  }
}

void orc_index(uint8_t* out, const uint8_t* weight, const void* indices, int is_i64, int64_t n,
               int64_t row_bytes) {
  for (int64_t i = 0; i < n; ++i)
    memcpy(out + i * row_bytes, weight + load_idx(indices, is_i64, i) * row_bytes, (size_t)row_bytes);
}

void orc_index_masked(uint8_t* out, const uint8_t* weight, const void* indices, int is_i64, int64_t n,
                      int64_t row_bytes, uint64_t start, uint64_t length) {
  for (int64_t i = 0; i < n; ++i) {
    /* the reference subtracts in the index type and compares as size_t */
    const uint64_t pos = (uint64_t)load_idx(indices, is_i64, i) - start;
    if (pos < length)
      memcpy(out + i * row_bytes, weight + pos * row_bytes, (size_t)row_bytes);
    else
      memset(out + i * row_bytes, 0, (size_t)row_bytes);
  }
}

int64_t orc_compare_key(const void* a, int64_t len_a, const void* b, int64_t len_b, int elem_bytes) {
  const int64_t n = len_a < len_b ? len_a : len_b;
  for (int64_t i = 0; i < n; ++i) {
    if (elem_bytes == 8) {
      if (((const int64_t*)a)[i] != ((const int64_t*)b)[i]) return i;
    } else {
      if (((const int32_t*)a)[i] != ((const int32_t*)b)[i]) return i;
    }
  }
  return n;
}

void orc_page_to_token(int32_t* out, const int32_t* pages, int64_t n_pages, int32_t page_size) {
  for (int64_t i = 0; i < n_pages; ++i)
    for (int32_t j = 0; j < page_size; ++j) out[i * page_size + j] = pages[i] + j;
}
