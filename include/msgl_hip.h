/*
 * msgl_hip.h -- C-ABI of the MI355X (gfx950) paged-attention serving core.
 *
 * This header is the drop-in boundary.  Every entry point replaces one native
 * interface the reference (sgl-project/mini-sglang) reaches through tvm-ffi or
 * through its third-party CUDA dependencies (flashinfer / sgl_kernel).  The
 * reference file:line each function stands in for is cited next to it; paths
 * are relative to the reference root, `P/` = python/minisgl/, `C/` =
 * python/minisgl/kernel/csrc/.
 *
 * Conventions (mirrors the reference FFI contract):
 *   - plain pointers + sizes + explicit strides, no torch / DLPack types;
 *   - all device pointers are raw HIP device addresses; `stream` is a
 *     hipStream_t passed as void* (the caller's current stream, cf.
 *     C/include/minisgl/utils.cuh:103-106);
 *   - every function returns 0 on success, a negative MSGL_E* code otherwise,
 *     and records a message retrievable with msgl_last_error() (the reference
 *     throws host::PanicError, C/include/minisgl/utils.h:40-87);
 *   - callee never allocates, frees or synchronises: every call is legal
 *     inside hipGraph stream capture;
 *   - strides are in ELEMENTS of the tensor dtype unless the name says bytes.
 *
 * Three shared objects implement it:
 *   libmsgl_hip.so   every hand-written kernel + host helpers (no library dependency)
 *   libmsgl_comm.so  msgl_comm_* (links librccl)
 *   libmsgl_gemm.so  msgl_gemm_* (links libhipblaslt)
 */
#ifndef MSGL_HIP_H_
#define MSGL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSGL_OK 0
#define MSGL_EINVAL (-1)   /* argument check failed (shape/stride/alignment/dtype) */
#define MSGL_ELAUNCH (-2)  /* HIP reported a launch error                          */
#define MSGL_ECOMM (-3)    /* RCCL reported an error                               */

/* dtype codes for 16-bit floating tensors */
#define MSGL_BF16 0
#define MSGL_FP16 1
/* dtype codes for logits */
#define MSGL_F32 2

#define MSGL_ABI_VERSION 5

/* Last error message of the calling thread ("" if none). */
const char* msgl_last_error(void);
int msgl_abi_version(void);
/* Number of compute units of the current device (0 if no device). */
int msgl_device_cu_count(void);

/* ------------------------------------------------------------------------
 * KV pool scatter-store.   Replaces: store_cache (P/kernel/store.py:30-42),
 * StoreKernel::run / store_kv_cache (C/jit/store.cu:27-121).
 *   k_cache[indices[w]] = k[w];  v_cache[indices[w]] = v[w]   (byte copy)
 * row_bytes must be a multiple of 16; all bases/strides 16-byte aligned.
 * indices are token slots (int32 or int64).
 * ---------------------------------------------------------------------- */
int msgl_store_kv(void* k_cache, void* v_cache, const void* indices, int indices_is_i64,
                  const void* k, const void* v, int64_t num_tokens, int64_t row_bytes,
                  int64_t cache_stride_bytes, int64_t k_stride_bytes, int64_t v_stride_bytes,
                  void* stream);

/* ------------------------------------------------------------------------
 * Embedding gather.  Replaces: indexing (P/kernel/index.py:30-50),
 * index_kernel / masked_index_kernel (C/jit/index.cu:33-96).
 *   out[i] = W[idx[i]]                       (has_mask == 0)
 *   p = idx[i] - start; out[i] = p < length (unsigned) ? W[p] : 0
 * ---------------------------------------------------------------------- */
int msgl_embedding_gather(void* out, const void* weight, const void* indices, int indices_is_i64,
                          int64_t num_indices, int64_t row_bytes, int has_mask, int64_t mask_start,
                          int64_t mask_length, void* stream);

/* ------------------------------------------------------------------------
 * Radix-tree key compare (host).  Replaces: fast_compare_key
 * (C/src/radix.cpp:19-40, P/kernel/radix.py:14-20).  Returns the index of
 * the first mismatch of two int32/int64 host arrays (>= 0), or MSGL_EINVAL.
 * ---------------------------------------------------------------------- */
int64_t msgl_fast_compare_key(const void* a, int64_t len_a, const void* b, int64_t len_b,
                              int elem_bytes);

/* ------------------------------------------------------------------------
 * RMSNorm family.  Replaces: flashinfer.rmsnorm / fused_add_rmsnorm as called
 * at P/layers/norm.py:17,20,36-37 (and the strided 3-D in-place calls of
 * P/layers/attention.py:50-53).
 * The logical input is [n0, n1, dim] with element strides (stride0, stride1, 1);
 * a 2-D [T, H] call passes n1 = 1.  out may alias x.  fp32 math.
 * ---------------------------------------------------------------------- */
int msgl_rmsnorm(void* out, const void* x, const void* weight, float eps, int64_t n0, int64_t n1,
                 int64_t dim, int64_t x_stride0, int64_t x_stride1, int64_t out_stride0,
                 int64_t out_stride1, int dtype, void* stream);
/* residual <- x + residual (rounded to dtype);  x <- rmsnorm(fp32 sum) * w */
/* fused_add_rmsnorm whose x is still a split-K projection's partial sums: x_row = round16(sum_s slabs[s]) in slab
 * order (bit-identical to reducing first), then exactly msgl_fused_add_rmsnorm.  x receives the normalised rows. */
int msgl_fused_add_rmsnorm_slabs(void* x, void* residual, const void* weight, float eps, int64_t rows,
                                 int64_t dim, int64_t x_stride, int64_t res_stride, const float* slabs,
                                 int num_slabs, int64_t slab_stride, int64_t slab_ld, int dtype, void* stream);
int msgl_fused_add_rmsnorm(void* x, void* residual, const void* weight, float eps, int64_t rows,
                           int64_t dim, int64_t x_stride, int64_t res_stride, int dtype,
                           void* stream);

/* ------------------------------------------------------------------------
 * NeoX RoPE in place.  Replaces: flashinfer.apply_rope_with_cos_sin_cache_inplace
 * at P/layers/rotary.py:45-51.  cos_sin_cache is fp32 [max_pos, head_dim]
 * = cat(cos, sin) (P/layers/rotary.py:24-32).  q: [T, Hq*D], k: [T, Hk*D]
 * with row strides (elements).
 * ---------------------------------------------------------------------- */
int msgl_rope_neox_inplace(void* q, void* k, const void* positions, int positions_is_i64,
                           const float* cos_sin_cache, int64_t num_tokens, int num_q_heads,
                           int num_k_heads, int head_dim, int64_t q_stride, int64_t k_stride,
                           int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Fused per-token pass over the qkv row:
 *   [q-norm, k-norm (optional)] -> NeoX RoPE (in place on q,k) -> scatter k,v
 * rows to the KV pool at out_loc.  Equivalent to the op sequence of
 * P/layers/attention.py:47-56 + P/kvcache/mha_pool.py:45-56.
 * q_norm_w / k_norm_w may be NULL (no qk-norm, e.g. Llama).
 * ---------------------------------------------------------------------- */
int msgl_qk_norm_rope_store(void* q, void* k, const void* v, const void* q_norm_w,
                            const void* k_norm_w, float eps, const void* positions,
                            int positions_is_i64, const float* cos_sin_cache, void* k_cache,
                            void* v_cache, const void* out_loc, int out_loc_is_i64,
                            int64_t num_tokens, int num_q_heads, int num_k_heads, int head_dim,
                            int64_t q_stride, int64_t k_stride, int64_t v_stride,
                            int64_t cache_stride, int dtype, void* stream);
/* The same pass for a qkv row that is still a split-K projection's partial sums (msgl_m256_gemm_slabs_nt):
 * element = round16(sum_s slabs[s]) in slab order, then exactly msgl_qk_norm_rope_store on the [num_tokens,
 * (Hq + 2 Hk) D] buffer `qkv` (q | k | v column blocks, row stride qkv_stride), which receives q, k AND v, i.e. ends
 * up bit-identical to reduce-then-msgl_qk_norm_rope_store.  One launch less per layer. */
int msgl_qk_norm_rope_store_slabs(void* qkv, int64_t qkv_stride, const float* slabs, int num_slabs,
                                  int64_t slab_stride, int64_t slab_ld, const void* q_norm_w,
                                  const void* k_norm_w, float eps, const void* positions,
                                  int positions_is_i64, const float* cos_sin_cache, void* k_cache,
                                  void* v_cache, const void* out_loc, int out_loc_is_i64,
                                  int64_t num_tokens, int num_q_heads, int num_k_heads, int head_dim,
                                  int64_t cache_stride, int dtype, void* stream);

/* out[t, j] = silu(x[t, j]) * x[t, d + j].  Replaces flashinfer.silu_and_mul
 * (P/layers/activation.py:9-12). */
int msgl_silu_and_mul(void* out, const void* x, int64_t num_tokens, int64_t d, int64_t x_stride,
                      int64_t out_stride, int dtype, void* stream);

/* The same activation for a gate_up row whose columns are interleaved in blocks of 32 (gate, up, gate, up per 128
 * columns: the weight layout msgl_g3_gemm_nt's fused epilogue needs, applied once at load time to the rows of
 * gate_up_proj.weight): out[t, 64 c + 32 q + r] = silu(x[t, 128 c + 64 q + r]) * x[t, 128 c + 64 q + 32 + r].
 * d % 64 == 0. */
int msgl_silu_and_mul_interleaved(void* out, const void* x, int64_t num_tokens, int64_t d, int64_t x_stride,
                                  int64_t out_stride, int dtype, void* stream);

/* out[t, j] = gelu(x[t, j]) * x[t, d + j], exact (erf) GELU.  Replaces flashinfer.gelu_and_mul
 * (P/layers/activation.py:15-18; chosen by hidden_act == "gelu", P/models/utils.py:37-41). */
int msgl_gelu_and_mul(void* out, const void* x, int64_t num_tokens, int64_t d, int64_t x_stride,
                      int64_t out_stride, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Paged decode attention (one query token per request).  Replaces the decode
 * phase of flash_attn_with_kvcache (P/attention/fa.py:158-182) and of
 * BatchDecodeWithPagedKVCacheWrapper.run (P/attention/fi.py:188).
 *
 * The page table is the reference's GLOBAL table (P/engine/engine.py:66-73):
 * int32 [rows, pt_stride], entry (row, t) = token slot of position t.  It is
 * read in place -- no per-step page-table copy (cf. P/attention/fa.py:92-97).
 * req_rows[b] gives the table row of request b (NULL: row b).
 *
 * Work is split by a device-side plan into equal slots of 16-token tiles, one
 * slot per resident wave of a kv head (a slot = consecutive (request, tile
 * range) pieces; pieces of one request are merged afterwards):
 *   msgl_attn_decode_plan      once per step (seq_lens -> slots / pieces)
 *   msgl_attn_decode           once per layer (partial attention + merge)
 * Both are capture-safe; grids are fixed by (max_bs, capacity, head counts).
 * min_chunk: smallest slot in tokens (power of two >= 16).  The plan buffer
 * must be 16-byte aligned; plan[0] = pieces, [1] = tokens per slot, [3] = slots.
 * ---------------------------------------------------------------------- */
/* Combining the split-KV partial sums: a merge kernel follows the partial kernel.  Alternative kept for measurement
 * (msgl_attn_decode_select(72)): the piece of a request that arrives LAST (per-(request, kv head) arrival counters at the end
 * of the plan buffer, zeroed by msgl_attn_decode_plan and left at zero by every launch) reads the others' partial sums --
 * published write-through -- and writes the output itself, in piece order with the merge kernel's arithmetic: same bits,
 * one launch per layer less; inside the captured Qwen3-14B step worth 0 .. 45 us per step, on the TP-shard shapes and small
 * batches 3 .. 8 us per layer SLOWER (profiles/r04_decode_ab_with_combine_as_impl0.txt, r04_decode_ab_final.txt), hence opt-in; msgl_attn_decode then WRITES those
 * counters inside `plan`. */
/* number of int32 words the plan buffer needs */
int64_t msgl_attn_decode_plan_words(int max_bs, int capacity);
/* bytes of fp32 workspace for split-KV partials */
int64_t msgl_attn_decode_workspace_bytes(int capacity, int num_q_heads, int head_dim);
int msgl_attn_decode_plan(int32_t* plan, const int32_t* seq_lens, int batch, int max_bs,
                          int capacity, int num_q_heads, int num_kv_heads, int min_chunk,
                          void* stream);
int msgl_attn_decode(void* out, const void* q, const void* k_cache, const void* v_cache,
                     const int32_t* page_table, int64_t pt_stride, const int32_t* req_rows,
                     const int32_t* seq_lens, const int32_t* plan, void* workspace, int batch,
                     int max_bs, int capacity, int num_q_heads, int num_kv_heads, int head_dim,
                     int64_t q_stride_tok, int64_t kv_stride_tok, int64_t kv_stride_head,
                     int64_t out_stride_tok, float sm_scale, int slot_run, int dtype, void* stream);
/* Which partial-attention kernel msgl_attn_decode_plan / msgl_attn_decode use from now on (process-wide; plan and
 * launch must be made under the same choice): 0 = default = the matrix-core kernel for every slot_run (round 6: token-granular
 * tables, slot_run < 16, are gathered by per-lane addresses on the same kernel; whole-line nt requests with K transposed through
 * LDS where a token row is >= 1 KB, round 5's 16-row x 64-B direct-to-operand requests below), 1 = the round-1 streaming kernel
 * only; for timing and diagnosis also 22 / 32 = matrix-core kernel held to 2 / 3 waves per SIMD (two-stage request ring), 60 / 61 =
 * the default variant with the round-5 / the whole-line request shape forced (same bits either way), 72 = variant 22 with the
 * in-kernel combine instead of the merge kernel, 71 = the default variant by number (72's A/B partner), 92 = variant 22 with the
 * products left out (what the request pattern alone costs), 93 = variant 22 leaving clock stamps (msgl_attn_decode_trace), 94 =
 * variant 22 without the per-piece record prefetch.  Also settable by MSGL_DECODE_IMPL before the first call.  Both kernels meet
 * the same tolerance against the oracle. */
int msgl_attn_decode_select(int impl);
/* Diagnosis: under msgl_attn_decode_select(93) every wave leaves 16 uint64 shader-clock stamps at stamps[16 * wave]:
 * [0] entry, [1] slot known, then per piece i: [2 + 4 i] metadata known, [3 + 4 i] first tile arrived, [4 + 4 i] last
 * tile consumed, [5 + 4 i] results stored; [14] = stamps taken, [15] = exit.  Clocks are per XCD: only differences
 * inside one wave mean anything.  `stamps` (device memory, 128 bytes per launched wave) stays registered until
 * replaced; NULL unregisters.  tools/decode_trace.py. */
int msgl_attn_decode_trace(void* stamps);
/* slot_run: the caller's guarantee that every ALIGNED run of slot_run positions of a request maps to
 * consecutive token slots (= the engine's page_size under the reference's page-aligned allocation,
 * P/scheduler/cache.py:42-53,127-146; the property fa.py:92-97 relies on).  1 = no guarantee.
 * With slot_run >= 16 the kernel reads one table entry per 16-token tile through the scalar cache; below, each 16-lane group
 * of a wave reads the four entries of its four tokens as one 16-byte load per tile. */

/* ------------------------------------------------------------------------
 * Paged varlen causal prefill attention (MFMA).  Replaces the prefill phase
 * of flash_attn_with_kvcache (P/attention/fa.py:158-182) and
 * BatchPrefillWithPagedKVCacheWrapper.run (P/attention/fi.py:188).
 * q: [T, Hq, D] (tokens of all requests concatenated, cu_seqlens_q [B+1]);
 * request b attends keys 0..seq_lens[b]-1 through the page table; query j of
 * q_len sees keys t <= seq_lens[b] - q_len + j (bottom-right causal mask).
 * tile_cu [B+1]: exclusive prefix of ceil(q_len_b / msgl_attn_prefill_q_tile(impl)).
 * ---------------------------------------------------------------------- */
#define MSGL_PREFILL_QTILE 128
int msgl_attn_prefill(void* out, const void* q, const void* k_cache, const void* v_cache,
                      const int32_t* page_table, int64_t pt_stride, const int32_t* req_rows,
                      const int32_t* seq_lens, const int32_t* cu_seqlens_q,
                      const int32_t* tile_cu, int batch, int total_tiles, int num_q_heads,
                      int num_kv_heads, int head_dim, int64_t q_stride_tok,
                      int64_t kv_stride_tok, int64_t kv_stride_head, int64_t out_stride_tok,
                      float sm_scale, int dtype, const int32_t* tile_order, int impl, void* stream);
/* tile_order (device, [total_tiles], may be NULL = natural order): the order in which q tiles are
 * scheduled -- the host passes tiles sorted by decreasing key count so the launch tail is made of light
 * tiles (results do not depend on it).  impl: 0 = default = 4: K/V tiles staged global -> LDS by DMA
 * (global_load_lds_dwordx4, ds_read_b64_tr_b16 V fragments, double-buffered LDS, XCD-contiguous kv heads, softmax scale
 * folded into the exponent's fma); 2 = the same math with register staging (the A/B partner; bit-identical results).
 * Every other value is refused.  (Rounds 1-3 also had 1, 3, 5 -- a first-generation and a counter-phase kernel -- and
 * timing-only ablation codes >= 16; the former were removed, the latter exist only in a -DMSGL_PREFILL_DIAG build.) */
/* Rows per q tile: tile_cu, total_tiles and tile_order of msgl_attn_prefill are in units of this many query rows
 * (MSGL_PREFILL_QTILE for both kernels). */
int msgl_attn_prefill_q_tile(int impl);

/* ------------------------------------------------------------------------
 * Sampling.  Replaces torch.argmax at P/engine/sample.py:73-74 and
 * flashinfer.sampling.{softmax, sampling_from_probs, top_k_sampling_from_probs,
 * top_p_sampling_from_probs, top_k_top_p_sampling_from_probs}
 * (P/engine/sample.py:30-45).
 * logits_dtype: MSGL_BF16 / MSGL_FP16 / MSGL_F32.  out: int32 [rows].
 * ---------------------------------------------------------------------- */
int msgl_argmax_rows(int32_t* out, const void* logits, int64_t rows, int64_t vocab,
                     int64_t row_stride, int logits_dtype, void* stream);
/* probs[r, :] = softmax(logits[r, :] / temperature[r])   (fp32 out) */
int msgl_softmax_temperature(float* probs, const void* logits, const float* temperatures,
                             int64_t rows, int64_t vocab, int64_t logits_stride,
                             int64_t probs_stride, int logits_dtype, void* stream);
/* Sample one token per row from probs restricted to top-k then top-p
 * (top_k == NULL: no top-k, top_p == NULL: no top-p).  Philox(seed, offset+row). */
int msgl_sample_top_k_top_p(int32_t* out, const float* probs, const int32_t* top_k,
                            const float* top_p, int64_t rows, int64_t vocab,
                            int64_t probs_stride, uint64_t seed, uint64_t offset, void* stream);
/* Sample one token per row straight from logits: index ~ softmax(logits[r, :] / temperature[r]), no
 * top-k / top-p filter, probabilities never materialised (the fused form of the reference's
 * softmax -> sampling_from_probs pair, P/engine/sample.py:24-44).  Rows 16-byte aligned.
 * Philox(seed, offset+row). */
int msgl_sample_from_logits(int32_t* out, const void* logits, const float* temperatures, int64_t rows,
                            int64_t vocab, int64_t logits_stride, int logits_dtype, uint64_t seed,
                            uint64_t offset, void* stream);

/* ------------------------------------------------------------------------
 * Weight-streaming projection GEMM for small decode batches (1 <= M <= 64):
 * out[M, N] = x[M, K] . w[N, K]^T, row-major, 16-bit in/out, fp32 accumulate -- the reference's
 * F.linear at P/layers/linear.py:32,103,124 for decode-sized M, where the BLAS library's kernels
 * stream weights at 1.4-4.4 TB/s.  N % 16 == 0, K % 64 == 0, leading dimensions in elements
 * (ldx, ldw multiples of 8, ldo of 4).  Two tuning knobs, results do not depend on them beyond fp32
 * summation order: `slices` = waves splitting K inside a workgroup (<= K/64 and <= 16, 8 or 4 as
 * the register footprint grows), `row_tiles` = 16-row weight tiles per wave (1, 2, 4; N must be a
 * multiple of 16 row_tiles).  Deterministic, no workspace, capturable.
 * ---------------------------------------------------------------------- */
int msgl_skinny_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx,
                        int64_t ldw, int64_t ldo, int dtype, int slices, int row_tiles, void* stream);
/* The gated-MLP pair at decode-sized M in one launch -- gate_up_proj then act_fn (P/models/utils.py:45-51, P/layers/
 * activation.py:9-12): w [N, K] is the gate_up weight with its rows interleaved in blocks of 32 (rows 64 j .. + 31 gate,
 * + 32 .. + 63 up: msgl_silu_and_mul_interleaved's layout), out[M, N/2] = silu(x . gate^T) * (x . up^T) with gate and up
 * rounded to the 16-bit type before the activation: the same bits as msgl_skinny_gemm_nt followed by
 * msgl_silu_and_mul_interleaved.  N % 64 == 0, ldo >= N/2; row_tiles 2 or 4 (a wave holds gate and up tiles) or 1 with an
 * even `slices` (half of the waves stream the gate tile, half the up tile, slices / 2 k-slices each). */
int msgl_skinny_gemm_silu_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx,
                             int64_t ldw, int64_t ldo, int dtype, int slices, int row_tiles, void* stream);

/* The same product for the smallest decode batches (1 <= M <= 8) as a pure stream of the weight matrix
 * (P/layers/linear.py:32,103,124 at batch sizes 1..8): one 8-wave workgroup per CU reads the consecutive rows
 * [c N / G, (c + 1) N / G) of w as one contiguous window, `depth` (8 or 16) 16-byte loads in flight per lane, x staged
 * once into LDS, fp32 dot products on the vector units, the eight wave sums of an output added in wave order
 * (deterministic; the summation order differs from msgl_skinny_gemm_nt's).  K % 512 == 0; x and the per-row partial
 * sums must fit the CU's LDS (msgl_rowstream_gemm_supported says).  No workspace, capturable.
 * `mode` folds the neighbouring row kernel of a decoder layer into the staging pass:
 *   0  out = x . w^T                                                     (x [M, K])
 *   1  out = (silu(x[:, :K]) * x[:, K:]) . w^T                            (x [M, 2 K]: gate | up halves; the bits of
 *      msgl_silu_and_mul followed by mode 0 -- P/layers/activation.py:9-12 + down_proj)
 *   2  as 1 with x in msgl_silu_and_mul_interleaved's layout (blocks of 32: gate, up)
 *   3  s = x + res_in; res_out = round(s); out = (rmsnorm(s) * gamma) . w^T   (x [M, K] = the previous projection's
 *      output: msgl_fused_add_rmsnorm -- P/layers/norm.py:33-38 -- followed by mode 0, bit for bit; M <= 4,
 *      1024 < K <= 8192; res_out must be a different buffer from res_in and x: every workgroup reads them while
 *      workgroup 0 writes the new residual)
 * res_in / res_out / gamma / eps / ldr_* are read in mode 3 only (pass NULL / 0 otherwise).
 * `variant` 0: dot products on the vector units, units of one row x 512 k (K % 512 == 0; the fastest form at M = 1);
 * variant 1: on the matrix cores (v_mfma_f32_4x4x4_16b), units of four rows x 128 k, one LDS read of x per four staged
 * rows whatever M is (N % 4 == 0, K % 128 == 0; for M = 2 .. 8).  The two variants add in different orders. */
int msgl_rowstream_gemm_supported(int M, int N, int K, int mode, int variant);
int msgl_rowstream_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                           int64_t ldo, int dtype, int depth, int mode, int variant, const void* res_in, void* res_out,
                           const void* gamma, float eps, int64_t ldr_in, int64_t ldr_out, void* stream);

/* Same product for mid-size decode batches (1 <= M <= 256; meant for 32 < M): the 8 waves of a workgroup
 * stream different weight rows over the same k range and share the activation tile through LDS.
 * N % (128 row_tiles) == 0, K % 64 == 0, ld* multiples of 8.  `row_tiles` (1, 2) and `k_splits`
 * (1 .. K/64) are tuning knobs; with k_splits > 1 the caller provides
 * msgl_wstream_gemm_workspace_bytes(M, N, k_splits) bytes of scratch (fp32 slabs, added in split order:
 * deterministic).  No allocation, no sync. */
int64_t msgl_wstream_gemm_workspace_bytes(int M, int N, int k_splits);
/* msgl_wstream_gemm_nt with k_splits >= 2 WITHOUT its reduce launch: the fp32 slabs part[k_splits][M][N] stay in
 * `workspace` for a slab-input consumer (msgl_fused_add_rmsnorm_slabs, msgl_qk_norm_rope_store_slabs), which adds them in
 * split order and rounds -- the bits the reduce launch would have stored. */
int msgl_wstream_gemm_slabs_nt(const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw, int dtype,
                               int row_tiles, int k_splits, void* workspace, int64_t workspace_bytes, void* stream);
int msgl_wstream_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx,
                         int64_t ldw, int64_t ldo, int dtype, int row_tiles, int k_splits,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* Full-batch decode projection, 128 < M <= 256 (csrc/gemm_m256.hip): the same `F.linear` of
 * P/layers/linear.py:32,103,124.  One workgroup per CU walks 256 x 128 output tiles with both operands staged
 * by LDS-DMA through a three-stage ring and v_mfma_f32_32x32x16.  Plan = (grid, full, tail_split): the first
 * `full` tiles of N / 128 are computed whole, the rest are cut into `tail_split` k-slices spread over all
 * workgroups (fp32 slabs in `workspace`, added in slice order by a second kernel: deterministic).
 * N % 128 == 0, K % 64 == 0. */
int64_t msgl_m256_gemm_workspace_bytes(int M, int N, int full, int tail_split);
int msgl_m256_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                      int64_t ldo, int dtype, int grid, int full, int tail_split, void* workspace,
                      int64_t workspace_bytes, void* stream);
/* The k-sliced plan (full == 0, tail_split > 1) without its reduce launch: `workspace` is left holding
 * tail_split fp32 slabs [M][N] (slab stride M * N floats) for msgl_fused_add_rmsnorm_slabs, the operation that
 * follows o_proj and down_proj in the reference's decoder layer (P/models/qwen3.py:36-41: the projection output
 * goes straight into the next layer norm's fused residual add).  The slabs are valid until the next GEMM that
 * uses the same workspace. */
int msgl_m256_gemm_slabs_nt(const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                            int dtype, int grid, int tail_split, void* workspace, int64_t workspace_bytes,
                            void* stream);

/* Generation 3 of the full-batch projection (csrc/gemm_g3.hip), same product and the same plan triple as
 * msgl_m256_gemm_nt: four loader waves per workgroup stream both operands by LDS-DMA into a three-stage LDS ring,
 * four matrix waves run v_mfma_f32_32x32x16 and never issue a vector-memory instruction inside the k loop.
 * `flags`:
 *   MSGL_G3_SILU        `w` is a gate_up matrix whose rows are interleaved in blocks of 32 (tile t of 128 rows =
 *                       gate[64t, +32), up[64t, +32), gate[64t+32, +32), up[64t+32, +32): ops.interleave_gate_up);
 *                       `out` is [M, N/2] = silu(gate) * up, bit-identical to rounding the projection to the 16-bit
 *                       type and applying msgl_silu_and_mul_interleaved (P/layers/activation.py:9-12 after
 *                       P/layers/linear.py:32).
 *   MSGL_G3_SLABS_ONLY  pure k-slicing (full == 0, tail_split > 1) without the reduce launch: `workspace` keeps the
 *                       fp32 slabs [tail_split][M][N] for the consumer (as msgl_m256_gemm_slabs_nt); `out` unused.
 *   bits 8-15           diagnosis variants (cache policy of the weight stream, ablations): 0 in production.
 * Workspace: msgl_m256_gemm_workspace_bytes(M, N, full, tail_split). */
#define MSGL_G3_SILU 1
#define MSGL_G3_SLABS_ONLY 2
int msgl_g3_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                    int64_t ldo, int dtype, int grid, int full, int tail_split, int flags, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* Row-owner generation of the decode projection (csrc/gemm_ro.hip), planned for 3 <= M <= 256 (the entry point takes 1 <= M <= 256): the same `F.linear` of
 * P/layers/linear.py:32,103,124 (and the LM head, P/layers/embedding.py:98).  The N / 16 sixteen-row units of `w` are cut
 * into `tiles` balanced contiguous ranges (widths differ by at most one unit; at most msgl_ro_gemm_max_units(M) units each)
 * and every range into `slices` k-slices; item (tile, slice) runs on workgroup (tile * slices + slice) % grid, grid = min(CU
 * count, items).  Four loader waves per workgroup stream both operands by LDS-DMA through a three-stage ring, eight matrix
 * waves run v_mfma_f32_16x16x32.  slices == 1: `out` [M, N] is written directly (a CU that owns whole rows of w: no split-K
 * partial sums).  slices > 1: fp32 slabs [slices][M][N] in `workspace`, added in slice order by a reduce launch, or left to the
 * consumer with MSGL_RO_SLABS_ONLY (msgl_fused_add_rmsnorm_slabs / msgl_qk_norm_rope_store_slabs, as msgl_m256_gemm_slabs_nt).
 * `flags`:
 *   MSGL_RO_SILU        slices == 1 only: `w` is a gate_up matrix in ops.interleave_gate_up order (as MSGL_G3_SILU), `out` is
 *                       [M, N/2] = silu(gate) * up, bit-identical to this entry point without the flag followed by
 *                       msgl_silu_and_mul_interleaved (P/layers/activation.py:9-12).
 *   MSGL_RO_SLABS_ONLY  slices > 1 without the reduce launch; `out` unused.
 *   bits 8-15           diagnosis: 1 = no x loads, 2 = no MFMAs, 4 = no w loads (timing only, results meaningless); 8 / 16 = A/B variants of
 *                       the matrix side with identical results; 0 in production.
 * N % 16 == 0 (MSGL_RO_SILU: N % 64 == 0), K % 64 == 0.  Workspace: msgl_ro_gemm_workspace_bytes(M, N, slices). */
#define MSGL_RO_SILU 1
#define MSGL_RO_SLABS_ONLY 2
int msgl_ro_gemm_max_units(int M);
int64_t msgl_ro_gemm_workspace_bytes(int M, int N, int slices);
int msgl_ro_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                    int64_t ldo, int dtype, int tiles, int slices, int flags, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Native radix prefix tree (libmsgl_hip.so, csrc/radix.cpp; host code, no device work).
 * Replaces: the tree walk of RadixPrefixCache -- _tree_walk, RadixTreeNode.split_at / get_match_len, lock_handle,
 * evict, _collect_leave_nodes_for_evict (python/minisgl/kvcache/radix_cache.py:65-77, 111-130, 147-175, 189-230),
 * which the reference runs in Python with one dict lookup, one tensor slice and one tvm-ffi call per node
 * (SURVEY.md section 8f rank 4).  The tree owns keys (token ids), reference counts, timestamps and the two size counters;
 * the VALUE tensors (pool slots, device memory) and the clock stay with the caller (mini-sglang_amd/radix.py), which
 * mirrors each reported split / eviction on them.  Nodes are named by int64 ids (the root is id 0).
 * Children are an insertion-ordered map and eviction is CPython's heapq over timestamps, so that every observable
 * (matched lengths, handles, evicted slots, their order) equals the reference's for the same calls and clock.
 *   create:     page_size = the scheduler's; now_ns stamps the root
 *   walk:       out[5] = {node, matched length, split head id, split tail id, split position} (-1 x 3: no split);
 *               stamps fully matched nodes with now_ns; a split cuts the tail's key (and must cut its value) at position
 *   add_child:  new leaf under `parent` with key[n] (whole pages), stamped now_ns; evictable size += n; returns its id
 *   lock:       lock_handle(node, unlock) along the path to the root
 *   evict:      at least `size` tokens from unreferenced leaves, oldest first; out_ids = the evicted nodes in order
 *               (their values, concatenated in this order, are what the reference's evict returns); returns the count
 *   path:       node ids from below the root down to `node` (get_matched_indices concatenates their values)
 *   info:       out[4] = {evictable, protected, live nodes, key length of `node` or -1}
 *   check:      recomputes sizes and links (check_integrity)
 * Errors: < 0 with msgl_last_error() (unknown / evicted node id, evict beyond the evictable size with the reference's
 * message, ragged key). */
int msgl_radix_create(void** tree, int page_size, int64_t now_ns);
int msgl_radix_destroy(void* tree);
int msgl_radix_walk(void* tree, const int32_t* ids, int64_t n, int64_t now_ns, int64_t* out);
int64_t msgl_radix_add_child(void* tree, int64_t parent, const int32_t* key, int64_t n, int64_t now_ns);
int msgl_radix_lock(void* tree, int64_t node, int unlock);
int64_t msgl_radix_evict(void* tree, int64_t size, int64_t* out_ids, int64_t capacity);
int64_t msgl_radix_path(void* tree, int64_t node, int64_t* out_ids, int64_t capacity);
int msgl_radix_info(void* tree, int64_t node, int64_t* out);
int msgl_radix_check(void* tree);

/* ------------------------------------------------------------------------
 * Peer-to-peer collectives over xGMI for decode-size messages (libmsgl_hip.so, csrc/comm_p2p.hip).
 * What the reference gets from its symmetric buffer (ncclMemAlloc + ncclCommWindowRegister(
 * NCCL_WIN_COLL_SYMMETRIC), C/src/pynccl.cu:81-90) and NCCL's symmetric kernels (all_reduce through the
 * window for messages <= max_bytes, pynccl.cu:105-123): every rank owns one fine-grained buffer, all ranks
 * map all buffers (hipIpc), one kernel per collective reads the peers directly: one-shot below
 * `one_shot_max_bytes`, reduce-scatter + all-gather (all links at once) above.  In place, SUM in rank order
 * (identical bits on every rank), capturable, never synchronises; bounded spins (msgl_p2p_error).
 * Set-up: create -> exchange msgl_p2p_ipc_handle bytes over the host's CPU group -> open. */
#define MSGL_IPC_HANDLE_BYTES 64
typedef struct msgl_p2p* msgl_p2p_t;
int msgl_p2p_create(msgl_p2p_t* out, int rank, int world_size, size_t max_bytes);
int msgl_p2p_ipc_handle(msgl_p2p_t comm, void* out_handle /* MSGL_IPC_HANDLE_BYTES */);
int msgl_p2p_open(msgl_p2p_t comm, const void* all_handles /* world_size x MSGL_IPC_HANDLE_BYTES, rank order */);
int msgl_p2p_configure(msgl_p2p_t comm, size_t one_shot_max_bytes, int blocks);
int msgl_p2p_all_reduce_sum(msgl_p2p_t comm, void* data, size_t count, int dtype, void* stream);
int msgl_p2p_all_gather(msgl_p2p_t comm, void* dst, const void* src, size_t count, int dtype, void* stream);
/* all-reduce + residual add + RMSNorm as one launch for decode-size row blocks: the reference's
 * `y = self._comm.all_reduce(F.linear(...))` (P/layers/linear.py:102-106, 123-127) followed by RMSNormFused's
 * fused_add_rmsnorm (P/layers/norm.py:33-38).  x [rows, dim] = this rank's partial projection in, normalised output
 * out; residual in / out.  Bit-identical to msgl_p2p_all_reduce_sum + msgl_fused_add_rmsnorm on every rank.  Returns
 * MSGL_EINVAL for shapes it does not cover (more than 512 rows per call, dim > 8192, message above the
 * buffer): the caller then issues the two launches. */
int msgl_p2p_all_reduce_add_rmsnorm(msgl_p2p_t comm, void* x, void* residual, const void* weight, float eps,
                                    int64_t rows, int64_t dim, int64_t x_stride, int64_t res_stride, int dtype,
                                    void* stream);
/* A barrier that gives up (a peer did not arrive within the spin limit) sets a sticky error word (bits 0-3: 1 + phase,
 * 4-7: collective kind 1 one-shot / 2 two-shot / 3 fused norm / 4 all-gather, 8-15: block, 16-19: the peer waited for)
 * in its own header AND in every peer's (there with bit 20 set and bits 16-19 = the rank that gave up), and the kernel
 * POISONS its output with NaN bit patterns instead of returning a partial sum; once the word is set every later
 * collective of the communicator -- on the failing rank and, from their next barrier on, on its peers -- poisons at once
 * (no further spinning).  msgl_p2p_error reads the word (synchronises
 * the device); msgl_p2p_error_async enqueues a 4-byte copy of it into pinned host memory on `stream` for hosts that
 * poll every few steps (kernel.P2PCommunicator.poll_error raises).  msgl_p2p_set_spin_limit: polls per barrier before
 * giving up (default 40 M ~ tens of seconds). */
int msgl_p2p_error(msgl_p2p_t comm);
int msgl_p2p_error_async(msgl_p2p_t comm, void* pinned_host_u32, void* stream);
int msgl_p2p_set_spin_limit(msgl_p2p_t comm, uint32_t spins);
void* msgl_p2p_get_buffer(msgl_p2p_t comm);
int msgl_p2p_destroy(msgl_p2p_t comm);   /* detaches; buffers stay mapped until msgl_p2p_release_all / exit */
int msgl_p2p_release_all(void);

/* ------------------------------------------------------------------------
 * Tensor-parallel communicator over RCCL (libmsgl_comm.so).  Replaces
 * NCCLWrapper (C/src/pynccl.cu:72-175) / init_pynccl (P/kernel/pynccl.py:47-78).
 * ---------------------------------------------------------------------- */
#define MSGL_UNIQUE_ID_BYTES 128
typedef struct msgl_comm* msgl_comm_t;
int msgl_comm_unique_id(char out_id[MSGL_UNIQUE_ID_BYTES]);
int msgl_comm_create(msgl_comm_t* out, int rank, int world_size,
                     const char id[MSGL_UNIQUE_ID_BYTES], size_t max_bytes);
/* in-place SUM all-reduce of `count` 16-bit floats */
int msgl_comm_all_reduce_sum(msgl_comm_t comm, void* data, size_t count, int dtype, void* stream);
/* dst[rank*count .. ] <- src of every rank */
int msgl_comm_all_gather(msgl_comm_t comm, void* dst, const void* src, size_t count, int dtype,
                         void* stream);
void* msgl_comm_get_buffer(msgl_comm_t comm);
/* ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the communicator (any pointer may be NULL) */
int msgl_comm_info(msgl_comm_t comm, int* nranks, int* rank, int* device);
int msgl_comm_destroy(msgl_comm_t comm);
const char* msgl_comm_last_error(void);

/* ------------------------------------------------------------------------
 * Projection GEMMs (libmsgl_gemm.so, links libhipblaslt).  Replaces the
 * `F.linear` call sites of the reference (P/layers/linear.py:32,103,124,
 * P/layers/embedding.py:98): out[M, N] = x[M, K] . w[N, K]^T, row-major,
 * bf16/fp16 in and out, fp32 accumulate; ld* are row strides in elements.
 * msgl_gemm_nt launches the solution remembered for the shape (the library's
 * heuristic pick until msgl_gemm_tune has run for it) with the caller's
 * workspace: no allocation, no sync.  msgl_gemm_tune times library solutions
 * on `n_w` rotating weight buffers (max_candidates: 0 = all, n > 1 = first n,
 * -n = heuristic top n, 1 = heuristic pick only; split_k_search != 0 additionally
 * tries every solution with K split over 2..16 workgroups in the exhaustive modes),
 * remembers the fastest and SYNCHRONISES (initialisation-time call, not capturable).
 * ---------------------------------------------------------------------- */
int msgl_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx,
                 int64_t ldw, int64_t ldo, int dtype, void* workspace, int64_t workspace_bytes,
                 void* stream);
int msgl_gemm_tune(void* out, const void* x, const void* const* w_list, int n_w, int M, int N,
                   int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype, void* workspace,
                   int64_t workspace_bytes, int max_candidates, int split_k_search, int iters,
                   float* best_us, float* default_us, int* best_index, int* best_split_k,
                   int* n_tried, void* stream);
/* kernel name of the remembered solution into buf; returns its library index (< 0 on error) */
int msgl_gemm_reset_plans(void);  /* forget all plans: shapes fall back to the library heuristic */
/* The best few candidates of the shape's last msgl_gemm_tune, fastest first (count returned, times in us_out), and the
 * switch that makes one of them the shape's plan: back-to-back timing separates the top library solutions by ~1 %,
 * inside a captured decode step they differ by up to 9 %, so the host re-ranks them in the step itself. */
int msgl_gemm_finalists(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype, float* us_out, int max_n);
int msgl_gemm_select_finalist(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype, int index);
/* A searched plan as data: (library solution index, split-K factor) of the shape -- get returns 1 / 0 (searched plan / none),
 * set installs one without a search (fails if this library build lacks the index or the solution does not support the shape
 * within workspace_bytes).  How one process's search result is replayed in another (tests/test_gpu_reference_driven.py). */
int msgl_gemm_get_plan(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype, int* algo_index, int* split_k);
int msgl_gemm_set_plan(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype, int algo_index, int split_k,
                       void* workspace, int64_t workspace_bytes);
int msgl_gemm_solution_name(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype,
                            char* buf, int buf_len);
const char* msgl_gemm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MSGL_HIP_H_ */
