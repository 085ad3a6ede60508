"""MI355X-native paged-attention serving core behind mini-sglang's operator API.

Layout (only what the hot path needs):
  csrc/              hand-written HIP kernels for gfx950 + the C-ABI (include/msgl_hip.h)
  _lib.py, ops.py    ctypes binding and torch-facing wrappers (pointers + strides + stream)
  kernel.py          mirror of minisgl.kernel   (store_cache, indexing, fast_compare_key, init_pynccl: peer-to-peer +
                     RCCL communicators)
  flashinfer_compat  mirror of the flashinfer functions the reference binds (norm, rope, act, sampling)
  core.py            Req / Batch / Context / SamplingParams mirrors (P/core.py)
  kvcache.py         MHAKVCache mirror (P/kvcache/mha_pool.py)
  attention.py       HipAttnBackend: the BaseAttnBackend plugin (P/attention/base.py)
  radix.py           RadixPrefixCache with the tree walk in native code (P/kvcache/radix_cache.py)
  sched_glue.py      vectorised positions / input / write index tensors (P/scheduler/scheduler.py:236-267)
  gemm_plan.py       per-shape kernel choice for the projection GEMMs, timed before graph capture
  model.py, engine.py, offline.py  dense decoder, engine and offline driver used by bench.py / smoke (callers of the path)
  minisgl_plugin.py  registers all of the above into a real `minisgl` install
"""
__version__ = "0.1.0"
