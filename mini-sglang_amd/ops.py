"""torch-facing wrappers over the C-ABI (include/msgl_hip.h).

PyTorch is used for device memory and streams only: each wrapper validates what the C
side cannot see (device, dtype), passes raw pointers / strides / the current HIP stream
through ctypes, and returns.  No wrapper allocates except where the mirrored reference
function does (e.g. `indexing` output), none synchronises.
"""
from __future__ import annotations

import functools
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _lib.BF16
    if t.dtype == torch.float16:
        return _lib.FP16
    raise TypeError(f"expected a bf16/fp16 tensor, got {t.dtype}")


def _logits_dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.F32
    return _dt(t)


def _need_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError(
                "mini_sglang_amd ops run on the HIP device only (got a CPU tensor); "
                "there is no CPU fallback in the product path"
            )


def _is_i64(t: torch.Tensor) -> int:
    if t.dtype == torch.int64:
        return 1
    if t.dtype == torch.int32:
        return 0
    raise TypeError(f"index tensor must be int32/int64, got {t.dtype}")


# ---------------------------------------------------------------------------- store / gather
def store_kv(k_cache: torch.Tensor, v_cache: torch.Tensor, indices: torch.Tensor, k: torch.Tensor,
             v: torch.Tensor) -> None:
    """k_cache[indices] = k; v_cache[indices] = v over rows of `row` elements.

    k_cache/v_cache: [slots, row] views (row stride free), k/v: [T, row] views (row stride
    free, e.g. slices of the fused qkv tensor).  Reference: C/jit/store.cu:59-121.
    """
    _need_cuda(k_cache, v_cache, indices, k, v)
    assert k_cache.dim() == 2 and v_cache.dim() == 2 and k.dim() == 2 and v.dim() == 2
    assert k_cache.shape[1] == k.shape[1] == v.shape[1] == v_cache.shape[1]
    assert k_cache.stride(1) == v_cache.stride(1) == k.stride(1) == v.stride(1) == 1
    assert k_cache.dtype == v_cache.dtype == k.dtype == v.dtype
    assert k_cache.stride(0) == v_cache.stride(0)
    assert indices.dim() == 1 and indices.shape[0] == k.shape[0] == v.shape[0] and indices.is_contiguous()
    es = k.element_size()
    check(
        lib().msgl_store_kv(
            k_cache.data_ptr(), v_cache.data_ptr(), indices.data_ptr(), _is_i64(indices), k.data_ptr(),
            v.data_ptr(), k.shape[0], k.shape[1] * es, k_cache.stride(0) * es, k.stride(0) * es,
            v.stride(0) * es, _stream(),
        ),
        "store_kv",
    )


def embedding_gather(weight: torch.Tensor, indices: torch.Tensor, out: Optional[torch.Tensor] = None,
                     vocab_range: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    _need_cuda(weight, indices)
    assert weight.dim() == 2 and weight.is_contiguous() and indices.dim() == 1 and indices.is_contiguous()
    if out is None:
        out = weight.new_empty(indices.shape[0], weight.shape[1])
    assert out.is_contiguous() and out.shape == (indices.shape[0], weight.shape[1]) and out.dtype == weight.dtype
    start, length = vocab_range if vocab_range is not None else (0, 0)
    check(
        lib().msgl_embedding_gather(
            out.data_ptr(), weight.data_ptr(), indices.data_ptr(), _is_i64(indices), indices.shape[0],
            weight.shape[1] * weight.element_size(), 1 if vocab_range is not None else 0, start, length,
            _stream(),
        ),
        "embedding_gather",
    )
    return out


def fast_compare_key(x: torch.Tensor, y: torch.Tensor) -> int:
    if not (x.dim() == 1 and y.dim() == 1 and x.is_contiguous() and y.is_contiguous() and x.device.type == "cpu"
            and y.device.type == "cpu" and x.dtype in (torch.int32, torch.int64)):
        raise RuntimeError("Both tensors must be 1D CPU int tensors.")
    if x.dtype != y.dtype:
        raise RuntimeError("fast_compare_key: dtype mismatch")
    r = lib().msgl_fast_compare_key(x.data_ptr(), x.shape[0], y.data_ptr(), y.shape[0], x.element_size())
    if r < 0:
        check(int(r), "fast_compare_key")
    return int(r)


# ---------------------------------------------------------------------------- norm / rope / act
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x * rsqrt(mean(x^2) + eps) * w over the last dim of a 2-D or 3-D (strided) tensor."""
    _need_cuda(x, weight)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    assert out.shape == x.shape and out.dtype == x.dtype == weight.dtype
    assert x.stride(-1) == 1 and out.stride(-1) == 1 and weight.is_contiguous() and weight.shape[0] == x.shape[-1]
    if x.dim() == 2:
        n0, n1, xs0, xs1, os0, os1 = x.shape[0], 1, x.stride(0), 0, out.stride(0), 0
    elif x.dim() == 3:
        n0, n1, xs0, xs1, os0, os1 = x.shape[0], x.shape[1], x.stride(0), x.stride(1), out.stride(0), out.stride(1)
    else:
        raise ValueError("rmsnorm expects a 2-D or 3-D tensor")
    check(
        lib().msgl_rmsnorm(out.data_ptr(), x.data_ptr(), weight.data_ptr(), float(eps), n0, n1, x.shape[-1],
                           xs0, xs1, os0, os1, _dt(x), _stream()),
        "rmsnorm",
    )
    return out


def fused_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> None:
    _need_cuda(x, residual, weight)
    assert x.dim() == 2 and x.shape == residual.shape and x.dtype == residual.dtype == weight.dtype
    assert x.stride(1) == 1 and residual.stride(1) == 1 and weight.is_contiguous()
    check(
        lib().msgl_fused_add_rmsnorm(x.data_ptr(), residual.data_ptr(), weight.data_ptr(), float(eps), x.shape[0],
                                     x.shape[1], x.stride(0), residual.stride(0), _dt(x), _stream()),
        "fused_add_rmsnorm",
    )


class Slabs:
    """A split-K projection whose reduce has not run: `count` fp32 slabs [rows][dim] in the GEMM workspace of
    `device`, consumed by fused_add_rmsnorm_slabs before anything else touches that workspace."""

    __slots__ = ("ptr", "count", "rows", "dim", "device")

    def __init__(self, ptr: int, count: int, rows: int, dim: int, device: int):
        self.ptr, self.count, self.rows, self.dim, self.device = ptr, count, rows, dim, device


# device index -> the Slabs whose reduce is still owed (at most one per workspace).  Every GEMM entry point that may
# use the workspace checks it: a deferred projection output that reached anything but fused_add_rmsnorm is a bug in
# the caller's wiring and must not turn into silently wrong numbers.
_PENDING_SLABS: dict = {}


# a row-parallel projection's output whose all-reduce was left to the fused all-reduce + add + RMSNorm launch of the norm
# that follows (minisgl_plugin / flashinfer_compat): any other GEMM in between means it reached a different consumer
_PENDING_ALLREDUCE: dict = {}


def _no_pending_slabs(device: torch.device) -> None:
    if _PENDING_ALLREDUCE and _PENDING_ALLREDUCE.get(device.index or 0) is not None:
        _PENDING_ALLREDUCE[device.index or 0] = None
        raise RuntimeError("a row-parallel projection's all-reduce was deferred to the fused all-reduce + RMSNorm launch, "
                           "but its output reached a different consumer (no fused_add_rmsnorm followed it)")
    if _PENDING_SLABS and _PENDING_SLABS.get(device.index or 0) is not None:
        _PENDING_SLABS[device.index or 0] = None  # report once: a failed forward must not poison every later GEMM
        raise RuntimeError("a split-K projection's partial sums are still waiting for fused_add_rmsnorm_slabs: "
                           "linear_slabs() output was handed to a different consumer (not fused_add_rmsnorm_slabs / "
                           "qk_norm_rope_store_slabs)")


def fused_add_rmsnorm_slabs(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                            slabs: Slabs) -> None:
    """fused_add_rmsnorm(x, ...) for an x = linear_slabs(...) output whose k-slice sums are still in `slabs`."""
    _need_cuda(x, residual, weight)
    assert x.dim() == 2 and x.shape == residual.shape == (slabs.rows, slabs.dim)
    assert x.dtype == residual.dtype == weight.dtype and x.stride(1) == 1 and residual.stride(1) == 1
    if _PENDING_SLABS.get(slabs.device) is not slabs:
        raise RuntimeError("fused_add_rmsnorm_slabs: these partial sums are no longer the workspace's content")
    _PENDING_SLABS[slabs.device] = None
    check(
        lib().msgl_fused_add_rmsnorm_slabs(x.data_ptr(), residual.data_ptr(), weight.data_ptr(), float(eps), x.shape[0],
                                           x.shape[1], x.stride(0), residual.stride(0), slabs.ptr, slabs.count,
                                           slabs.rows * slabs.dim, slabs.dim, _dt(x), _stream()),
        "fused_add_rmsnorm_slabs",
    )


def rope_neox_inplace(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                      cos_sin_cache: torch.Tensor) -> None:
    _need_cuda(positions, query, key, cos_sin_cache)
    assert query.dim() == 2 and key.dim() == 2 and query.stride(1) == 1 and key.stride(1) == 1
    assert cos_sin_cache.dtype == torch.float32 and cos_sin_cache.is_contiguous()
    assert cos_sin_cache.shape[1] == head_size and positions.is_contiguous()
    assert query.shape[1] % head_size == 0 and key.shape[1] % head_size == 0
    check(
        lib().msgl_rope_neox_inplace(
            query.data_ptr(), key.data_ptr(), positions.data_ptr(), _is_i64(positions), cos_sin_cache.data_ptr(),
            query.shape[0], query.shape[1] // head_size, key.shape[1] // head_size, head_size, query.stride(0),
            key.stride(0), _dt(query), _stream(),
        ),
        "rope_neox_inplace",
    )


def qk_norm_rope_store(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q_norm_w: Optional[torch.Tensor],
                       k_norm_w: Optional[torch.Tensor], eps: float, positions: torch.Tensor,
                       cos_sin_cache: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                       out_loc: torch.Tensor, head_dim: int) -> None:
    """Fused qk-norm -> RoPE -> KV store over [T, H*D] row views q, k, v (see msgl_hip.h)."""
    _need_cuda(q, k, v, positions, cos_sin_cache, k_cache, v_cache, out_loc)
    assert q.dim() == 2 and k.dim() == 2 and v.dim() == 2 and k_cache.dim() == 2 and v_cache.dim() == 2
    assert k.shape[1] == v.shape[1] == k_cache.shape[1] == v_cache.shape[1]
    assert k_cache.stride(0) == v_cache.stride(0)
    check(
        lib().msgl_qk_norm_rope_store(
            q.data_ptr(), k.data_ptr(), v.data_ptr(),
            q_norm_w.data_ptr() if q_norm_w is not None else None,
            k_norm_w.data_ptr() if k_norm_w is not None else None, float(eps), positions.data_ptr(),
            _is_i64(positions), cos_sin_cache.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
            out_loc.data_ptr(), _is_i64(out_loc), q.shape[0], q.shape[1] // head_dim, k.shape[1] // head_dim,
            head_dim, q.stride(0), k.stride(0), v.stride(0), k_cache.stride(0), _dt(q), _stream(),
        ),
        "qk_norm_rope_store",
    )


def qk_norm_rope_store_slabs(qkv: torch.Tensor, slabs: "Slabs", num_q_heads: int, num_k_heads: int,
                             q_norm_w: Optional[torch.Tensor], k_norm_w: Optional[torch.Tensor], eps: float,
                             positions: torch.Tensor, cos_sin_cache: torch.Tensor, k_cache: torch.Tensor,
                             v_cache: torch.Tensor, out_loc: torch.Tensor, head_dim: int) -> None:
    """qk_norm_rope_store for a qkv = linear_slabs(...) output whose k-slice sums are still in `slabs`: the reduce
    happens here; qkv receives q, k and v."""
    _need_cuda(qkv, positions, cos_sin_cache, k_cache, v_cache, out_loc)
    assert qkv.dim() == 2 and qkv.stride(1) == 1 and qkv.shape == (slabs.rows, slabs.dim)
    assert qkv.shape[1] == (num_q_heads + 2 * num_k_heads) * head_dim
    assert k_cache.dim() == 2 and k_cache.shape[1] == num_k_heads * head_dim and k_cache.stride(0) == v_cache.stride(0)
    if _PENDING_SLABS.get(slabs.device) is not slabs:
        raise RuntimeError("qk_norm_rope_store_slabs: these partial sums are no longer the workspace's content")
    _PENDING_SLABS[slabs.device] = None
    check(
        lib().msgl_qk_norm_rope_store_slabs(
            qkv.data_ptr(), qkv.stride(0), slabs.ptr, slabs.count, slabs.rows * slabs.dim, slabs.dim,
            q_norm_w.data_ptr() if q_norm_w is not None else None,
            k_norm_w.data_ptr() if k_norm_w is not None else None, float(eps), positions.data_ptr(),
            _is_i64(positions), cos_sin_cache.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out_loc.data_ptr(),
            _is_i64(out_loc), qkv.shape[0], num_q_heads, num_k_heads, head_dim, k_cache.stride(0), _dt(qkv), _stream(),
        ),
        "qk_norm_rope_store_slabs",
    )


def _act_and_mul(name: str, x: torch.Tensor, out: Optional[torch.Tensor]) -> torch.Tensor:
    _need_cuda(x)
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 2 == 0
    d = x.shape[1] // 2
    if out is None:
        out = torch.empty((x.shape[0], d), dtype=x.dtype, device=x.device)
    assert out.shape == (x.shape[0], d) and out.stride(1) == 1 and out.dtype == x.dtype
    check(
        getattr(lib(), f"msgl_{name}")(out.data_ptr(), x.data_ptr(), x.shape[0], d, x.stride(0), out.stride(0),
                                       _dt(x), _stream()),
        name,
    )
    return out


def silu_and_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _act_and_mul("silu_and_mul", x, out)


def silu_and_mul_interleaved(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """silu_and_mul for a gate_up row whose columns are in the block-32 interleaved order of `interleave_gate_up`."""
    assert (x.shape[1] // 2) % 64 == 0
    return _act_and_mul("silu_and_mul_interleaved", x, out)


def gate_up_interleave_index(inter: int, device=None) -> torch.Tensor:
    """Row permutation that turns [gate (inter rows); up (inter rows)] (P/layers/linear.py:50-62, LinearColParallelMerged)
    into the layout the fused projection epilogue wants: per 128 rows gate[64t, +32), up[64t, +32), gate[64t+32, +32),
    up[64t+32, +32).  new_rows = old_rows[index]."""
    assert inter % 64 == 0, inter
    r = torch.arange(2 * inter, device=device)
    t, q, u, j = r // 128, (r % 128) // 64, (r % 64) // 32, r % 32
    return u * inter + 64 * t + 32 * q + j


def interleave_gate_up(w: torch.Tensor) -> torch.Tensor:
    """A gate_up weight [2 * inter, K] (or an activation's last dimension when w.dim() == 2 and rows are tokens: use
    .t()) with its rows in the interleaved order (a new tensor)."""
    return w.index_select(0, gate_up_interleave_index(w.shape[0] // 2, w.device))


def gelu_and_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Exact (erf) GELU gate, flashinfer.gelu_and_mul (P/layers/activation.py:15-18)."""
    return _act_and_mul("gelu_and_mul", x, out)


# ---------------------------------------------------------------------------- attention
def attn_decode_plan_words(max_bs: int, capacity: int) -> int:
    n = lib().msgl_attn_decode_plan_words(max_bs, capacity)
    check(int(n), "attn_decode_plan_words")
    return int(n)


def attn_decode_workspace_bytes(capacity: int, num_q_heads: int, head_dim: int) -> int:
    n = lib().msgl_attn_decode_workspace_bytes(capacity, num_q_heads, head_dim)
    check(int(n), "attn_decode_workspace_bytes")
    return int(n)


def attn_decode_plan(plan: torch.Tensor, seq_lens: torch.Tensor, batch: int, max_bs: int, capacity: int,
                     num_q_heads: int, num_kv_heads: int, min_chunk: int = 64) -> None:
    _need_cuda(plan, seq_lens)
    assert plan.dtype == torch.int32 and seq_lens.dtype == torch.int32 and seq_lens.is_contiguous()
    assert plan.numel() >= attn_decode_plan_words(max_bs, capacity) and seq_lens.numel() >= batch
    check(
        lib().msgl_attn_decode_plan(plan.data_ptr(), seq_lens.data_ptr(), batch, max_bs, capacity, num_q_heads,
                                    num_kv_heads, min_chunk, _stream()),
        "attn_decode_plan",
    )


def attn_decode(out: torch.Tensor, q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                page_table: torch.Tensor, req_rows: Optional[torch.Tensor], seq_lens: torch.Tensor,
                plan: torch.Tensor, workspace: torch.Tensor, batch: int, max_bs: int, capacity: int,
                sm_scale: float, slot_run: int = 1) -> None:
    """q: [B, Hq, D] (token stride free), k_cache/v_cache: [slots, Hkv, D], out: [B, Hq, D].
    slot_run: aligned runs of this many positions map to consecutive slots (the engine's page_size)."""
    _need_cuda(out, q, k_cache, v_cache, page_table, seq_lens, plan, workspace)
    assert q.dim() == 3 and k_cache.dim() == 3 and out.dim() == 3
    hq, d = q.shape[1], q.shape[2]
    hkv = k_cache.shape[1]
    assert q.stride(2) == 1 and q.stride(1) == d and out.stride(2) == 1 and out.stride(1) == d
    assert k_cache.stride(2) == 1 and v_cache.stride() == k_cache.stride() and v_cache.shape == k_cache.shape
    assert page_table.dtype == torch.int32 and page_table.dim() == 2 and page_table.stride(1) == 1
    assert q.dtype == k_cache.dtype == v_cache.dtype == out.dtype
    check(
        lib().msgl_attn_decode(
            out.data_ptr(), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), page_table.data_ptr(),
            page_table.stride(0), req_rows.data_ptr() if req_rows is not None else None, seq_lens.data_ptr(),
            plan.data_ptr(), workspace.data_ptr(), batch, max_bs, capacity, hq, hkv, d, q.stride(0),
            k_cache.stride(0), k_cache.stride(1), out.stride(0), float(sm_scale), int(slot_run), _dt(q), _stream(),
        ),
        "attn_decode",
    )


def attn_decode_select(impl: int) -> None:
    """0: default kernel choice, 1: streaming (VALU) kernel only -- msgl_attn_decode_select (A/B timing, tests)."""
    check(lib().msgl_attn_decode_select(int(impl)), "attn_decode_select")


PREFILL_IMPL = int(os.environ.get("MSGL_PREFILL_IMPL", "0"))  # 0 = the library's default kernel (include/msgl_hip.h)
if PREFILL_IMPL not in (0, 2, 4):
    raise ValueError(f"MSGL_PREFILL_IMPL={PREFILL_IMPL}: 0 / 4 (DMA-staged, the default) or 2 (register-staged A/B partner); the "
                     "timing-only ablation codes exist in a diagnostic build only and are passed as impl= by tools/prefill_ablate.py")


def prefill_q_tile(impl: Optional[int] = None) -> int:
    """Query rows per tile of the prefill kernel behind `impl` (None: the process default): the unit of tile_cu,
    total_tiles and tile_order."""
    return int(lib().msgl_attn_prefill_q_tile(PREFILL_IMPL if impl is None else int(impl)))


def attn_prefill(out: torch.Tensor, q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                 page_table: torch.Tensor, req_rows: Optional[torch.Tensor], seq_lens: torch.Tensor,
                 cu_seqlens_q: torch.Tensor, tile_cu: torch.Tensor, batch: int, total_tiles: int,
                 sm_scale: float, tile_order: Optional[torch.Tensor] = None, impl: Optional[int] = None) -> None:
    """tile_order: optional int32 [total_tiles] schedule of the q tiles (heaviest first); impl: None = process default
    (PREFILL_IMPL), 0 = 4 the DMA-staged kernel, 2 its register-staged predecessor (include/msgl_hip.h; anything else is
    refused by the library).  tile_cu / total_tiles / tile_order are in units of prefill_q_tile(impl) query rows."""
    _need_cuda(out, q, k_cache, v_cache, page_table, seq_lens, cu_seqlens_q, tile_cu)
    if impl is None:
        impl = PREFILL_IMPL
    if tile_order is not None:
        _need_cuda(tile_order)
        assert tile_order.dtype == torch.int32 and tile_order.is_contiguous() and tile_order.numel() == total_tiles
    hq, d = q.shape[1], q.shape[2]
    hkv = k_cache.shape[1]
    assert q.stride(2) == 1 and q.stride(1) == d and out.stride(2) == 1 and out.stride(1) == d
    assert k_cache.stride(2) == 1 and v_cache.stride() == k_cache.stride()
    assert page_table.dtype == torch.int32 and page_table.stride(1) == 1
    check(
        lib().msgl_attn_prefill(
            out.data_ptr(), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), page_table.data_ptr(),
            page_table.stride(0), req_rows.data_ptr() if req_rows is not None else None, seq_lens.data_ptr(),
            cu_seqlens_q.data_ptr(), tile_cu.data_ptr(), batch, total_tiles, hq, hkv, d, q.stride(0),
            k_cache.stride(0), k_cache.stride(1), out.stride(0), float(sm_scale), _dt(q),
            tile_order.data_ptr() if tile_order is not None else None, int(impl), _stream(),
        ),
        "attn_prefill",
    )


# ---------------------------------------------------------------------------- sampling
def argmax_rows(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_cuda(logits)
    assert logits.dim() == 2 and logits.stride(1) == 1
    if out is None:
        out = torch.empty(logits.shape[0], dtype=torch.int32, device=logits.device)
    check(
        lib().msgl_argmax_rows(out.data_ptr(), logits.data_ptr(), logits.shape[0], logits.shape[1],
                               logits.stride(0), _logits_dt(logits), _stream()),
        "argmax_rows",
    )
    return out


def softmax_temperature(logits: torch.Tensor, temperatures: torch.Tensor) -> torch.Tensor:
    _need_cuda(logits, temperatures)
    assert logits.dim() == 2 and logits.stride(1) == 1 and temperatures.dtype == torch.float32
    probs = torch.empty(logits.shape, dtype=torch.float32, device=logits.device)
    check(
        lib().msgl_softmax_temperature(probs.data_ptr(), logits.data_ptr(), temperatures.data_ptr(),
                                       logits.shape[0], logits.shape[1], logits.stride(0), probs.stride(0),
                                       _logits_dt(logits), _stream()),
        "softmax_temperature",
    )
    return probs


def sample_top_k_top_p(probs: torch.Tensor, top_k: Optional[torch.Tensor], top_p: Optional[torch.Tensor],
                       seed: int, offset: int) -> torch.Tensor:
    _need_cuda(probs)
    assert probs.dim() == 2 and probs.stride(1) == 1 and probs.dtype == torch.float32
    out = torch.empty(probs.shape[0], dtype=torch.int32, device=probs.device)
    if top_k is not None:
        assert top_k.dtype == torch.int32 and top_k.is_cuda and top_k.numel() == probs.shape[0]
    if top_p is not None:
        assert top_p.dtype == torch.float32 and top_p.is_cuda and top_p.numel() == probs.shape[0]
    check(
        lib().msgl_sample_top_k_top_p(
            out.data_ptr(), probs.data_ptr(), top_k.data_ptr() if top_k is not None else None,
            top_p.data_ptr() if top_p is not None else None, probs.shape[0], probs.shape[1], probs.stride(0),
            seed & 0xFFFFFFFFFFFFFFFF, offset & 0xFFFFFFFFFFFFFFFF, _stream(),
        ),
        "sample_top_k_top_p",
    )
    return out


def sample_from_logits(logits: torch.Tensor, temperatures: torch.Tensor, seed: int, offset: int) -> torch.Tensor:
    """index ~ softmax(logits / T) per row, fused (no probs tensor); rows must be 16-byte aligned."""
    _need_cuda(logits, temperatures)
    assert logits.dim() == 2 and logits.stride(1) == 1 and temperatures.dtype == torch.float32
    assert temperatures.numel() == logits.shape[0]
    out = torch.empty(logits.shape[0], dtype=torch.int32, device=logits.device)
    check(
        lib().msgl_sample_from_logits(out.data_ptr(), logits.data_ptr(), temperatures.data_ptr(), logits.shape[0],
                                      logits.shape[1], logits.stride(0), _logits_dt(logits),
                                      seed & 0xFFFFFFFFFFFFFFFF, offset & 0xFFFFFFFFFFFFFFFF, _stream()),
        "sample_from_logits",
    )
    return out


# ---------------------------------------------------------------------------- projection GEMMs
_GEMM_WS: dict = {}
GEMM_WORKSPACE_BYTES = 128 << 20  # stream-K / split-K solutions need scratch; one buffer per device


def gemm_workspace(device: torch.device) -> torch.Tensor:
    key = torch.device(device).index or 0
    ws = _GEMM_WS.get(key)
    if ws is None:
        ws = torch.empty(GEMM_WORKSPACE_BYTES, dtype=torch.uint8, device=device)
        _GEMM_WS[key] = ws
    return ws


def _gemm_args(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor]):
    _need_cuda(x, w)
    assert x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1], (x.shape, w.shape)
    assert x.dtype == w.dtype and x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == x.dtype and out.is_cuda
    return out, M, N, K


# (device, M, N, K, ldx, ldw, dtype code) -> (k-slices, row tiles) of the hand-written weight-streaming kernel, for the
# shapes where skinny_tune() measured it faster than the library's best solution
_SKINNY_PLAN: dict = {}
SKINNY_MAX_M = 64
# a hand-written kernel replaces the library's pick only when the isolated timing says it is clearly faster:
# inside the captured step (other kernels' tails, cache state) a 1-3 % edge measured back to back does not
# survive (M = 256 qkv / o: 37.7 + 6.5 us reduce vs 39 / 46 us library, step time unchanged)
PLAN_MARGIN = 0.95


def skinny_supported(M: int, N: int, K: int) -> bool:
    return 1 <= M <= SKINNY_MAX_M and N % 16 == 0 and K % 64 == 0


ROWSTREAM_MAX_M = 8
ROWSTREAM_PLAIN, ROWSTREAM_SILU, ROWSTREAM_SILU_INTERLEAVED, ROWSTREAM_ADD_NORM = 0, 1, 2, 3


ROWSTREAM_VECTOR, ROWSTREAM_MATRIX = 0, 1  # variant: dot products on the vector units / on the matrix cores (4x4x4 MFMA)


@functools.lru_cache(maxsize=None)
def rowstream_supported(M: int, N: int, K: int, mode: int = 0, variant: int = 0) -> bool:
    """Whether msgl_rowstream_gemm_nt takes this shape (M <= 8; variant 0: K % 512 == 0, variant 1: K % 128 == 0 and
    N % 4 == 0; x + partial sums within the CU's LDS; mode 3: M <= 4 and a hidden size rmsnorm_wide_row_kernel runs)."""
    return bool(lib().msgl_rowstream_gemm_supported(M, N, K, mode, variant))


def rowstream_linear(x: torch.Tensor, w: torch.Tensor, depth: int = 16, out: Optional[torch.Tensor] = None, mode: int = 0,
                     res_in: Optional[torch.Tensor] = None, res_out: Optional[torch.Tensor] = None,
                     gamma: Optional[torch.Tensor] = None, eps: float = 0.0, variant: int = 0) -> torch.Tensor:
    """out[M, N] = f(x) @ w[N, K]^T by msgl_rowstream_gemm_nt (M <= 8; csrc/gemm_rowstream.hip): the weight matrix as
    one contiguous stream per CU.  mode 0: f(x) = x [M, K]; 1 / 2: x [M, 2 K] is a gate_up output (halves / interleaved)
    and f = silu(gate) * up; 3: f = fused_add_rmsnorm(x, res_in) with the new residual written to res_out.
    variant 0 / 1: products on the vector units (M = 1) / on the matrix cores (M = 2 .. 8); different summation orders."""
    _need_cuda(x, w)
    assert x.dim() == 2 and w.dim() == 2 and x.dtype == w.dtype and x.stride(1) == 1 and w.stride(1) == 1
    M, N, K = x.shape[0], w.shape[0], w.shape[1]
    assert x.shape[1] == (2 * K if mode in (ROWSTREAM_SILU, ROWSTREAM_SILU_INTERLEAVED) else K), (x.shape, w.shape, mode)
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == x.dtype and out.is_cuda
    _no_pending_slabs(x.device)
    if mode == ROWSTREAM_ADD_NORM:
        assert res_in is not None and res_out is not None and gamma is not None
        _need_cuda(res_in, res_out, gamma)
        assert res_in.shape == x.shape and res_out.shape == x.shape and res_in.stride(1) == 1 and res_out.stride(1) == 1
        assert res_in.dtype == x.dtype and res_out.dtype == x.dtype and gamma.dtype == x.dtype and gamma.numel() == K
        extra = (res_in.data_ptr(), res_out.data_ptr(), gamma.data_ptr(), float(eps), res_in.stride(0), res_out.stride(0))
    else:
        extra = (None, None, None, 0.0, 0, 0)
    check(
        lib().msgl_rowstream_gemm_nt(out.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0),
                                     out.stride(0), _dt(x), depth, mode, variant, *extra, _stream()),
        "rowstream_gemm_nt",
    )
    return out


def rowstream_planned(M: int, w: torch.Tensor, mode: int = 0):
    """(ring depth, variant) if the plan of `linear(x [M, K], w)` is the row-streaming kernel and that kernel also takes
    the shape with staging `mode`, else None: the decoder layer folds a neighbouring row kernel into the projection only
    where the projection already is this kernel (the folded launch then costs what the plain one does)."""
    if M > ROWSTREAM_MAX_M or not _SKINNY_PLAN:
        return None
    N, K = w.shape
    key = (w.device.index or 0, M, N, K, K, w.stride(0), _dt(w))
    plan = _SKINNY_PLAN.get(key)
    # (a shape whose plan is the row-owner kernel is NOT this kernel's, whatever the skinny table still holds: linear() looks at
    # _RO_PLAN first, and the fold must take the same kernel as the plain launch -- ADVICE r5)
    if not plan or plan[0] > 0 or _WSTREAM_PLAN.get(key) or _RO_PLAN.get(key) or not rowstream_supported(M, N, K, mode, -plan[0]):
        return None
    return plan[1], -plan[0]


def skinny_linear(x: torch.Tensor, w: torch.Tensor, slices: int, out: Optional[torch.Tensor] = None,
                  row_tiles: int = 1) -> torch.Tensor:
    """out[M, N] = x[M, K] @ w[N, K]^T by msgl_skinny_gemm_nt (M <= 64).  A plan (0, depth) / (-1, depth) names the
    row-streaming kernel of the same family (M <= 8, rowstream_linear, variant 0 / 1): the plan tables keep one (a, b) pair
    per shape."""
    if slices <= 0:
        return rowstream_linear(x, w, row_tiles, out, variant=-slices)
    out, M, N, K = _gemm_args(x, w, out)
    check(
        lib().msgl_skinny_gemm_nt(out.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0),
                                  out.stride(0), _dt(x), slices, row_tiles, _stream()),
        "skinny_gemm_nt",
    )
    return out


_SKINNY_SILU_PLAN: dict = {}  # (device, M, N, K, ldx, ldw, dtype code), w = interleaved gate_up -> (k-slices, row tiles) of the fused launch


def skinny_linear_silu(x: torch.Tensor, w: torch.Tensor, slices: int, out: Optional[torch.Tensor] = None,
                       row_tiles: int = 2) -> torch.Tensor:
    """out[M, N/2] = silu(x @ gate^T) * (x @ up^T) for a gate_up weight in interleave_gate_up order, one launch of
    msgl_skinny_gemm_silu_nt (M <= 64): the same bits as skinny_linear + silu_and_mul_interleaved."""
    _need_cuda(x, w)
    assert x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1] and x.dtype == w.dtype
    assert x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N // 2), dtype=x.dtype, device=x.device)
    assert out.shape == (M, N // 2) and out.stride(1) == 1 and out.dtype == x.dtype
    _no_pending_slabs(x.device)
    check(
        lib().msgl_skinny_gemm_silu_nt(out.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0),
                                       out.stride(0), _dt(x), slices, row_tiles, _stream()),
        "skinny_gemm_silu_nt",
    )
    return out


def skinny_silu_candidates(M: int, N: int, K: int):
    """(k-slices, row tiles per wave) settings of the fused gate_up + SiLU.mul launch (row tiles come in gate / up pairs)."""
    if N % 64:
        return []
    mt = 1 if M <= 16 else 2 if M <= 32 else 4
    out = []
    cap1 = 16 if mt <= 2 else 8
    out += [(sl, 1) for sl in (2, 4, 8, 16) if sl <= cap1 and sl // 2 <= K // 64]  # gate / up on the two halves of the waves
    for nt in (2, 4):
        cap = min(16 if mt * nt <= 2 else 8 if mt * nt <= 8 else 4, K // 64)
        out += [(sl, nt) for sl in (1, 2, 4, 8, 16) if sl <= cap]
    return out


def skinny_candidates(M: int, N: int, K: int):
    """(k-slices, row tiles per wave) settings the kernel accepts for this shape (csrc/gemm_skinny.hip)."""
    mt = 1 if M <= 16 else 2 if M <= 32 else 4
    out = []
    for nt in (1, 2, 4):
        if N % (16 * nt):
            continue
        cap = min(16 if mt * nt <= 2 else 8 if mt * nt <= 8 else 4, K // 64)
        out += [(sl, nt) for sl in (1, 2, 4, 8, 16) if sl <= cap]
    if M <= ROWSTREAM_MAX_M and os.environ.get("MSGL_DISABLE_ROWSTREAM") != "1":
        # the row-streaming kernel, 8 / 16 loads in flight per lane: (0, depth) on the vector units, (-1, depth) on the matrix cores
        out += [(-v, d) for v in (ROWSTREAM_VECTOR, ROWSTREAM_MATRIX) if rowstream_supported(M, N, K, 0, v) for d in (8, 16)]
    return out


# what a folded row kernel is worth to the step when the projection behind it is the row-streaming kernel: the launch that
# disappears minus the staging work that appears, measured inside the captured B = 1 step (tools/small_batch_ab.py:
# 0.8-0.9 us per fold; the search below works on back-to-back times, which also have rowstream qkv 0.7 us behind the
# matrix-core kernel where the step has it ahead)
ROWSTREAM_FOLD_BONUS_US = 1.5
ROWSTREAM_FOLD_NORM_MAX_M, ROWSTREAM_FOLD_ACT_MAX_M = 2, 1  # rows up to which the decoder layer folds (model.DenseDecoder.forward)


def skinny_tune(x: torch.Tensor, weights, library_us: float, iters: int = 8, fold_mode: Optional[int] = None) -> dict:
    """Time the weight-streaming kernels over their settings on rotating weights (as gemm_tune does) and plan the best for
    this shape if it beats `library_us`.  fold_mode: the staging mode (ROWSTREAM_ADD_NORM / ROWSTREAM_SILU*) by which the
    decoder layer would fold the row kernel in front of this projection if the row-streaming kernel is planned; settings
    that can are credited ROWSTREAM_FOLD_BONUS_US.  Synchronises; call before capture."""
    weights = list(weights)
    w0 = weights[0]
    M, K = x.shape
    N = w0.shape[0]
    res = dict(M=M, N=N, K=K, library_us=library_us, skinny_us=None, slices=0, row_tiles=0, used=False)
    if not skinny_supported(M, N, K):
        return res
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)

    def time_us(sl, nt, rounds):
        skinny_linear(x, w0, sl, out, nt)  # warm-up
        ts = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                skinny_linear(x, weights[(i + 1) % len(weights)], sl, out, nt)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / iters)
        return min(ts)

    def credit(sl):
        folds = fold_mode is not None and sl <= 0 and rowstream_supported(M, N, K, fold_mode, -sl)
        return ROWSTREAM_FOLD_BONUS_US if folds else 0.0

    ranked = sorted((time_us(sl, nt, 1) - credit(sl), sl, nt) for sl, nt in skinny_candidates(M, N, K))
    timed = [(time_us(sl, nt, 3), sl, nt) for _, sl, nt in ranked[:5]]  # re-time the best few
    best = min(timed, key=lambda t: t[0] - credit(t[1]))
    # skinny_us is the MEASURED time of the chosen setting; fold_credit_us is what the ranking subtracted from it because the
    # decoder layer folds a row kernel into its staging pass (a deliberate trade, not a regression, when skinny_us is up to
    # that much above the library's best)
    res.update(skinny_us=best[0], slices=best[1], row_tiles=best[2], best_plain_us=min(t[0] for t in timed),
               rowstream_us=min((t[0] for t in timed if t[1] <= 0), default=None), fold_credit_us=credit(best[1]))
    key = (x.device.index or 0, M, N, K, x.stride(0), w0.stride(0), _dt(x))
    if best[0] - credit(best[1]) < PLAN_MARGIN * library_us:
        _SKINNY_PLAN[key] = (best[1], best[2])
        res["used"] = True
    else:
        _SKINNY_PLAN.pop(key, None)
    return res


# ---- mid-size decode batches: LDS-shared activation tile (csrc/gemm_wstream.hip)
_WSTREAM_PLAN: dict = {}  # (device, M, N, K, ldx, ldw, dtype code) -> (row tiles, k splits)
WSTREAM_MAX_M = 256


def wstream_supported(M: int, N: int, K: int) -> bool:
    return 1 <= M <= WSTREAM_MAX_M and N % 128 == 0 and K % 64 == 0


def wstream_linear(x: torch.Tensor, w: torch.Tensor, row_tiles: int, k_splits: int,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, N] = x[M, K] @ w[N, K]^T by msgl_wstream_gemm_nt (M <= 256); split-K scratch = the GEMM workspace."""
    out, M, N, K = _gemm_args(x, w, out)
    ws = gemm_workspace(x.device)
    check(
        lib().msgl_wstream_gemm_nt(out.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0),
                                   out.stride(0), _dt(x), row_tiles, k_splits, ws.data_ptr(), ws.numel(), _stream()),
        "wstream_gemm_nt",
    )
    return out


def wstream_candidates(M: int, N: int, K: int):
    """(row tiles, k splits) settings worth timing: enough workgroups to occupy the chip, scratch within the
    GEMM workspace."""
    out = []
    for nt in (1, 2):
        if N % (128 * nt) or (nt == 2 and M > 128):  # two row tiles per wave at M > 128 spill: not built
            continue
        groups = N // (128 * nt)
        for ks in (1, 2, 3, 4, 6, 8, 12, 16):
            if ks > K // 64 or (ks > 1 and ks * M * N * 4 > GEMM_WORKSPACE_BYTES):
                continue
            if ks > 1 and groups * (ks // 2 if ks > 2 else 1) >= 1024:  # already far more workgroups than CUs
                continue
            out.append((nt, ks))
    return out


def wstream_tune(x: torch.Tensor, weights, incumbent_us: float, iters: int = 8) -> dict:
    """Time the LDS-shared weight-streaming kernel over (row tiles, k splits) on rotating weights and plan it for
    this shape if it beats `incumbent_us` (the best of the library and the skinny kernel)."""
    weights = list(weights)
    w0 = weights[0]
    M, K = x.shape
    N = w0.shape[0]
    res = dict(M=M, N=N, K=K, incumbent_us=incumbent_us, wstream_us=None, row_tiles=0, k_splits=0, used=False)
    if not wstream_supported(M, N, K):
        return res
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)

    def time_us(nt, ks, rounds):
        wstream_linear(x, w0, nt, ks, out)  # warm-up
        ts = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                wstream_linear(x, weights[(i + 1) % len(weights)], nt, ks, out)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / iters)
        return min(ts)

    cands = wstream_candidates(M, N, K)
    if not cands:
        return res
    ranked = sorted((time_us(nt, ks, 1), nt, ks) for nt, ks in cands)
    best = min((time_us(nt, ks, 3), nt, ks) for _, nt, ks in ranked[:4])
    res.update(wstream_us=best[0], row_tiles=best[1], k_splits=best[2], all={f"{nt}x{ks}": round(t, 1) for t, nt, ks in ranked})
    key = (x.device.index or 0, M, N, K, x.stride(0), w0.stride(0), _dt(x))
    if best[0] < PLAN_MARGIN * incumbent_us:
        _WSTREAM_PLAN[key] = (best[1], best[2])
        _SKINNY_PLAN.pop(key, None)
        res["used"] = True
    else:
        _WSTREAM_PLAN.pop(key, None)
    return res


# ---- full decode batches: one workgroup per CU, LDS-DMA ring, 32x32x16 MFMA (csrc/gemm_m256.hip)
_M256_PLAN: dict = {}  # (device, M, N, K, ldx, ldw, dtype code) -> (grid, full, tail_split, impl); impl 0 = gemm_m256.hip, 1 = gemm_g3.hip
_FUSED_SILU_PLAN: dict = {}  # same key (w = interleaved gate_up) -> (grid, full, tail_split) of the fused g3 launch
M256_MIN_M, M256_MAX_M = 129, 256


def m256_supported(M: int, N: int, K: int) -> bool:
    return M256_MIN_M <= M <= M256_MAX_M and N % 128 == 0 and N >= 128 and K % 64 == 0 and K >= 64


def m256_linear(x: torch.Tensor, w: torch.Tensor, grid: int, full: int, tail_split: int,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, N] = x[M, K] @ w[N, K]^T by msgl_m256_gemm_nt with the plan (grid, full, tail_split)."""
    out, M, N, K = _gemm_args(x, w, out)
    ws = gemm_workspace(x.device)
    check(
        lib().msgl_m256_gemm_nt(out.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0),
                                out.stride(0), _dt(x), grid, full, tail_split, ws.data_ptr(), ws.numel(), _stream()),
        "m256_gemm_nt",
    )
    return out


def m256_candidates(M: int, N: int, K: int, cus: int):
    """(grid, full, tail_split) plans worth timing.  grid = CU count (one resident workgroup per CU); whole tiles
    in rounds of `grid`, the remaining tiles cut into k-slices so that remainder x slices fills the grid; without a
    whole round, k-slices so that tiles x slices is about one or two grids."""
    tiles, nsteps = N // 128, K // 64
    out = []

    def add(full, split):
        split = max(1, min(split, nsteps, 64))
        if split > 1 and split * M * (tiles - full) * 128 * 4 > GEMM_WORKSPACE_BYTES:
            return
        if (cus, full, split) not in out:
            out.append((cus, full, split))

    full = tiles // cus * cus
    rest = tiles - full
    add(tiles, 1)                                    # every tile whole (rounds of `grid`)
    if rest:
        for div in (1, 2, 4):                        # remainder spread over all / half / a quarter of the grid
            if cus // (rest * div) >= 2:
                add(full, cus // (rest * div))
    if full == 0:
        for target in (cus, 2 * cus, 3 * cus):
            add(0, target // tiles)
            add(0, -(-target // tiles))
    return out


def _time_launches_us(fn, weights, iters: int, rounds: int) -> float:
    """min over `rounds` of the mean time of `iters` launches of fn(w) on rotating weights (events on this stream)."""
    fn(weights[0])  # warm-up
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(weights[(i + 1) % len(weights)])
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return min(ts)


def full_batch_linear(x: torch.Tensor, w: torch.Tensor, plan, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The planned full-batch kernel: plan = (grid, full, tail_split, impl)."""
    if len(plan) > 3 and plan[3] == 1:
        return g3_linear(x, w, plan[0], plan[1], plan[2], out)
    return m256_linear(x, w, plan[0], plan[1], plan[2], out)


def m256_tune(x: torch.Tensor, weights, incumbent_us: float, iters: int = 8) -> dict:
    """Time the plans of m256_candidates with both full-batch kernels (register-staged gemm_m256.hip, loader / matrix
    wave gemm_g3.hip) on rotating weights and keep the fastest for this shape if it beats `incumbent_us` (the best of
    the library and the other hand-written kernels) by PLAN_MARGIN."""
    weights = list(weights)
    w0 = weights[0]
    M, K = x.shape
    N = w0.shape[0]
    res = dict(M=M, N=N, K=K, incumbent_us=incumbent_us, m256_us=None, plan=None, used=False)
    if not m256_supported(M, N, K):
        return res
    cus = int(lib().msgl_device_cu_count())
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    impls = (0, 1) if os.environ.get("MSGL_DISABLE_G3") != "1" else (0,)
    cands = [p + (impl,) for p in m256_candidates(M, N, K, cus) for impl in impls]
    ranked = sorted((_time_launches_us(lambda w: full_batch_linear(x, w, p, out), weights, iters, 1), p) for p in cands)
    best = min((_time_launches_us(lambda w: full_batch_linear(x, w, p, out), weights, iters, 3), p) for _, p in ranked[:4])
    res.update(m256_us=best[0], plan=best[1], all={"/".join(map(str, p)): round(t, 1) for t, p in ranked})
    key = (x.device.index or 0, M, N, K, x.stride(0), w0.stride(0), _dt(x))
    if best[0] < PLAN_MARGIN * incumbent_us:
        _M256_PLAN[key] = best[1]
        res["used"] = True
    else:
        _M256_PLAN.pop(key, None)
    return res


def fused_silu_tune(x: torch.Tensor, weights, iters: int = 8) -> dict:
    """gate_up in interleave_gate_up order: time `linear` (whatever plan the search above left for the shape) followed
    by silu_and_mul_interleaved against the fused-epilogue launches of gemm_g3.hip, plan the fused one if faster."""
    weights = list(weights)
    w0 = weights[0]
    M, K = x.shape
    N = w0.shape[0]
    res = dict(M=M, N=N, K=K, unfused_us=None, fused_us=None, plan=None, used=False)
    key = (x.device.index or 0, M, N, K, x.stride(0), w0.stride(0), _dt(x))
    _FUSED_SILU_PLAN.pop(key, None)
    _SKINNY_SILU_PLAN.pop(key, None)
    if M <= SKINNY_MAX_M and skinny_supported(M, N, K) and N % 64 == 0 and os.environ.get("MSGL_DISABLE_SKINNY_SILU") != "1":
        # decode-sized batch: the weight-streaming kernel with the activation in its epilogue against whatever `linear`
        # is planned to do for the shape + the activation kernel
        half = torch.empty((M, N // 2), dtype=x.dtype, device=x.device)
        full = torch.empty((M, N), dtype=x.dtype, device=x.device)  # (no allocation inside the timed pair: it is not there under replay)
        res["unfused_us"] = _time_launches_us(lambda w: silu_and_mul_interleaved(linear(x, w, full), half), weights, iters, 3)
        ranked = sorted((_time_launches_us(lambda w: skinny_linear_silu(x, w, sl, half, nt), weights, iters, 1), (sl, nt))
                        for sl, nt in skinny_silu_candidates(M, N, K))
        best = min((_time_launches_us(lambda w: skinny_linear_silu(x, w, p[0], half, p[1]), weights, iters, 3), p)
                   for _, p in ranked[:4])
        res.update(fused_us=best[0], plan=best[1], kind="skinny")
        if best[0] < res["unfused_us"]:  # one launch less in front of the same consumer: no margin asked
            _SKINNY_SILU_PLAN[key] = best[1]
            res["used"] = True
        return res
    if not (m256_supported(M, N, K) and (N // 2) % 64 == 0) or os.environ.get("MSGL_DISABLE_G3") == "1":
        return res
    cus = int(lib().msgl_device_cu_count())
    half = torch.empty((M, N // 2), dtype=x.dtype, device=x.device)
    res["unfused_us"] = _time_launches_us(lambda w: silu_and_mul_interleaved(linear(x, w), half), weights, iters, 3)
    ranked = sorted((_time_launches_us(lambda w: g3_linear(x, w, *p, out=half, silu=True), weights, iters, 1), p)
                    for p in m256_candidates(M, N, K, cus))
    best = min((_time_launches_us(lambda w: g3_linear(x, w, *p, out=half, silu=True), weights, iters, 3), p)
               for _, p in ranked[:3])
    res.update(fused_us=best[0], plan=best[1], all={"/".join(map(str, p)): round(t, 1) for t, p in ranked})
    # MSGL_FORCE_FUSED_SILU=1 / =0: A/B switch (inside a captured step the two rank differently than back to back)
    force = os.environ.get("MSGL_FORCE_FUSED_SILU")
    if force != "0" and (force == "1" or best[0] < PLAN_MARGIN * res["unfused_us"]):
        _FUSED_SILU_PLAN[key] = best[1]
        res["used"] = True
    return res


def linear_silu(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """silu(x @ gate^T) * (x @ up^T) for a gate_up weight in interleave_gate_up order (P/models/utils.py:45-51:
    gate_up_proj then act_fn): one fused launch where fused_silu_tune planned it, else linear + the activation kernel."""
    M, K = x.shape
    N = w.shape[0]
    if _RO_SILU_PLAN:
        plan = _RO_SILU_PLAN.get((x.device.index or 0, M, N, K, x.stride(0), w.stride(0), _dt(x)))
        if plan:
            return ro_linear(x, w, plan[0], 1, out, silu=True)
    if M <= SKINNY_MAX_M and _SKINNY_SILU_PLAN:
        plan = _SKINNY_SILU_PLAN.get((x.device.index or 0, M, N, K, x.stride(0), w.stride(0), _dt(x)))
        if plan:
            return skinny_linear_silu(x, w, plan[0], out, plan[1])
    if M >= M256_MIN_M and _FUSED_SILU_PLAN:
        plan = _FUSED_SILU_PLAN.get((x.device.index or 0, M, N, K, x.stride(0), w.stride(0), _dt(x)))
        if plan:
            return g3_linear(x, w, plan[0], plan[1], plan[2], out, silu=True)
    return silu_and_mul_interleaved(linear(x, w), out)


# ---- generation 3 of the full-batch kernel: loader waves + matrix waves (csrc/gemm_g3.hip); same plan triples
G3_SILU, G3_SLABS_ONLY = 1, 2


def g3_linear(x: torch.Tensor, w: torch.Tensor, grid: int, full: int, tail_split: int,
              out: Optional[torch.Tensor] = None, silu: bool = False, variant: int = 0) -> torch.Tensor:
    """out[M, N] = x[M, K] @ w[N, K]^T by msgl_g3_gemm_nt with the plan (grid, full, tail_split).  silu=True: `w` is a
    gate_up weight in interleave_gate_up order and out[M, N/2] = silu(gate) * up."""
    _need_cuda(x, w)
    assert x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1] and x.dtype == w.dtype
    assert x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    cols = N // 2 if silu else N
    if out is None:
        out = torch.empty((M, cols), dtype=x.dtype, device=x.device)
    assert out.shape == (M, cols) and out.stride(1) == 1 and out.dtype == x.dtype
    _no_pending_slabs(x.device)
    ws = gemm_workspace(x.device)
    check(
        lib().msgl_g3_gemm_nt(out.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0),
                              out.stride(0), _dt(x), grid, full, tail_split, (G3_SILU if silu else 0) | (variant << 8),
                              ws.data_ptr(), ws.numel(), _stream()),
        "g3_gemm_nt",
    )
    return out


# ---- row-owner generation (csrc/gemm_ro.hip), RO_MIN_M = 3 <= M <= 256: balanced 16-row-unit tiles x k-slices, plan = (tiles, slices)
_RO_PLAN: dict = {}       # plan key -> (tiles, slices)
_RO_SILU_PLAN: dict = {}  # plan key (w = interleaved gate_up) -> (tiles, 1): projection + SiLU.mul in one launch
RO_SILU, RO_SLABS_ONLY = 1, 2
RO_MIN_M, RO_MAX_M = 3, 256


def ro_max_units(M: int) -> int:
    """Units (16 weight rows) a tile may hold at this batch size (the accumulator budget of csrc/gemm_ro.hip)."""
    return 9 if M > 128 else 18


def ro_supported(M: int, N: int, K: int) -> bool:
    return RO_MIN_M <= M <= RO_MAX_M and N % 16 == 0 and N >= 16 and K % 64 == 0 and K >= 64


def ro_linear(x: torch.Tensor, w: torch.Tensor, tiles: int, slices: int, out: Optional[torch.Tensor] = None,
              silu: bool = False, ablate: int = 0) -> torch.Tensor:
    """out[M, N] = x[M, K] @ w[N, K]^T by msgl_ro_gemm_nt with the plan (tiles, slices).  silu=True (slices == 1): `w` is a
    gate_up weight in interleave_gate_up order and out[M, N/2] = silu(gate) * up."""
    _need_cuda(x, w)
    assert x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1] and x.dtype == w.dtype
    assert x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    cols = N // 2 if silu else N
    if out is None:
        out = torch.empty((M, cols), dtype=x.dtype, device=x.device)
    assert out.shape == (M, cols) and out.stride(1) == 1 and out.dtype == x.dtype
    _no_pending_slabs(x.device)
    ws = gemm_workspace(x.device)
    check(
        lib().msgl_ro_gemm_nt(out.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0), out.stride(0),
                              _dt(x), tiles, slices, (RO_SILU if silu else 0) | (ablate << 8), ws.data_ptr(), ws.numel(), _stream()),
        "ro_gemm_nt",
    )
    return out


def ro_candidates(M: int, N: int, K: int, cus: int, silu: bool = False):
    """(tiles, slices) plans worth timing: items = tiles x slices close to one or two rounds of the CUs, every tile within the
    accumulator budget; fewer, wider tiles (more k-slices) move less x per weight byte but more partial sums."""
    units, nsteps, umax = N // 16, K // 64, ro_max_units(M)
    min_tiles = -(-units // umax)
    out = []

    def add(tiles, slices):
        tiles = max(min_tiles, min(tiles, units))
        slices = max(1, min(slices, nsteps, 64))
        if silu and slices != 1:
            return
        if slices > 1 and slices * M * N * 4 > GEMM_WORKSPACE_BYTES:
            return
        if (tiles, slices) not in out:
            out.append((tiles, slices))

    r0 = -(-min_tiles // cus)  # the fewest whole rounds of the CUs the accumulator budget allows (LM head: 5)
    for rounds in sorted({1, 2, 3, r0, r0 + 1}):
        items = rounds * cus
        for slices in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
            tiles = items // slices
            if tiles < min_tiles:
                continue
            if slices == 1 or tiles <= 4 * min_tiles:  # k-slicing only while the tiles stay reasonably wide
                add(tiles, slices)
    if not out:
        add(min_tiles, 1)
    return out


def ro_tune(x: torch.Tensor, weights, incumbent_us: float, iters: int = 8) -> dict:
    """Time the plans of ro_candidates on rotating weights, keep the fastest for this shape if it beats `incumbent_us` (the
    best of the library and the other hand-written kernels) by PLAN_MARGIN."""
    weights = list(weights)
    w0 = weights[0]
    M, K = x.shape
    N = w0.shape[0]
    res = dict(M=M, N=N, K=K, incumbent_us=incumbent_us, ro_us=None, plan=None, used=False)
    key = (x.device.index or 0, M, N, K, x.stride(0), w0.stride(0), _dt(x))
    _RO_PLAN.pop(key, None)
    if not ro_supported(M, N, K) or os.environ.get("MSGL_DISABLE_RO") == "1":
        return res
    cus = int(lib().msgl_device_cu_count())
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    ranked = sorted((_time_launches_us(lambda w: ro_linear(x, w, p[0], p[1], out), weights, iters, 1), p)
                    for p in ro_candidates(M, N, K, cus))
    best = min((_time_launches_us(lambda w: ro_linear(x, w, p[0], p[1], out), weights, iters, 3), p) for _, p in ranked[:4])
    res.update(ro_us=best[0], plan=best[1], all={"/".join(map(str, p)): round(t, 1) for t, p in ranked})
    if best[0] < PLAN_MARGIN * incumbent_us:
        _RO_PLAN[key] = best[1]
        res["used"] = True
    return res


def ro_silu_tune(x: torch.Tensor, weights, unfused_us: float, iters: int = 8) -> dict:
    """gate_up in interleave_gate_up order: the row-owner launch with the activation in its epilogue against `unfused_us`
    (whatever the search left for projection + activation).  One launch less in front of the same consumer: no margin asked."""
    weights = list(weights)
    w0 = weights[0]
    M, K = x.shape
    N = w0.shape[0]
    res = dict(M=M, N=N, K=K, unfused_us=unfused_us, fused_us=None, plan=None, used=False)
    key = (x.device.index or 0, M, N, K, x.stride(0), w0.stride(0), _dt(x))
    _RO_SILU_PLAN.pop(key, None)
    if not (ro_supported(M, N, K) and N % 64 == 0) or os.environ.get("MSGL_DISABLE_RO") == "1":
        return res
    cus = int(lib().msgl_device_cu_count())
    half = torch.empty((M, N // 2), dtype=x.dtype, device=x.device)
    ranked = sorted((_time_launches_us(lambda w: ro_linear(x, w, p[0], 1, half, silu=True), weights, iters, 1), p)
                    for p in ro_candidates(M, N, K, cus, silu=True))
    best = min((_time_launches_us(lambda w: ro_linear(x, w, p[0], 1, half, silu=True), weights, iters, 3), p) for _, p in ranked[:3])
    res.update(fused_us=best[0], plan=best[1], all={"/".join(map(str, p)): round(t, 1) for t, p in ranked})
    if best[0] < unfused_us:
        _RO_SILU_PLAN[key] = best[1]
        res["used"] = True
    return res


def silu_pair_us(x: torch.Tensor, weights, iters: int = 8) -> float:
    """Back-to-back time of what linear_silu does for this shape today (its planned fused launch, or projection + activation)."""
    weights = list(weights)
    N = weights[0].shape[0]
    half = torch.empty((x.shape[0], N // 2), dtype=x.dtype, device=x.device)
    return _time_launches_us(lambda w: linear_silu(x, w, half), weights, iters, 3)


def linear(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, N] = x[M, K] @ w[N, K]^T (the reference's `F.linear`, P/layers/linear.py:32): the hand-written
    weight-streaming kernel where skinny_tune() planned it (decode batches <= 64), else msgl_gemm_nt with the
    tuned library solution for the shape if gemm_tune() ran, else the library's heuristic."""
    out, M, N, K = _gemm_args(x, w, out)
    if M == 0:
        return out
    _no_pending_slabs(x.device)
    if _RO_PLAN:
        plan = _RO_PLAN.get((x.device.index or 0, M, N, K, x.stride(0), w.stride(0), _dt(x)))
        if plan:
            return ro_linear(x, w, plan[0], plan[1], out)
    if M >= M256_MIN_M and _M256_PLAN:
        plan = _M256_PLAN.get((x.device.index or 0, M, N, K, x.stride(0), w.stride(0), _dt(x)))
        if plan:
            return full_batch_linear(x, w, plan, out)
    if M <= WSTREAM_MAX_M and _WSTREAM_PLAN:
        plan = _WSTREAM_PLAN.get((x.device.index or 0, M, N, K, x.stride(0), w.stride(0), _dt(x)))
        if plan:
            return wstream_linear(x, w, plan[0], plan[1], out)
    if M <= SKINNY_MAX_M and _SKINNY_PLAN:
        plan = _SKINNY_PLAN.get((x.device.index or 0, M, N, K, x.stride(0), w.stride(0), _dt(x)))
        if plan:
            return skinny_linear(x, w, plan[0], out, plan[1])
    ws = gemm_workspace(x.device)
    _lib.check_gemm(
        _lib.gemm_lib().msgl_gemm_nt(out.data_ptr(), x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0),
                                     out.stride(0), _dt(x), ws.data_ptr(), ws.numel(), _stream()),
        "gemm_nt",
    )
    return out


def linear_slabs(x: torch.Tensor, w: torch.Tensor):
    """`linear` for a projection whose output feeds fused_add_rmsnorm directly: returns (out, slabs).  When the
    shape's plan is the k-sliced full-batch kernel (or, mid-size batches, the k-split weight-streaming kernel), the reduce
    launch is left to the norm (slabs is a Slabs, `out` is allocated but NOT yet written: pass both to
    fused_add_rmsnorm_slabs); otherwise (out, None) = linear(x, w)."""
    if _RO_PLAN:
        M, K, N = x.shape[0], x.shape[1], w.shape[0]
        plan = _RO_PLAN.get((x.device.index or 0, M, N, K, x.stride(0), w.stride(0), _dt(x)))
        if plan:
            if plan[1] == 1:
                return linear(x, w), None
            _no_pending_slabs(x.device)
            out, M, N, K = _gemm_args(x, w, None)
            ws = gemm_workspace(x.device)
            check(
                lib().msgl_ro_gemm_nt(None, x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0), N, _dt(x), plan[0],
                                      plan[1], RO_SLABS_ONLY, ws.data_ptr(), ws.numel(), _stream()),
                "ro_gemm_nt(slabs)",
            )
            slabs = Slabs(ws.data_ptr(), plan[1], M, N, x.device.index or 0)
            _PENDING_SLABS[slabs.device] = slabs
            return out, slabs
    if x.shape[0] >= M256_MIN_M and _M256_PLAN:
        _no_pending_slabs(x.device)
        out, M, N, K = _gemm_args(x, w, None)
        plan = _M256_PLAN.get((x.device.index or 0, M, N, K, x.stride(0), w.stride(0), _dt(x)))
        if plan and plan[1] == 0 and plan[2] > 1:
            ws = gemm_workspace(x.device)
            if len(plan) > 3 and plan[3] == 1:
                check(
                    lib().msgl_g3_gemm_nt(None, x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0), N, _dt(x),
                                          plan[0], 0, plan[2], G3_SLABS_ONLY, ws.data_ptr(), ws.numel(), _stream()),
                    "g3_gemm_nt(slabs)",
                )
            else:
                check(
                    lib().msgl_m256_gemm_slabs_nt(x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0), _dt(x),
                                                  plan[0], plan[2], ws.data_ptr(), ws.numel(), _stream()),
                    "m256_gemm_slabs_nt",
                )
            slabs = Slabs(ws.data_ptr(), plan[2], M, N, x.device.index or 0)
            _PENDING_SLABS[slabs.device] = slabs
            return out, slabs
        if plan:  # a full-batch plan that writes its own output (whole tiles / one slice)
            return linear(x, w, out), None
    if x.shape[0] <= WSTREAM_MAX_M and _WSTREAM_PLAN and os.environ.get("MSGL_DISABLE_WSTREAM_SLABS") != "1":
        # mid-size decode batches: the LDS-shared weight-streaming kernel's k splits, same hand-off (its reduce launch,
        # ~5.5 us x 3 per layer at B = 64 .. 128, is left to the norm / qk pass that reads the output next)
        out, M, N, K = _gemm_args(x, w, None)
        plan = _WSTREAM_PLAN.get((x.device.index or 0, M, N, K, x.stride(0), w.stride(0), _dt(x)))
        if plan and plan[1] > 1 and not (M >= M256_MIN_M and _M256_PLAN.get((x.device.index or 0, M, N, K, x.stride(0), w.stride(0), _dt(x)))):
            _no_pending_slabs(x.device)
            ws = gemm_workspace(x.device)
            check(
                lib().msgl_wstream_gemm_slabs_nt(x.data_ptr(), w.data_ptr(), M, N, K, x.stride(0), w.stride(0), _dt(x),
                                                 plan[0], plan[1], ws.data_ptr(), ws.numel(), _stream()),
                "wstream_gemm_slabs_nt",
            )
            slabs = Slabs(ws.data_ptr(), plan[1], M, N, x.device.index or 0)
            _PENDING_SLABS[slabs.device] = slabs
            return out, slabs
        return linear(x, w, out), None
    return linear(x, w), None


# ---- candidates for re-ranking inside the captured step (engine.Engine.refine_plans_in_graph)
_CANDIDATES: dict = {}  # plan key -> {"name": str, "cands": [(label, spec), ...]}; spec: ("lib", i) | ("hand", plan4) | ("fused", plan3)


def plan_key(x: torch.Tensor, w: torch.Tensor):
    return (x.device.index or 0, x.shape[0], w.shape[0], x.shape[1], x.stride(0), w.stride(0), _dt(x))


def gemm_finalists(x: torch.Tensor, w: torch.Tensor, out_ld: Optional[int] = None) -> list:
    """Times (us, back to back) of the best library candidates of the last gemm_tune of this shape, fastest first."""
    import ctypes as C

    M, K = x.shape
    N = w.shape[0]
    buf = (C.c_float * 8)()
    n = _lib.gemm_lib().msgl_gemm_finalists(M, N, K, x.stride(0), w.stride(0), out_ld or N, _dt(x), buf, 8)
    _lib.check_gemm(n, "gemm_finalists")
    return [float(buf[i]) for i in range(min(n, 8))]


def register_candidates(name: str, x: torch.Tensor, w: torch.Tensor, report: dict) -> None:
    """After the back-to-back search of one (batch size, projection): remember the few best of every kind for the
    in-graph re-ranking.  Only full-batch shapes (the kinds differ by several percent there)."""
    key = plan_key(x, w)
    M = x.shape[0]
    if M < M256_MIN_M:
        return
    cands = []
    for i, us in enumerate(gemm_finalists(x, w)[:3]):
        cands.append((f"library #{i} ({us:.1f} us)", ("lib", i)))
    allp = report.get("m256_all") or {}
    for lab, us in sorted(allp.items(), key=lambda kv: kv[1])[:2]:
        plan = tuple(int(v) for v in lab.split("/"))
        cands.append((f"{('m256', 'g3')[plan[3]]} {plan[:3]} ({us:.1f} us)", ("hand", plan)))
    for lab, us in sorted((report.get("silu_fused_all") or {}).items(), key=lambda kv: kv[1])[:2]:
        cands.append((f"g3 fused SiLU.mul {lab} ({us:.1f} us)", ("fused", tuple(int(v) for v in lab.split("/")))))
    for lab, us in sorted((report.get("ro_all") or {}).items(), key=lambda kv: kv[1])[:2]:
        cands.append((f"ro {lab} ({us:.1f} us)", ("ro", tuple(int(v) for v in lab.split("/")))))
    for lab, us in sorted((report.get("ro_silu_all") or {}).items(), key=lambda kv: kv[1])[:1]:
        cands.append((f"ro fused SiLU.mul {lab} ({us:.1f} us)", ("ro_silu", tuple(int(v) for v in lab.split("/")))))
    _CANDIDATES[key] = dict(name=name, cands=cands, out_ld=w.shape[0])


def current_candidate(key) -> str:
    if key in _RO_SILU_PLAN:
        return f"ro fused SiLU.mul {_RO_SILU_PLAN[key]}"
    if key in _RO_PLAN and key not in _FUSED_SILU_PLAN:
        return f"ro {_RO_PLAN[key]}"
    if key in _SKINNY_SILU_PLAN:
        return f"skinny fused SiLU.mul {_SKINNY_SILU_PLAN[key]}"
    if key in _FUSED_SILU_PLAN:
        return f"g3 fused SiLU.mul {_FUSED_SILU_PLAN[key]}"
    if key in _M256_PLAN:
        p = _M256_PLAN[key]
        return f"{('m256', 'g3')[p[3] if len(p) > 3 else 0]} {tuple(p[:3])}"
    return "library (search's pick)"


def apply_candidate(key, spec) -> None:
    """Make `spec` the plan of the shape `key` (see _CANDIDATES)."""
    kind, arg = spec
    dev, M, N, K, ldx, ldw, dt = key
    if kind in ("lib", "hand", "ro", "fused", "ro_silu"):
        # linear() / linear_silu() consult the skinny / weight-streaming tables BEFORE these: a candidate timed under its own
        # label must be the kernel that runs (the re-ranking runs at the largest graph batch, where they are empty)
        _SKINNY_SILU_PLAN.pop(key, None)
        _SKINNY_PLAN.pop(key, None)
        if kind != "hand":
            _WSTREAM_PLAN.pop(key, None)
    if kind == "lib":
        _M256_PLAN.pop(key, None)
        _WSTREAM_PLAN.pop(key, None)
        _FUSED_SILU_PLAN.pop(key, None)
        _RO_PLAN.pop(key, None)
        _RO_SILU_PLAN.pop(key, None)
        _lib.check_gemm(_lib.gemm_lib().msgl_gemm_select_finalist(M, N, K, ldx, ldw, _CANDIDATES[key]["out_ld"], dt, int(arg)),
                        "gemm_select_finalist")
    elif kind == "hand":
        _FUSED_SILU_PLAN.pop(key, None)
        _RO_PLAN.pop(key, None)
        _RO_SILU_PLAN.pop(key, None)
        _M256_PLAN[key] = tuple(arg)
    elif kind == "fused":
        _RO_SILU_PLAN.pop(key, None)
        _FUSED_SILU_PLAN[key] = tuple(arg)
    elif kind == "ro":
        _FUSED_SILU_PLAN.pop(key, None)
        _RO_SILU_PLAN.pop(key, None)
        _RO_PLAN[key] = tuple(arg)
    elif kind == "ro_silu":
        _RO_SILU_PLAN[key] = tuple(arg)
    else:
        raise ValueError(spec)


def snapshot_plan(key):
    """Opaque state of the shape's hand-written plans (the library's pick is restored by index 0 of its finalists)."""
    return (_M256_PLAN.get(key), _WSTREAM_PLAN.get(key), _FUSED_SILU_PLAN.get(key), _RO_PLAN.get(key), _RO_SILU_PLAN.get(key),
            _SKINNY_PLAN.get(key), _SKINNY_SILU_PLAN.get(key))


def restore_search_pick(key, snap) -> None:
    """Back to what the back-to-back search left for the shape: its hand-written plans (`snap`) and, on the library
    side, the fastest finalist (= the search's own pick: the finalists are sorted by its times)."""
    for d, v in zip((_M256_PLAN, _WSTREAM_PLAN, _FUSED_SILU_PLAN, _RO_PLAN, _RO_SILU_PLAN, _SKINNY_PLAN, _SKINNY_SILU_PLAN), snap):
        if v is None:
            d.pop(key, None)
        else:
            d[key] = v
    if any(spec[0] == "lib" for _l, spec in _CANDIDATES.get(key, {}).get("cands", [])):
        dev, M, N, K, ldx, ldw, dt = key
        _lib.check_gemm(_lib.gemm_lib().msgl_gemm_select_finalist(M, N, K, ldx, ldw, _CANDIDATES[key]["out_ld"], dt, 0),
                        "gemm_select_finalist")


_PLAN_TABLES = {"skinny": _SKINNY_PLAN, "skinny_silu": _SKINNY_SILU_PLAN, "wstream": _WSTREAM_PLAN, "m256": _M256_PLAN,
                "fused_silu": _FUSED_SILU_PLAN, "ro": _RO_PLAN, "ro_silu": _RO_SILU_PLAN}


def export_gemm_plans(shapes) -> dict:
    """Every per-shape kernel choice of this process as plain data: the hand-written kernels' plan tables and, for each
    (M, N, K, ldx, ldw, ldo, dtype code) in `shapes`, the library's searched solution (index, split-K).  import_gemm_plans in
    another process then runs `linear` / `linear_silu` / `linear_slabs` on exactly the same kernels."""
    import ctypes as C

    tables = {n: {repr(k): list(v) for k, v in t.items()} for n, t in _PLAN_TABLES.items()}
    library = []
    for (M, N, K, ldx, ldw, ldo, dt) in shapes:
        idx, sk = C.c_int(-1), C.c_int(0)
        rc = _lib.gemm_lib().msgl_gemm_get_plan(M, N, K, ldx, ldw, ldo, dt, C.byref(idx), C.byref(sk))
        _lib.check_gemm(rc, "gemm_get_plan")
        if rc == 1:
            library.append([M, N, K, ldx, ldw, ldo, dt, idx.value, sk.value])
    return dict(tables=tables, library=library)


def import_gemm_plans(plans: dict, device_index: int = 0, reset: bool = True) -> None:
    """Install what export_gemm_plans recorded (plan keys are re-targeted at `device_index`); reset: drop everything else first."""
    import ast

    if reset:
        reset_gemm_plans()
    for name, table in plans["tables"].items():
        for k, v in table.items():
            key = ast.literal_eval(k)
            _PLAN_TABLES[name][(device_index,) + tuple(key[1:])] = tuple(v)
    ws = gemm_workspace(torch.device("cuda", device_index))
    with torch.cuda.device(device_index):  # msgl_gemm_set_plan files the plan under the CURRENT HIP device's handle
        for M, N, K, ldx, ldw, ldo, dt, idx, sk in plans["library"]:
            _lib.check_gemm(_lib.gemm_lib().msgl_gemm_set_plan(M, N, K, ldx, ldw, ldo, dt, idx, sk, ws.data_ptr(), ws.numel()),
                            "gemm_set_plan")


def reset_gemm_plans() -> None:
    """Drop every per-shape kernel choice made so far in this process (hand-written kernel plans and library
    solutions): `linear` is the library's heuristic again until the next search."""
    _SKINNY_PLAN.clear()
    _WSTREAM_PLAN.clear()
    _M256_PLAN.clear()
    _RO_PLAN.clear()
    _RO_SILU_PLAN.clear()
    _FUSED_SILU_PLAN.clear()
    _SKINNY_SILU_PLAN.clear()
    _CANDIDATES.clear()
    _PENDING_SLABS.clear()
    _PENDING_ALLREDUCE.clear()
    _lib.check_gemm(_lib.gemm_lib().msgl_gemm_reset_plans(), "gemm_reset_plans")


def gemm_tune(x: torch.Tensor, weights, out: Optional[torch.Tensor] = None, max_candidates: int = 0,
              iters: int = 10, split_k: bool = True) -> dict:
    """Search the library's solutions for x @ w^T over the same-shaped `weights` (rotated so the
    Infinity Cache cannot hold them) and remember the fastest.  Synchronises; call before capture."""
    import ctypes as C

    weights = list(weights)
    w0 = weights[0]
    out, M, N, K = _gemm_args(x, w0, out)
    for w in weights:
        assert w.shape == w0.shape and w.stride() == w0.stride() and w.dtype == w0.dtype and w.is_cuda
    ws = gemm_workspace(x.device)
    ptrs = (C.c_void_p * len(weights))(*[w.data_ptr() for w in weights])
    best, default = C.c_float(0), C.c_float(0)
    idx, tried, split = C.c_int(-1), C.c_int(0), C.c_int(0)
    _lib.check_gemm(
        _lib.gemm_lib().msgl_gemm_tune(out.data_ptr(), x.data_ptr(), ptrs, len(weights), M, N, K, x.stride(0),
                                       w0.stride(0), out.stride(0), _dt(x), ws.data_ptr(), ws.numel(),
                                       max_candidates, int(split_k), iters, C.byref(best), C.byref(default),
                                       C.byref(idx), C.byref(split), C.byref(tried), _stream()),
        "gemm_tune",
    )
    buf = C.create_string_buffer(512)
    _lib.gemm_lib().msgl_gemm_solution_name(M, N, K, x.stride(0), w0.stride(0), out.stride(0), _dt(x), buf, 512)
    return dict(M=M, N=N, K=K, best_us=best.value, default_us=default.value, index=idx.value, split_k=split.value,
                tried=tried.value,
                kernel=buf.value.decode(errors="replace"))
