"""CPU oracle: restatement of the reference's hot-path semantics (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (mini-sglang_amd/) never does and has no CPU fallback.

Every function cites the reference file:line it follows (paths relative to the reference
root, P/ = python/minisgl/, C/ = python/minisgl/kernel/csrc/).  Where the arithmetic
lives in a third-party dependency that is not vendored under /root/reference
(flashinfer-python>=0.5.3, sgl_kernel>=0.3.17.post1: pyproject.toml:30,37 -- version
ranges only, no lock file) the function restates the published behaviour at the
reference's call site; the reference has no test that pins those
numerics, so for attention / RMSNorm / RoPE / activation / sampling **parity is
unpinned** beyond the tolerances stated in tests/.  Integer / byte paths (store, gather,
key compare, metadata, sampler clamps, RoPE cache construction) ARE pinned against the
reference's own Python, see tests/golden/make_golden.py.

All math is fp32 on inputs upcast exactly from bf16/fp16 (fp64 where noted).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch


# --------------------------------------------------------------------------- byte movers
def store_kv_ref(k_cache: torch.Tensor, v_cache: torch.Tensor, indices: torch.Tensor, k: torch.Tensor,
                 v: torch.Tensor) -> None:
    """C/jit/store.cu:42-50: k_cache[indices[w]] = k[w]; v_cache[indices[w]] = v[w]."""
    idx = indices.long()
    k_cache[idx] = k
    v_cache[idx] = v


def indexing_ref(weights: torch.Tensor, indices: torch.Tensor,
                 vocab_range: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """C/jit/index.cu:51-56 (plain) and :84-92 (masked: pos = idx - start as UNSIGNED;
    out = pos < length ? W[pos] : 0).  Same result as tests/kernel/test_index.py:13-30."""
    idx = indices.long()
    if vocab_range is None:
        return weights[idx].clone()
    start, length = vocab_range
    pos = idx - start
    ok = (pos >= 0) & (pos < length)  # unsigned compare == both bounds
    out = torch.zeros((idx.shape[0], weights.shape[1]), dtype=weights.dtype)
    out[ok] = weights[pos[ok]]
    return out


def fast_compare_key_ref(x: torch.Tensor, y: torch.Tensor) -> int:
    """C/src/radix.cpp:19-40: index of the first mismatch over the common length."""
    n = min(x.numel(), y.numel())
    neq = (x[:n] != y[:n]).nonzero()
    return int(neq[0]) if neq.numel() else n


# --------------------------------------------------------------------------- norms
def rmsnorm_ref(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """flashinfer.rmsnorm as called at P/layers/norm.py:17,20 (Appendix B):
    y = x_f32 * rsqrt(mean(x_f32^2) + eps) * w_f32, cast to x.dtype; last-dim rows."""
    xf = x.float()
    inv = torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    return (xf * inv * w.float()).to(x.dtype)


def fused_add_rmsnorm_ref(x: torch.Tensor, residual: torch.Tensor, w: torch.Tensor,
                          eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """flashinfer.fused_add_rmsnorm at P/layers/norm.py:37: s = x + residual (fp32);
    residual <- cast(s); x <- cast(rmsnorm(s) * w) with the UNROUNDED fp32 sum."""
    s = x.float() + residual.float()
    inv = torch.rsqrt(s.pow(2).mean(dim=-1, keepdim=True) + eps)
    return (s * inv * w.float()).to(x.dtype), s.to(x.dtype)


# --------------------------------------------------------------------------- RoPE
def rope_inv_freq(rotary_dim: int, base: float, rope_scaling: Optional[Dict[str, Any]] = None) -> torch.Tensor:
    """P/layers/rotary.py:24-26 plus the llama3 (:69-91) and yarn (:93-112) post-processing."""
    inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
    if rope_scaling is None or rope_scaling.get("rope_type", "default") == "default":
        return inv_freq
    kind = rope_scaling["rope_type"]
    if kind == "llama3":
        factor = rope_scaling["factor"]
        lo, hi = rope_scaling["low_freq_factor"], rope_scaling["high_freq_factor"]
        orig = rope_scaling["original_max_position_embeddings"]
        wave_len = 2 * math.pi / inv_freq
        if lo == hi:
            return torch.where(wave_len < orig / hi, inv_freq, inv_freq / factor)
        smooth = torch.clamp((orig / wave_len - lo) / (hi - lo), 0, 1)
        return ((1 - smooth) / factor + smooth) * inv_freq
    if kind == "yarn":
        factor = rope_scaling["factor"]
        beta_fast = rope_scaling.get("beta_fast", 32.0)
        beta_slow = rope_scaling.get("beta_slow", 1.0)
        orig = rope_scaling["original_max_position_embeddings"]

        def corr(num_rot: float) -> float:
            return rotary_dim * math.log(orig / (num_rot * 2 * math.pi)) / (2 * math.log(base))

        low = max(math.floor(corr(beta_fast)), 0)
        high = min(math.ceil(corr(beta_slow)), rotary_dim // 2 - 1)
        ramp = torch.clamp((torch.arange(rotary_dim // 2, dtype=torch.float32) - low) / max(high - low, 1), 0, 1)
        return (inv_freq / factor) * ramp + inv_freq * (1 - ramp)
    raise ValueError(f"unsupported rope scaling {rope_scaling}")


def rope_cos_sin_cache(rotary_dim: int, max_position: int, base: float,
                       rope_scaling: Optional[Dict[str, Any]] = None) -> torch.Tensor:
    """P/layers/rotary.py:27-32: fp32 [max_pos, rotary_dim] = cat(cos, sin) of t (x) inv_freq."""
    inv_freq = rope_inv_freq(rotary_dim, base, rope_scaling)
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


def rope_neox_ref(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, head_size: int,
                  cache: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """flashinfer.apply_rope_with_cos_sin_cache_inplace (is_neox=True) at
    P/layers/rotary.py:45-51: x'[i] = x[i]c - x[i+D/2]s ; x'[i+D/2] = x[i+D/2]c + x[i]s
    in fp32, for every q and k head; returns new tensors (the op itself is in place)."""
    cs = cache[positions.long()]
    cos, sin = cs[:, : head_size // 2].unsqueeze(1), cs[:, head_size // 2:].unsqueeze(1)

    def rot(x: torch.Tensor) -> torch.Tensor:
        t = x.shape[0]
        xf = x.float().reshape(t, -1, head_size)
        a, b = xf[..., : head_size // 2], xf[..., head_size // 2:]
        return torch.cat((a * cos - b * sin, b * cos + a * sin), dim=-1).reshape(t, -1).to(x.dtype)

    return rot(q), rot(k)


def silu_and_mul_ref(x: torch.Tensor) -> torch.Tensor:
    """flashinfer.silu_and_mul at P/layers/activation.py:9-12: silu(x[:, :d]) * x[:, d:]."""
    d = x.shape[-1] // 2
    g, u = x[..., :d].float(), x[..., d:].float()
    return (g / (1.0 + torch.exp(-g)) * u).to(x.dtype)


# --------------------------------------------------------------------------- attention
def paged_attention_ref(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                        page_table: torch.Tensor, req_rows: Sequence[int], seq_lens: Sequence[int],
                        q_lens: Sequence[int], sm_scale: float, double: bool = False) -> torch.Tensor:
    """Appendix B clause "paged attention" (call contract P/attention/fa.py:139-182,
    metadata P/attention/fa.py:67-105):
      q [T, Hq, D] (requests concatenated), k_cache/v_cache [slots, Hkv, D],
      page_table[row, t] = token slot of position t (page-size agnostic, P/core.py:103-104).
    Request i has q_i new tokens (already stored) and k_i = seq_lens[i] total tokens; query
    row j attends keys t <= k_i - q_i + j; P = softmax(scale * q.K^T) and o = P.V in fp32
    (fp64 if double); query head h uses kv head h // (Hq/Hkv)."""
    T, hq, d = q.shape
    hkv = k_cache.shape[1]
    group = hq // hkv
    acc = torch.float64 if double else torch.float32
    out = torch.zeros((T, hq, d), dtype=acc)
    off = 0
    for row, k_len, q_len in zip(req_rows, seq_lens, q_lens):
        slots = page_table[row, :k_len].long()
        K = k_cache[slots].to(acc)  # [k, Hkv, D]
        V = v_cache[slots].to(acc)
        Q = q[off: off + q_len].to(acc)  # [q, Hq, D]
        Kx = K.repeat_interleave(group, dim=1)  # [k, Hq, D]
        Vx = V.repeat_interleave(group, dim=1)
        s = torch.einsum("qhd,khd->hqk", Q, Kx) * sm_scale
        qpos = torch.arange(q_len).unsqueeze(1) + (k_len - q_len)
        mask = torch.arange(k_len).unsqueeze(0) <= qpos  # bottom-right aligned causal
        s = s.masked_fill(~mask.unsqueeze(0), float("-inf"))
        p = torch.softmax(s, dim=-1)
        out[off: off + q_len] = torch.einsum("hqk,khd->qhd", p, Vx)
        off += q_len
    return out if double else out.to(q.dtype)


# --------------------------------------------------------------------------- sampling
@dataclass
class SamplingParamsRef:
    """P/core.py:16-25."""
    temperature: float = 0.0
    top_k: int = -1
    top_p: float = 1.0

    @property
    def is_greedy(self) -> bool:
        return (self.temperature <= 0.0 or self.top_k == 1) and self.top_p == 1.0


def sampler_prepare_ref(params: Sequence[SamplingParamsRef], vocab_size: int):
    """P/engine/sample.py:53-68: returns (temperatures, top_k, top_p) python lists or None."""
    if all(p.is_greedy for p in params):
        return None, None, None
    MIN_P = MIN_T = 1e-6
    ts = [max(0.0 if p.is_greedy else p.temperature, MIN_T) for p in params]
    top_ks = [p.top_k if p.top_k >= 1 else vocab_size for p in params]
    top_ps = [min(max(p.top_p, MIN_P), 1.0) for p in params]
    top_k = top_ks if any(k != vocab_size for k in top_ks) else None
    top_p = top_ps if any(p < 1.0 for p in top_ps) else None
    return ts, top_k, top_p


def argmax_ref(logits: torch.Tensor) -> torch.Tensor:
    """P/engine/sample.py:73-74 + P/engine/engine.py:202: first index of the row max, int32."""
    return torch.argmax(logits.float(), dim=-1).to(torch.int32)


def softmax_temperature_ref(logits: torch.Tensor, temperatures: torch.Tensor) -> torch.Tensor:
    """flashinfer.sampling.softmax(logits, temperatures) at P/engine/sample.py:32."""
    return torch.softmax(logits.double() / temperatures.double().unsqueeze(1), dim=-1)


def top_k_top_p_filter_ref(probs: torch.Tensor, top_k: Optional[Sequence[int]],
                           top_p: Optional[Sequence[float]]) -> torch.Tensor:
    """Target distribution of flashinfer's (joint) top-k / top-p sampling ops
    (P/engine/sample.py:33-45, Appendix B "stochastic"): keep the top_k most probable
    tokens (ties at the k-th value kept), renormalise, keep the smallest descending-prob
    prefix whose mass >= top_p (ties at the cut kept), renormalise.  fp64."""
    out = torch.zeros_like(probs, dtype=torch.float64)
    for r in range(probs.shape[0]):
        p = probs[r].double().clone()
        if top_k is not None and top_k[r] < p.numel():
            kth = torch.topk(p, int(top_k[r])).values[-1]
            p = torch.where(p >= kth, p, torch.zeros_like(p))
        p = p / p.sum()
        if top_p is not None and top_p[r] < 1.0:
            sp, _ = torch.sort(p, descending=True)
            csum = torch.cumsum(sp, 0)
            cut = int((csum < top_p[r]).sum())  # first index where mass >= top_p
            thr = sp[min(cut, p.numel() - 1)]
            p = torch.where(p >= thr, p, torch.zeros_like(p))
            p = p / p.sum()
        out[r] = p
    return out


# --------------------------------------------------------------------------- attention metadata
@dataclass
class ReqRef:
    """The fields of P/core.py:29-58 the backend reads."""
    table_idx: int
    cached_len: int
    device_len: int

    @property
    def extend_len(self) -> int:
        return self.device_len - self.cached_len


def fa_metadata_ref(reqs: Sequence[ReqRef], page_table: torch.Tensor, page_size: int) -> Dict[str, Any]:
    """P/attention/fa.py:67-105: the integer metadata of one batch."""
    seqlens_q = [r.extend_len for r in reqs]
    seqlens_k = [r.device_len for r in reqs]
    cached = [r.cached_len for r in reqs]
    max_k, max_q = max(seqlens_k), max(seqlens_q)
    cu_k = torch.tensor([0] + seqlens_k, dtype=torch.int32).cumsum(0).to(torch.int32)
    if max_q == 1:
        cu_q = torch.arange(0, len(reqs) + 1, dtype=torch.int32)
    elif all(c == 0 for c in cached):
        cu_q = cu_k
    else:
        cu_q = torch.tensor([0] + seqlens_q, dtype=torch.int32).cumsum(0).to(torch.int32)
    new_table = torch.stack([page_table[r.table_idx, :max_k:page_size] for r in reqs])
    if page_size > 1:
        new_table = torch.div(new_table, page_size, rounding_mode="floor")
    return dict(cu_seqlens_k=cu_k, cu_seqlens_q=cu_q, cache_seqlens=torch.tensor(seqlens_k, dtype=torch.int32),
                max_seqlen_k=max_k, max_seqlen_q=max_q, page_table=new_table,
                last_indices=cu_q[1: 1 + len(reqs)] - 1)  # get_last_indices, fa.py:32-33


# --------------------------------------------------------------------------- collectives
def all_reduce_sum_ref(per_rank: List[torch.Tensor]) -> torch.Tensor:
    """C/src/pynccl.cu:93-133 semantics: element-wise SUM over ranks (order unspecified)."""
    return torch.stack([t.float() for t in per_rank]).sum(0).to(per_rank[0].dtype)


def all_gather_ref(per_rank: List[torch.Tensor]) -> torch.Tensor:
    """C/src/pynccl.cu:136-161: rank chunks concatenated along dim 0 in rank order."""
    return torch.cat(per_rank, dim=0)
